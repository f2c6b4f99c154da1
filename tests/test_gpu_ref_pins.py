"""The HIP path held to outputs of the REFERENCE'S OWN PYTHON run in the build container
(tests/golden/ref_*, made by tools/make_golden_ref.py; tests/test_ref_pins.py holds the oracle to the
same files).  Everything goes through the C ABI; bit exact."""
import hashlib
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
NS = (16, 32, 64, 128, 192, 256, 512)


@pytest.fixture(scope="module")
def ref_recur():
    return json.load(open(GOLDEN / "ref_recur.json"))


def _words64(states_hex, n):
    """reference states (HDL integers) -> int64 tensor [nstates, ceil(n/64)] of bit patterns"""
    nw = (n + 63) // 64
    a = np.zeros((len(states_hex), nw), dtype=np.uint64)
    for i, h in enumerate(states_hex):
        v = int(h, 16)
        for w in range(nw):
            a[i, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return torch.from_numpy(a.view(np.int64)).cuda()


def test_clt_tree_equals_reference_script(gpu):
    """bbb_clt_tree_i16 on the 100 000 input words of software/clt-grng/clt-grng-evaluate.py (seeded run)
    equals the script's own `samples`."""
    z = np.load(GOLDEN / "ref_clt.npz")
    st = torch.from_numpy(z["states"].view(np.int64)).cuda()
    got = gpu.CLTGRNG.tree(st, 256).cpu().numpy()
    assert np.array_equal(got, z["samples"])


@pytest.mark.parametrize("n", NS)
def test_states_and_sample_stream_equal_reference_recur(gpu, ref_recur, n):
    """binarymatrix.recur() pinned every bit of the first 64 states: the device's word stream (k % 32 == 0)
    reproduces them, and the sample stream equals the adder tree -- itself pinned by the reference script
    above -- applied to those reference states."""
    for label in ("init1", "seed2"):
        ent = ref_recur[str(n)][label]
        init = int(ent["init"], 16)
        u = gpu.LUTOPT.shipped(n, init=init)
        states = [int(h, 16) for h in ent["states_hex"]]
        assert [u.state_at(t + 1) for t in range(64)] == states
        if n % 32 == 0:
            w = u.generate_words(64).cpu().numpy().view(np.uint32).reshape(64, n // 32)
            got = [sum(int(x) << (32 * j) for j, x in enumerate(row)) for row in w]
            assert got == states
            # bit 0 of 4096 states
            w = u.generate_words(4096).cpu().numpy().view(np.uint32).reshape(4096, n // 32)
            assert "".join(str(int(x) & 1) for x in w[:, 0]) == ent["bit0"]
        if n & (n - 1) == 0:
            g = gpu.CLTGRNG(u)
            got = g.generate(64).cpu().numpy().astype(np.int64)
            if n >= 64:
                tree = gpu.CLTGRNG.tree(_words64(ent["states_hex"], n), n).cpu().numpy().astype(np.int64)
            else:       # bbb_clt_tree_i16 takes whole 64-bit words; closed form for the two small orders
                tree = np.array([sum((-1) ** bin(i).count("1") * ((s >> i) & 1) for i in range(n)) for s in states])
            logn = n.bit_length() - 1
            wrapped = ((tree + (1 << (logn - 1))) % (1 << logn)) - (1 << (logn - 1))      # rng.py:78,108
            assert np.array_equal(got, wrapped)
        else:
            with pytest.raises(gpu._lib.BbbError) as e:      # CLTGRNG needs a power of two (rng.py:72-76)
                gpu._lib.check(gpu._lib.lib().bbb_awgn_fill_i8(u._h, 16, 16, 0), "bbb_awgn_fill_i8")
            assert e.value.code == gpu._lib.BBB_EUNSUP


@pytest.mark.parametrize("n", (192, 256))
def test_word_stream_equals_reference_dieharder_dump(gpu, n):
    """bbb_lutopt_fill_words(msb_first) = the `outnums` file software/rnghunt/util/verify.py:37-52 wrote:
    all 200 000 states (sha256), head and tail literally."""
    z = np.load(GOLDEN / "ref_words.npz")
    meta = json.load(open(GOLDEN / "ref_words_meta.json"))[str(n)]
    init = sum(int(b) << i for i, b in enumerate(z[f"init_bits_{n}"]))
    u = gpu.LUTOPT.shipped(n, init=init)
    wps = meta["words_per_state"]
    words = u.generate_words(meta["nstates"], msb_first=True).cpu().numpy().view(np.uint32)
    assert np.array_equal(words[:2048 * wps], z[f"head_{n}"])
    assert np.array_equal(words[-512 * wps:], z[f"tail_{n}"])
    assert hashlib.sha256(words.astype("<u4").tobytes()).hexdigest() == meta["sha256_le_u32"]
    # a window in the middle, asked for on its own (jump-ahead), equals the same window of the whole run
    mid = u.generate_words(1000, first_step=123_456, msb_first=True).cpu().numpy().view(np.uint32)
    assert np.array_equal(mid, words[123_456 * wps:(123_456 + 1000) * wps])


@pytest.mark.parametrize("n", (32, 64, 128, 192, 256, 512))
def test_word_stream_matches_oracle(gpu, oracle, n):
    m = oracle.Lutopt(path=oracle.data_path(n))
    u = gpu.LUTOPT.shipped(n, init=0x1234567 | 1)
    for nstates, first, msb in ((1, 0, False), (63, 5, True), (70_001, 16, False), (300_000, 999, True)):
        got = u.generate_words(nstates, first_step=first, msb_first=msb).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, m.words_u32(0x1234567 | 1, first, nstates, msb_first=msb))


def test_word_stream_needs_whole_words(gpu):
    u = gpu.LUTOPT.shipped(16)
    with pytest.raises(ValueError):
        u.generate_words(4)


@pytest.mark.parametrize("k", (7, 9, 11, 15, 20, 23, 31))
def test_prbs_prefix_and_state_of_survey_appendix_b(gpu, k):
    """SURVEY.md Appendix B's PRBS(k) lines (reference model prbs.py:112-113) on the device: bbb_prbs_fill's first 64 bits and
    bbb_prbs_state_at(64); for k = 31 also as the head of the 1e10-bit fill of BASELINE configs[2] (the same kernel at the
    size the metric is quoted on: bit t at word t / 64, bit t % 64)."""
    import json
    import pathlib
    bits, s = json.load(open(pathlib.Path(__file__).parent / "golden" / "survey_appendix_b_prbs.json"))["vectors"][str(k)]
    p = gpu.PRBS(k)
    w = int(p.generate(64).cpu().numpy().view(np.uint64)[0])
    assert "".join(str((w >> t) & 1) for t in range(64)) == bits
    assert p.state_at(64) == int(s, 16)
    if k == 31:
        big = p.generate(10_000_000_000)
        assert int(big[:1].cpu().numpy().view(np.uint64)[0]) == w
        del big
