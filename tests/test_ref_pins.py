"""The oracle (and the shipped data) held to outputs of the REFERENCE'S OWN PYTHON run in the build
container -- tests/golden/ref_* from tools/make_golden_ref.py:

  ref_recur.json  software/rnghunt/util/binarymatrix.py:30-35 recur(), imported and called
  ref_clt.npz     software/clt-grng/clt-grng-evaluate.py executed unchanged under a seeded np.random
  ref_pack.json   software/rnghunt/util/pack.py run on matrices/N; gateware/bbb/rng_recurrences.py imported
  ref_words.npz   software/rnghunt/util/verify.py run: its dieharder dump of 200 000 states
  ref_lfsr.json   software/rnghunt/util/lfsr.py run

These are CPU tests (oracle + data); tests/test_gpu_ref_pins.py holds the HIP path to the same files."""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

NS = (16, 32, 64, 128, 192, 256, 512)


@pytest.fixture(scope="module")
def ref_recur():
    return json.load(open(GOLDEN / "ref_recur.json"))


@pytest.mark.parametrize("n", NS)
def test_lutopt_states_equal_reference_recur(oracle, ref_recur, n):
    """Every state bit of the first 64 steps and bit 0 of 4096 steps, both seeds, every shipped matrix."""
    m = oracle.Lutopt(path=oracle.data_path(n))
    for label in ("init1", "seed2"):
        ent = ref_recur[str(n)][label]
        init = int(ent["init"], 16)
        st = m.states(init, 0, 4096)
        got = [oracle.words_to_int(w, n) for w in st[:64]]
        assert got == [int(h, 16) for h in ent["states_hex"]]
        assert "".join(str(int(w[0]) & 1) for w in st) == ent["bit0"]
        # the literal step function, one call per step, agrees with the bulk form
        x = init
        for h in ent["states_hex"][:8]:
            x = m.step_int(x)
            assert x == int(h, 16)


@pytest.mark.parametrize("n", NS)
def test_product_jump_ahead_equals_reference_recur(ref_recur, n):
    """bbb_lutopt_state_at (the product's host-side GF(2) jump-ahead; no GPU needed for it) on every shipped
    matrix, including the non-power-of-two n192 (gateware/bbb/rng_recurrences.py:105)."""
    import basebandboard_amd as bbb
    for label in ("init1", "seed2"):
        ent = ref_recur[str(n)][label]
        u = bbb.LUTOPT.shipped(n, init=int(ent["init"], 16), device=-1)
        for t, h in enumerate(ent["states_hex"]):
            assert u.state_at(t + 1) == int(h, 16)
        for t in (100, 1000, 4096):
            assert u.state_at(t) & 1 == int(ent["bit0"][t - 1])


def test_reference_recur_equals_the_retyped_fixture(ref_recur, golden_lutopt):
    """The vectors tools/make_golden.py derived by re-typing rng.py:134-135 equal what the reference's
    own recur() produced."""
    for n in (16, 32, 64, 128, 256):
        assert ref_recur[str(n)]["init1"]["states_hex"] == golden_lutopt[str(n)]["states_hex"]


def test_clt_tree_equals_reference_script(oracle):
    """All 100 000 samples of software/clt-grng/clt-grng-evaluate.py (seeded run) from its own input bits."""
    z = np.load(GOLDEN / "ref_clt.npz")
    states, samples = z["states"], z["samples"]
    assert states.shape == (100000, 4) and samples.shape == (100000,) and int(z["n"]) == 256
    m = oracle.Lutopt(path=oracle.data_path(256))
    assert np.array_equal(m.clt_tree_bulk(states), samples)
    # closed form and the one-at-a-time entry on a subset; output truncation leaves these values alone
    for i in range(0, 100000, 997):
        x = oracle.words_to_int(states[i], 256)
        assert m.clt_tree(x) == m.clt_popcount(x) == int(samples[i]) == m.clt_wrap(int(samples[i]))
    meta = json.load(open(GOLDEN / "ref_clt_meta.json"))
    assert abs(meta["mean"]) < 0.1 and abs(meta["var"] - 64.0) < 1.0        # clt-grng-evaluate.py:18-31
    assert any("6.4000e+01" in l for l in meta["printed"])


@pytest.mark.parametrize("n", NS)
def test_shipped_tap_lists_equal_reference_pack_output(n):
    ent = json.load(open(GOLDEN / "ref_pack.json"))[str(n)]
    mine = [[int(x) for x in l.split()] for l in open(ROOT / "basebandboard_amd" / "data" / f"lutopt_{n}.taps") if l.strip()]
    assert mine == ent["pack_py"]
    if n <= 256:
        assert ent["rng_recurrences"] == ent["pack_py"]     # gateware/bbb/rng_recurrences.py nN


@pytest.mark.parametrize("n", (192, 256))
def test_uniform_word_stream_equals_reference_dump(oracle, n):
    """software/rnghunt/util/verify.py:37-52: 200 000 states as 32-bit words (x[32j] = MSB of word j)."""
    z = np.load(GOLDEN / "ref_words.npz")
    meta = json.load(open(GOLDEN / "ref_words_meta.json"))[str(n)]
    init = sum(int(b) << i for i, b in enumerate(z[f"init_bits_{n}"]))
    m = oracle.Lutopt(path=oracle.data_path(n))
    words = m.words_u32(init, 0, meta["nstates"], msb_first=True)
    wps = meta["words_per_state"]
    assert np.array_equal(words[:2048 * wps], z[f"head_{n}"])
    assert np.array_equal(words[-512 * wps:], z[f"tail_{n}"])
    assert hashlib.sha256(words.astype("<u4").tobytes()).hexdigest() == meta["sha256_le_u32"]
    # the LSB-first form is the same bits mirrored inside each word
    lsb = m.words_u32(init, 0, 64, msb_first=False)
    rev = np.array([int(f"{int(w):032b}"[::-1], 2) for w in words[:64 * wps]], dtype=np.uint32)
    assert np.array_equal(lsb, rev)


def test_lfsr_strings_of_the_reference_script():
    """software/rnghunt/util/lfsr.py prints the strings its Rust Berlekamp-Massey tests hold
    (berlekamp_massey.rs:50-65); Berlekamp-Massey on them gives the LFSRs the script implements."""
    lines = json.load(open(GOLDEN / "ref_lfsr.json"))["lines"]
    assert len(lines) == 2 and len(lines[0]) == 32 and len(lines[1]) == 128
    from oracle import gf2poly
    # lfsr.py:5-6 taps 0,2,3,5 of a 16-bit right-shifting register -> b[t] = b[t-16]^b[t-14]^b[t-13]^b[t-11];
    # lfsr.py:13-14 taps 0,2,3,63 of 64 bits -> b[t] = b[t-64]^b[t-62]^b[t-61]^b[t-1]
    want = ((1 | 1 << 11 | 1 << 13 | 1 << 14 | 1 << 16, 16), (1 | 1 << 1 | 1 << 61 | 1 << 62 | 1 << 64, 64))
    for s, w in zip(lines, want):
        assert gf2poly.berlekamp_massey([int(c) for c in s]) == w


@pytest.mark.parametrize("k", (7, 9, 11, 15, 20, 23, 31))
def test_prbs_prefix_and_state_of_survey_appendix_b(oracle, k):
    """The survey's known-answer lines for PRBS(k) from reset state 1 (from the reference's model
    prbs.py:112-113): the first 64 emitted bits and the LFSR state after them -- the one literal this repository holds
    for PRBS-31, whose 2^31 - 1 period no reference file spells out."""
    bits, s = json.load(open(GOLDEN / "survey_appendix_b_prbs.json"))["vectors"][str(k)]
    got, state = oracle.prbs_bits(k, 64)
    assert "".join(str(int(b)) for b in got) == bits
    assert state == int(s, 16)
