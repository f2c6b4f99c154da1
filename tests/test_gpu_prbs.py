"""PRBS kernels vs the oracle and the golden vectors (through the C ABI).  Bit exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KS = (7, 9, 11, 15, 20, 23, 31)


def unpack(words_i64, nbits):
    w = words_i64.cpu().numpy().view(np.uint64)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    return bits[:nbits]


@pytest.mark.parametrize("k", KS)
def test_prbs_golden_prefix(gpu, golden_prbs, k):
    """First 4096 bits from state 1 equal the reference model's (prbs.py:112-113)."""
    got = unpack(gpu.PRBS(k).generate(4096), 4096)
    assert "".join(map(str, got)) == golden_prbs[str(k)]["bits"]
    g2 = golden_prbs[f"{k}_seed2"]
    got = unpack(gpu.PRBS(k, init=g2["init"]).generate(1024), 1024)
    assert "".join(map(str, got)) == g2["bits"]


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("nbits,first", [(1, 0), (63, 0), (64, 5), (65, 0), (8191, 3), (8193, 0),
                                         (1_000_003, 0), (300_000, 123_456_789), (5_000_000, 2**40 + 17)])
def test_prbs_fill_matches_oracle(gpu, oracle, k, nbits, first):
    p = gpu.PRBS(k)
    got = p.generate(nbits, first_bit=first).cpu().numpy().view(np.uint64)
    s0 = p.state_at(first)
    exp, s_end = oracle.prbs_packed(k, nbits, state=s0, fast=True)
    assert np.array_equal(got, exp)
    assert p.state_at(first + nbits) == s_end


@pytest.mark.parametrize("k", (7, 20, 31))
def test_prbs_state_at_small(gpu, oracle, k):
    p = gpu.PRBS(k)
    for n in (0, 1, 2, 63, 64, 1000):
        _, s = oracle.prbs_bits(k, n)
        assert p.state_at(n) == s


@pytest.mark.parametrize("k", KS)
def test_prbs_loopback_and_injected_errors(gpu, k):
    """Generator -> checker: 0 errors clean; an XOR mask of known weight is counted exactly."""
    nbits = 20_000_003
    p = gpu.PRBS(k)
    det = gpu.PRBSErrorDetector(k)
    buf = p.generate(nbits, first_bit=77)
    assert det.count_errors(buf, nbits, first_bit=77) == 0
    rng = np.random.default_rng(k)
    pos = np.unique(rng.integers(0, nbits, size=5000))
    mask = np.zeros(buf.numel(), dtype=np.uint64)
    np.bitwise_xor.at(mask, pos // 64, np.uint64(1) << (pos % 64).astype(np.uint64))
    buf ^= torch.from_numpy(mask.view(np.int64)).to(buf.device)
    assert det.count_errors(buf, nbits, first_bit=77) == len(pos)
    # garbage above nbits in the last word must be ignored
    buf2 = p.generate(1000)
    buf2[-1] |= torch.tensor(-1 << (1000 % 64), dtype=torch.int64, device=buf2.device)
    assert det.count_errors(buf2, 1000) == 0


def test_prbs_invalid_k(gpu):
    for k in (0, 8, 10, 32, 63):
        with pytest.raises(ValueError, match="invalid for PRBS"):
            gpu.PRBS(k)
        with pytest.raises(ValueError, match="invalid for PRBS"):
            gpu.PRBSErrorDetector(k)


def test_prbs_empty(gpu):
    assert gpu.PRBS(31).generate(0).numel() == 0


@pytest.mark.parametrize("k", KS)
def test_prbs_detector_fsm_matches_oracle(gpu, oracle, k):
    """Cycle-exact FSM, many independent streams at once, random error patterns incl. bursts."""
    rng = np.random.default_rng(100 + k)
    nstreams, n = 300, 700
    tx, _ = oracle.prbs_bits(k, n)
    bits = np.tile(tx, (nstreams, 1))
    errs = (rng.random((nstreams, n)) < 0.03).astype(np.uint8)
    errs[:, : 2 * k] = 0
    errs[::3, n // 2: n // 2 + 3 * k] = 1
    bits ^= errs
    bits[5] = rng.integers(0, 2, n)            # pure noise stream: reload logic exercised hard
    e, r = gpu.PRBSErrorDetector(k).run(torch.from_numpy(bits).cuda())
    e, r = e.cpu().numpy(), r.cpu().numpy()
    for s in range(0, nstreams, 7):
        eo, ro = oracle.prbs_detector_run(k, bits[s])
        assert np.array_equal(e[s], eo) and np.array_equal(r[s], ro), s


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("seed", range(4))
def test_prbs_detector_reference_protocol(gpu, oracle, k, seed):
    """The reference's own test design (prbs.py:124-163), see tests/detector_protocol.py."""
    from detector_protocol import make_case, check_case
    wire, tx_errors = make_case(k, lambda kk, n: oracle.prbs_bits(kk, n)[0], seed)
    e, r = gpu.PRBSErrorDetector(k).run(torch.from_numpy(wire[None, :]).cuda())
    check_case(tx_errors, e[0].cpu().numpy(), r[0].cpu().numpy())


@pytest.mark.parametrize("k", KS)
def test_prbs_detector_reference_protocol_literal_windows(gpu, oracle, k):
    """The same with the reference's own 2k / 3k / 2k windows (prbs.py:133-138), twelve seeded draws per k run as twelve
    parallel streams: the device's outputs equal the oracle's on every draw, and the reference's assertion holds on every
    draw in which no injected error meets a reload window (tests/test_oracle.py has the count and the k = 7 case)."""
    from detector_protocol import make_case, check_case, errors_inside_reload
    cases = [make_case(k, lambda kk, n: oracle.prbs_bits(kk, n)[0], seed, literal=True) for seed in range(12)]
    wires = np.stack([c[0] for c in cases])
    e, r = gpu.PRBSErrorDetector(k).run(torch.from_numpy(wires).cuda())
    e, r = e.cpu().numpy(), r.cpu().numpy()
    for i, (wire, tx_errors) in enumerate(cases):
        eo, ro = oracle.prbs_detector_run(k, wire)
        assert np.array_equal(e[i], eo) and np.array_equal(r[i], ro)
        if not errors_inside_reload(k, tx_errors, r[i]):
            check_case(tx_errors, e[i], r[i])


def test_prbs31_beyond_2_pow_35_bits(gpu, oracle):
    """4 GiB of packed PRBS-31 in one call: loopback is clean, far words match the oracle."""
    nbits = (1 << 35) + 77
    p = gpu.PRBS(31)
    buf = p.generate(nbits, first_bit=5)
    assert gpu.PRBSErrorDetector(31).count_errors(buf, nbits, first_bit=5) == 0
    rng = np.random.default_rng(1)
    for w in [0, (nbits // 64) - 40] + [int(x) for x in rng.integers(0, nbits // 64 - 40, size=10)]:
        got = buf[w: w + 32].cpu().numpy().view(np.uint64)
        exp, _ = oracle.prbs_packed(31, 32 * 64, state=p.state_at(5 + 64 * w), fast=True)
        assert np.array_equal(got, exp), w
    buf[12345] ^= 0x10
    buf[-1] ^= 1
    assert gpu.PRBSErrorDetector(31).count_errors(buf, nbits, first_bit=5) == 2
    del buf
    torch.cuda.empty_cache()


def test_baseline_config3_full_size_with_error_mask(gpu, oracle):
    """BASELINE.json configs[2] at its full size: 1e10 bits of PRBS-31 through generator -> checker, clean and
    with a fixed XOR mask on every (1e6+7)-th bit: exact counts from the phase-known
    checker AND from the self-synchronising detector (isolated errors: each flagged exactly once).
    The generator's whole 1.25 GB output is first held to the ORACLE's sequential stream (prbs.py:32-35, 112-113 restated
    word-parallel: one core, under a second), so the loopback is not only the device agreeing with itself; the
    read-back form of the fill (the one the timed loopback uses) against the same words."""
    nbits = 10_000_000_000
    p = gpu.PRBS(31)
    det = gpu.PRBSErrorDetector(31)
    buf2 = p.generate(nbits)
    exp, s_end = oracle.prbs_packed(31, nbits, fast=True)
    exp_t = torch.from_numpy(exp.view(np.int64))
    assert buf2.numel() == exp_t.numel() == (nbits + 63) // 64
    piece = 1 << 24                                            # 128 MiB of words at a time on the host
    for lo in range(0, buf2.numel(), piece):
        assert torch.equal(buf2[lo: lo + piece].cpu(), exp_t[lo: lo + piece]), f"PRBS-31 fill differs from the oracle in words [{lo}, {lo + piece})"
    assert s_end == p.state_at(nbits)                         # and the LFSR state behind the last bit (the host's jump-ahead)
    hinted = p.generate(nbits, will_read_back=True)
    assert torch.equal(hinted, buf2)
    del hinted, exp, exp_t
    assert det.count_errors(buf2, nbits) == 0
    pos = torch.arange(0, nbits, 1_000_007, dtype=torch.int64, device=buf2.device)
    flip = torch.zeros_like(buf2)
    flip.index_put_((pos // 64,), torch.ones_like(pos) << (pos % 64))      # distinct words
    buf2 ^= flip
    del flip
    assert det.count_errors(buf2, nbits) == pos.numel() == 10_000
    st = det.run_stream(buf2, nbits)
    first_in_reload = 1 if st["reload_clocks"] > 0 else 0                  # bit 0 arrives during the reload out of reset
    assert st["errors_raw"] + 0 >= 10_000 - first_in_reload and st["errors"] in (10_000, 10_000 - first_in_reload)


def test_concurrent_checks_do_not_share_a_counter(gpu, oracle):
    """bbb_prbs_check from several threads and streams on one device at once: every call owns its counter
    (stream-ordered allocation), so the counts do not bleed into each other."""
    import threading
    nbits = 40_000_003
    p = gpu.PRBS(31)
    bufs, want = [], []
    for t in range(4):
        b = p.generate(nbits)
        for i in range(t * 3):                       # t * 3 flipped bits in buffer t
            b[1000 + 17 * i] ^= 1 << (i % 60)
        bufs.append(b)
        want.append(t * 3)
    torch.cuda.synchronize()
    got = [[None] * 6 for _ in range(4)]

    def work(t):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            det = gpu.PRBSErrorDetector(31)
            for r in range(6):
                got[t][r] = det.count_errors(bufs[t], nbits)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert got == [[w] * 6 for w in want]


@pytest.mark.parametrize("k", [7, 20, 31])
def test_read_back_hint_changes_no_bit(gpu, k):
    """bbb_prbs_fill_hint(BBB_PRBS_WILL_READ_BACK): non-temporal stores in the generator -- the same words as bbb_prbs_fill at
    ragged sizes and offsets, and a clean check behind it; an unknown flag is refused."""
    import ctypes as C
    p = gpu.PRBS(k)
    for nbits, first in ((1, 0), (4097, 3), (3_000_001, 64 * 1000 + 5), (50_000_000, 1 << 40)):
        a = p.generate(nbits, first_bit=first)
        b = p.generate(nbits, first_bit=first, will_read_back=True)
        assert torch.equal(a, b)
        assert gpu.PRBSErrorDetector(k).count_errors(b, nbits, first_bit=first) == 0
    buf = torch.zeros(4, dtype=torch.int64, device="cuda")
    l = gpu._lib.lib()
    assert l.bbb_prbs_fill_hint(k, 1, 0, 100, C.c_void_p(buf.data_ptr()), 2, 0, None) == gpu._lib.BBB_EINVAL


def test_fills_after_the_stream_of_an_earlier_fill_is_gone(gpu, oracle):
    """Round 4's crash scenario, kept as a regression although the cache that caused it is gone (round 5 deleted the shared
    region seeds: DESIGN.md 3.5): the transmitter's data bits are generated on an internal stream that dies with its handle; a
    process-wide PRBS state that remembered that stream synchronised with it later.  Here: such a fill is made, its handle
    destroyed, then a dozen fills at other positions and one position from two live streams -- against the oracle.  (The
    scheduler model now reports any use of a destroyed stream: tests/test_sched_model.py.)"""
    import gc
    n = (1 << 24) + 4096
    x = gpu.TX(31, 1, 0, 16, 1, 8)
    with x.stream(n, first_sample=0) as st:
        st.next()
    torch.cuda.synchronize()
    del st, x
    gc.collect()
    p = gpu.PRBS(31)
    for i in range(12):
        first = 1_000_003 * (i + 1)
        got = p.generate(100_000, first_bit=first).cpu().numpy().view(np.uint64)
        exp, _ = oracle.prbs_packed(31, 100_000, state=p.state_at(first), fast=True)
        assert np.array_equal(got, exp), i
    # and a plan shared between two live streams: the second user waits for the seed kernel by the plan's event
    s2 = torch.cuda.Stream()
    a = p.generate(3_000_000, first_bit=77)
    with torch.cuda.stream(s2):
        b = p.generate(3_000_000, first_bit=77)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def _state_at(oracle, k, first):
    """LFSR state after `first` clocks from state 1 (the oracle's own stepping: small offsets only)."""
    _, s = oracle.prbs_packed(k, first, state=1, fast=True)
    return s
