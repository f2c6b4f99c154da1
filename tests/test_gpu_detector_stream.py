"""PRBSErrorDetector over one long packed stream, executed in parallel chunks with state hand-off
(bbb_prbs_detector_stream), against the serial restatement of gateware/bbb/prbs.py:61-99.  Every
err / reload bit and every total must be identical, whatever the chunking and however bad the
speculative starts are."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KS = (7, 9, 11, 15, 20, 23, 31)
TOTALS = ("errors", "errors_raw", "reload_clocks", "resyncs")


def corrupt(words, nbits, ber, seed, bursts=()):
    """XOR Bernoulli(ber) errors and the given (start, length) bursts into a packed numpy stream."""
    rng = np.random.default_rng(seed)
    w = words.copy()
    n = int(rng.binomial(nbits, ber))
    pos = np.unique(rng.integers(0, nbits, size=n))
    for a, l in bursts:
        pos = np.union1d(pos, np.arange(a, min(nbits, a + l)))
    pos = pos.astype(np.uint64)
    np.bitwise_xor.at(w, (pos // 64).astype(np.int64), np.uint64(1) << (pos % np.uint64(64)))
    return w


def run_both(gpu, oracle, k, words, nbits, **kw):
    det = gpu.PRBSErrorDetector(k)
    t = torch.from_numpy(words.view(np.int64)).cuda()
    got = det.run_stream(t, nbits, want_err=True, want_reload=True, **kw)
    e, r, st = oracle.prbs_detector_packed(k, words, nbits)
    assert np.array_equal(got["err"].cpu().numpy().view(np.uint64), e)
    assert np.array_equal(got["reload"].cpu().numpy().view(np.uint64), r)
    for name in TOTALS:
        assert got[name] == st[name], name
    assert got["bits"] == nbits
    return got


@pytest.mark.parametrize("k", KS)
def test_clean_stream(gpu, oracle, k):
    """The reference's preamble: after the reload out of reset (k + k//2 clocks; the junk that the half-loaded
    LFSR leaves in the error history can trigger a second one) nothing is flagged."""
    nbits = 300_001
    words = gpu.PRBS(k).generate(nbits, first_bit=5).cpu().numpy().view(np.uint64)
    got = run_both(gpu, oracle, k, words, nbits)
    assert got["errors"] == 0 and 1 <= got["resyncs"] <= 3 and got["chunks_rerun"] == 0


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("ber", (1e-4, 2e-2))
def test_reference_test_protocol_at_scale(gpu, oracle, k, ber):
    """prbs.py:124-163 on 2e6 clocks: Bernoulli errors, a 3k burst in the middle (forces a resync), then
    clean; `errors` (err while reload == 0) is what the reference asserts equal to the injected errors."""
    nbits = 2_000_000
    words = gpu.PRBS(k).generate(nbits).cpu().numpy().view(np.uint64)
    bad = corrupt(words, nbits, ber, seed=k, bursts=[(nbits // 2, 3 * k)])
    got = run_both(gpu, oracle, k, bad, nbits)
    assert got["resyncs"] >= 2


@pytest.mark.parametrize("chunk_bits,warm_bits", [(64, 64), (128, 64), (640, 128), (4096, 64), (4096, 1024), (65536, 256),
                                                  (40960, 1024), (57344, 1024), (65536, 1024)])      # (round 5: the fused kernel's longer chunks)
def test_chunking_never_changes_the_result(gpu, oracle, chunk_bits, warm_bits):
    """Short warm-ups make many speculative starts wrong: the verify / re-run passes must repair all of them."""
    k, nbits = 31, 700_003
    words = gpu.PRBS(k).generate(nbits).cpu().numpy().view(np.uint64)
    bad = corrupt(words, nbits, 1e-2, seed=chunk_bits + warm_bits, bursts=[(1000, 200), (350_000, 93), (699_000, 2000)])
    got = run_both(gpu, oracle, k, bad, nbits, chunk_bits=chunk_bits, warm_bits=warm_bits)
    if warm_bits <= 64:
        assert got["chunks_rerun"] > 0


@pytest.mark.parametrize("nbits", (1, 2, 63, 64, 65, 127, 128, 4095, 4096, 4097, 12_345))
def test_ragged_lengths(gpu, oracle, nbits):
    k = 9
    words = gpu.PRBS(k).generate(nbits + 64).cpu().numpy().view(np.uint64)[: (nbits + 63) // 64].copy()
    run_both(gpu, oracle, k, corrupt(words, nbits, 0.01, seed=nbits), nbits, chunk_bits=128, warm_bits=64)


@pytest.mark.parametrize("k", (7, 31))
def test_noise_input_still_exact(gpu, oracle, k):
    """Random bits: the detector resynchronises for ever; the chunked run must still be the serial one."""
    nbits = 400_000
    rng = np.random.default_rng(99 + k)
    words = rng.integers(0, 2**64, size=(nbits + 63) // 64, dtype=np.uint64)
    got = run_both(gpu, oracle, k, words, nbits, chunk_bits=1024, warm_bits=128)
    assert got["resyncs"] > 100


def test_serial_guard_path(gpu, oracle):
    """All-zero input after a PRBS prefix with no warm-up at all: whatever the repair passes do, the
    result must be exact (exercises the many-pass / serial continuation logic)."""
    k, nbits = 15, 200_000
    words = gpu.PRBS(k).generate(nbits).cpu().numpy().view(np.uint64).copy()
    words[100:] = 0
    run_both(gpu, oracle, k, words, nbits, chunk_bits=64, warm_bits=64)


def test_empty_and_errors(gpu):
    det = gpu.PRBSErrorDetector(7)
    t = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert det.run_stream(t, 0)["bits"] == 0
    with pytest.raises(ValueError):
        det.run_stream(t, 10, chunk_bits=100)
    with pytest.raises(ValueError):
        det.run_stream(t, 1000)


def test_full_size_loopback(gpu):
    """1e9 bits of PRBS-31 through generator -> detector: one reload out of reset, no error, all chunks consistent."""
    nbits = 1_000_000_000
    buf = gpu.PRBS(31).generate(nbits)
    got = gpu.PRBSErrorDetector(31).run_stream(buf, nbits)
    assert got["errors"] == 0 and 1 <= got["resyncs"] <= 3 and got["reload_clocks"] >= 31 + 15 and got["chunks_rerun"] == 0
    base = got["resyncs"]
    # one flipped bit far inside: the detector's LFSR runs on its own feedback, so exactly ONE flagged clock
    buf[5_000_000] ^= 1 << 17
    got = gpu.PRBSErrorDetector(31).run_stream(buf, nbits)
    assert got["errors"] == 1 and got["resyncs"] == base
