"""Pulse shaper and TX output kernels vs the oracle and the golden vectors (bit exact)."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_shaper():
    return json.load(open(GOLDEN / "shaper.json"))


def test_reference_shaper_test_on_gpu(gpu, golden_shaper):
    """The reference's test_prbs_shaper (bitshaper.py:112-157): PRBS9 through the beta = 0.5 pulse."""
    t = golden_shaper["test_prbs_shaper"]
    sh = gpu.PRBSShaper(gpu.PRBS(9), 0, [t["coeffs"]])
    shaped = sh.generate(t["nsamples"]).cpu().numpy()
    assert shaped[73:].tolist() == t["shaped_from_73"]


@pytest.mark.parametrize("name", ["prbs31_set10", "prbs7_set31"])
def test_shaper_golden_sets(gpu, golden_shaper, name):
    t = golden_shaper[name]
    sh = gpu.PRBSShaper.from_rcf(gpu.PRBS(t["k"]), t["set"], golden_shaper["betas"])
    assert sh.generate(t["nsamples"]).cpu().numpy()[73:].tolist() == t["shaped_from_73"]


@pytest.mark.parametrize("k,setsel,n,first", [(31, 0, 1, 0), (31, 5, 7, 0), (9, 31, 8, 11), (7, 16, 100_003, 0),
                                              (20, 3, 400_000, 123_457), (23, 20, 1_000_001, 10**12 + 3), (15, 8, 64, 17),
                                              # windows that need NO data bit yet (the bit buffer is empty) or end inside the
                                              # first bits: the kernel must not read past the bits it was given
                                              (31, 10, 17, 0), (31, 10, 8, 8), (31, 10, 16, 1), (9, 4, 33, 0), (7, 2, 25, 16)])
def test_shaper_matches_oracle(gpu, oracle, golden_shaper, k, setsel, n, first):
    sh = gpu.PRBSShaper.from_rcf(gpu.PRBS(k), setsel, golden_shaper["betas"])
    got = sh.generate(n, first_sample=first).cpu().numpy()
    if first > 10**9:                      # the oracle walks the PRBS sequentially: hand it the jumped state
        bit0 = (first - 17) // 8 - 7       # first data bit the window needs
        p = gpu.PRBS(k)
        exp = oracle.shaper(golden_shaper["rcf_coeffs"][setsel], k, n, first_sample=first - 8 * bit0,
                            prbs_state=p.state_at(bit0))
    else:
        exp = oracle.shaper(golden_shaper["rcf_coeffs"][setsel], k, n, first_sample=first)
    assert np.array_equal(got, exp)


def test_pulse_source_and_rect(gpu, oracle, golden_shaper):
    sh = gpu.PRBSShaper.from_rcf(gpu.Pulser(), 1, [0.25])             # set 1 = the appended rectangular pulse
    got = sh.generate(8 * 700).cpu().numpy()
    assert np.array_equal(got, oracle.shaper(golden_shaper["rect"], 0, 8 * 700, source=1))
    assert (got == 254).sum() == 12


@pytest.mark.parametrize("bit_en,noise_en,nv,src", [(1, 1, 8, 0), (1, 0, 8, 0), (0, 1, 15, 0), (0, 0, 3, 0), (1, 1, 15, 1), (1, 1, 1, 0)])
def test_tx_matches_oracle(gpu, oracle, golden_shaper, bit_en, noise_en, nv, src):
    tx = gpu.TX(31, bit_en, src, 16, noise_en, nv)
    n, first = 300_007, 40_001
    got = tx.generate(n, first_sample=first).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp = oracle.tx(m, 1, golden_shaper["rcf_coeffs"][16], 31, n, first_sample=first, source=src, bit_en=bit_en,
                    noise_en=noise_en, noise_var=nv, warmup=16)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("n,first", [(17, 0), (8, 8), (16, 1), (40, 0), (26, 3)])
def test_tx_short_windows_at_the_start(gpu, oracle, golden_shaper, n, first):
    """The first call on a fresh handle with a window before / around the first data bit."""
    tx = gpu.TX(31, 1, 0, 16, 1, 8)
    got = tx.generate(n, first_sample=first).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp = oracle.tx(m, 1, golden_shaper["rcf_coeffs"][16], 31, n, first_sample=first, noise_var=8, warmup=16)
    assert np.array_equal(got, exp)


def test_tx_large_and_statistics(gpu, oracle, golden_shaper):
    """2^26 samples: prefix/suffix against the oracle; the noise-only stream has the CLT moments
    scaled by noise_var; bits-only equals the shaper."""
    n = 1 << 26
    tx = gpu.TX(31, 1, 0, 10, 1, 8)
    x = tx.generate(n)
    m = oracle.Lutopt(path=oracle.data_path(256))
    c = golden_shaper["rcf_coeffs"][10]
    assert np.array_equal(x[:100_000].cpu().numpy(), oracle.tx(m, 1, c, 31, 100_000, noise_var=8))
    noise_only = gpu.TX(31, 0, 0, 10, 1, 8).generate(1 << 22).cpu().numpy().astype(np.float64)
    assert abs(noise_only.mean()) < 0.5 and abs(noise_only.var() / (64 * 64) - 1) < 0.02
    bits_only = gpu.TX(31, 1, 0, 10, 0, 8).generate(50_000)
    assert torch.equal(bits_only, gpu.PRBSShaper.from_rcf(gpu.PRBS(31), 10, golden_shaper["betas"]).generate(50_000))


def test_tx_errors(gpu):
    with pytest.raises(ValueError, match="invalid for PRBS"):
        gpu.TX(12, 1, 0, 0, 1, 8)
    with pytest.raises(ValueError):
        gpu.TX(31, 1, 0, 0, 1, 16)
    with pytest.raises(ValueError):
        gpu.TX(31, 1, 0, 40, 1, 8)


def test_rx_slice_matches_oracle(gpu, oracle):
    rng = np.random.default_rng(5)
    x = rng.integers(-2048, 2048, size=100_003).astype(np.int16)
    x[::7] = 0                                           # the threshold value itself: >= vs >
    xt = torch.from_numpy(x).cuda()
    for spb, delay, stride, strict in ((8, 0, None, False), (8, 5, None, False), (16, 16, 4, False), (4, 1, None, True), (1, 0, None, False)):
        rx = gpu.RX(31, spb, delay)
        bits, nbits = rx.slice(xt, stride=stride, strict=strict)
        exp = oracle.rx_slice(x, stride or spb, delay, strict)
        assert nbits == len(exp)
        got = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:nbits]
        assert np.array_equal(got, exp)
    with pytest.raises(ValueError, match="invalid for PRBS"):
        gpu.RX(10, 8, 0)
    with pytest.raises(ValueError):
        gpu.RX(31, 12, 0)


@pytest.mark.parametrize("stride", [1, 2, 3, 4, 8, 16])
def test_rx_slice_sizes_and_phases(gpu, oracle, stride):
    """Every slicer form (byte per lane for strides 1, 2, 4; ballot per word otherwise; their unrolled and tail paths) at
    ragged sizes: bit counts around byte and word boundaries, odd phases (unaligned 16-byte loads), both thresholds, and
    zeroed padding behind the last bit."""
    rng = np.random.default_rng(stride)
    big = rng.integers(-3, 4, size=1_200_000).astype(np.int16)          # many zeros: >= and > differ often
    for n in (1, 2, 7, 8, 9, 63, 64, 65, 127, 129, 511, 513, 4096, 70_001, 1_200_000):
        for first, delay in ((0, 0), (1, 0), (5, 1), (16, stride if stride <= 16 else 0), (1001, stride - 1)):
            x = big[:n]
            xt = torch.from_numpy(x.copy()).cuda()
            for strict in (False, True):
                rx = gpu.RX(31, 16, delay)
                bits, nbits = rx.slice(xt, first_sample=first, stride=stride, strict=strict)
                exp = oracle.rx_slice(x[first:], stride, delay, strict) if first < n else np.zeros(0, np.uint8)
                assert nbits == len(exp), (n, first, delay)
                raw = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")
                assert np.array_equal(raw[:nbits], exp), (n, first, delay, strict)
                assert not raw[nbits:].any()


def test_tx_rx_loopback(gpu, oracle, golden_shaper):
    """TX -> slicer at the pulse centre -> PRBS checker.  Noise-free: the decided bits ARE the PRBS
    (0 errors); with noise the count equals the oracle's count over its own TX samples."""
    n = 8 * 200_000
    centre = 17 + 32                                     # impulse of bit m at 8m+4, pulse peak at tap 32, delay 13
    clean = gpu.TX(31, 1, 0, 16, 0, 0).generate(n)
    rx = gpu.RX(31, 8, centre % 8)
    first = centre - centre % 8                          # bit 0 is decided at sample `centre`
    errs, nbits = rx.count_errors(clean, first_sample=first)
    assert errs == 0 and nbits >= 199_990
    noisy_tx = gpu.TX(31, 1, 0, 16, 1, 15)
    noisy = noisy_tx.generate(n)
    errs, nbits = rx.count_errors(noisy, first_sample=first)
    m = oracle.Lutopt(path=oracle.data_path(256))
    ref = oracle.tx(m, 1, golden_shaper["rcf_coeffs"][16], 31, n, noise_var=15)
    rbits = oracle.rx_slice(ref, 8, centre)
    pbits, _ = oracle.prbs_bits(31, len(rbits))
    assert nbits == len(rbits) and errs == int((rbits != pbits).sum()) and errs > 0


def test_rx_detect_and_phase_search(gpu, oracle, golden_shaper):
    """The reference's receiver at scale: TX waveform -> slicer at every `sample_delay` setting -> exact
    self-synchronising detector.  Each phase's totals equal the oracle's slicer + serial detector on the
    same samples; the pulse centre is found, and there the noise-free stream has no error after lock."""
    n = 8 * 150_000
    centre = (17 + 32) % 8
    for nv in (0, 15):
        x = gpu.TX(31, 1, 0, 16, 1 if nv else 0, nv).generate(n)
        rx = gpu.RX(31, 8, centre)
        stats, best = rx.phase_search(x)
        xs = x.cpu().numpy()
        for p in range(8):
            bits = oracle.rx_slice(xs, 8, p)
            w = np.packbits(bits, bitorder="little")
            w = np.concatenate([w, np.zeros((-len(w)) % 8, dtype=np.uint8)]).view(np.uint64)
            _, _, st = oracle.prbs_detector_packed(31, w, len(bits))
            assert stats[p]["bits"] == len(bits)
            for name in ("errors", "errors_raw", "reload_clocks", "resyncs"):
                assert stats[p][name] == st[name], (p, name)
        assert abs(best - centre) <= 1 or abs(best - centre) >= 7
        d = rx.detect(x)
        assert d["errors"] == stats[centre]["errors"]
        if nv == 0:
            assert d["errors"] == 0 and d["resyncs"] <= 3
        else:
            assert d["errors"] > 0
