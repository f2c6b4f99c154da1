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
                                              (20, 3, 400_000, 123_457), (23, 20, 1_000_001, 10**12 + 3), (15, 8, 64, 17)])
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
