"""Pulse shaper / TX restatement (oracle) and the host-side coefficient tables against the golden
vectors derived from the reference's own test model (bitshaper.py:143-155, scipy lfilter)."""
import json

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def golden_shaper():
    return json.load(open(GOLDEN / "shaper.json"))


def test_rcf_coefficient_tables(golden_shaper):
    """PRBSShaper.from_rcf tables for the 32 roll-offs TX uses (tx.py:54, bitshaper.py:97-109)."""
    from basebandboard_amd.bitshaper import PRBSShaper, Pulser, rcf_coefficients
    for beta, c in zip(golden_shaper["betas"], golden_shaper["rcf_coeffs"]):
        assert rcf_coefficients(beta) == c
    sh = PRBSShaper.from_rcf(Pulser(), 0, golden_shaper["betas"])
    assert len(sh.coefficients) == 32                                  # 32 roll-offs: no rect set appended
    sh = PRBSShaper.from_rcf(Pulser(), 1, [0.5])
    assert sh.coefficients[1] == golden_shaper["rect"]                 # fewer than 32: rect appended (bitshaper.py:107-108)
    assert sh.coefficients[0] == golden_shaper["test_prbs_shaper"]["coeffs"]
    with pytest.raises(ValueError):
        PRBSShaper(Pulser(), 3, [[0] * 64])
    with pytest.raises(ValueError):
        PRBSShaper(Pulser(), 0, [[300] * 64]).generate  # noqa: B018  (range is checked when the set is used)
        from basebandboard_amd.bitshaper import _cfg
        _cfg([300] * 64, Pulser())


def test_oracle_shaper_matches_reference_test_model(oracle, golden_shaper):
    t = golden_shaper["test_prbs_shaper"]
    s = oracle.shaper(t["coeffs"], t["k"], t["nsamples"])
    assert s[73:].tolist() == t["shaped_from_73"]                      # bitshaper.py:155
    for name in ("prbs31_set10", "prbs7_set31"):
        t = golden_shaper[name]
        s = oracle.shaper(golden_shaper["rcf_coeffs"][t["set"]], t["k"], t["nsamples"])
        assert s[73:].tolist() == t["shaped_from_73"]
    t = golden_shaper["prbs15_rect"]
    assert oracle.shaper(golden_shaper["rect"], 15, t["nsamples"])[73:].tolist() == t["shaped_from_73"]


def test_oracle_shaper_structure(oracle, golden_shaper):
    c = golden_shaper["rcf_coeffs"][12]
    whole = oracle.shaper(c, 23, 6000)
    assert np.array_equal(oracle.shaper(c, 23, 1234, first_sample=3210), whole[3210:3210 + 1234])
    # rectangular pulse: +-254 for 4 samples in the middle of each bit period, 0 elsewhere
    r = oracle.shaper(golden_shaper["rect"], 9, 8 * 60)
    bits, _ = oracle.prbs_bits(9, 60)
    for m in range(8, 50):
        seg = r[8 * m + 17 + 30: 8 * m + 17 + 34]      # tap j of impulse 8m+4 lands on sample 8m+4+j+13
        assert set(seg.tolist()) == {254 if bits[m] else -254}
    # pulse source: one positive pulse every 256 bit periods on a -1 background
    p = oracle.shaper(golden_shaper["rect"], 0, 8 * 600, source=1)
    assert p.max() == 254 and (p == 254).sum() == 4 * 3 and p.min() == -254
    with pytest.raises(ValueError, match="invalid for PRBS"):
        oracle.shaper(c, 8, 16)


def test_oracle_tx_combination(oracle, golden_shaper):
    """TX.x = wrap12(bits + wrap12(g * noise_var)) with the enables (tx.py:65-81)."""
    m = oracle.Lutopt(path=oracle.data_path(256))
    c = golden_shaper["rcf_coeffs"][16]
    n = 3000
    shaped = oracle.shaper(c, 31, n).astype(np.int64)
    g = m.awgn(1, 16, n).astype(np.int64)
    wrap = lambda v: ((v + 2048) % 4096) - 2048  # noqa: E731
    assert np.array_equal(oracle.tx(m, 1, c, 31, n, noise_var=8), wrap(shaped + wrap(g * 8)))
    assert np.array_equal(oracle.tx(m, 1, c, 31, n, noise_en=0), shaped)
    assert np.array_equal(oracle.tx(m, 1, c, 31, n, bit_en=0, noise_var=15), wrap(g * 15))
    assert not oracle.tx(m, 1, c, 31, n, bit_en=0, noise_en=0).any()
    big = [255] * 64                                                   # sums reach 8*255: noise pushes past 2047 -> wrap
    x = oracle.tx(m, 1, big, 31, 20000, noise_var=15)
    assert x.min() >= -2048 and x.max() <= 2047
