"""GF(2) helpers against the literal known answers of the reference's Rust unit tests, and as a
cross-check of the generators (CPU only)."""
import numpy as np
import pytest

import basebandboard_amd as bbb
from basebandboard_amd import gf2


@pytest.mark.parametrize("bits,poly", [
    ("0000100011000010011", "x^9 + x^5 + 1"),                                   # berlekamp_massey.rs:40-42 (PRBS-9)
    ("00000000101000000100010", "x^11 + x^9 + 1"),                              # :45-47 (PRBS-11)
    ("01000100111000101110110000100011", "x^16 + x^14 + x^13 + x^11 + 1"),      # :50-52
    ("101010100110010000111101100101010011111000110110100010111010101011",
     "x^33 + x^31 + x^29 + x^26 + x^24 + x^22 + x^19 + x^14 + x^8 + x^7 + x^2 + 1"),   # :58-60
    ("10110100101101001011010010110100101101001011010010110100101101010111110101111101011111010111110101111101011111010111110101110010",
     "x^64 + x^62 + x^61 + x + 1"),                                             # :63-65
])
def test_berlekamp_massey_rnghunt_kats(bits, poly):
    assert gf2.poly_str(gf2.berlekamp_massey([int(c) for c in bits])) == poly


def test_recur_rnghunt_kat(golden_gf2):
    g = golden_gf2["test_recur"]                                                # binary_matrix.rs:183-192
    out = gf2.recur(g["nrows"], g["ncols"], [int(w, 16) for w in g["col_words_hex"]], g["x_bits"], g["n"])
    assert out.tolist() == g["out_bits"]


@pytest.mark.parametrize("k", sorted(bbb.TAPS))
def test_prbs_golden_bits_have_the_right_minimal_polynomial(golden_prbs, k):
    bits = [int(c) for c in golden_prbs[str(k)]["bits"][: 2 * k + 8]]
    assert gf2.berlekamp_massey(bits) == [k, bbb.TAPS[k], 0]                    # x^k + x^tap + 1 (prbs.py:12-14)


@pytest.mark.parametrize("n", (16, 32, 64, 128, 256))
def test_lutopt_bit0_sequence_has_full_linear_complexity(golden_lutopt, n):
    """What the reference's search checks first (rnghunt.rs:28-39): 2n steps of the recurrence, bit 0
    of each state, Berlekamp-Massey -> a polynomial of degree exactly n."""
    u = bbb.LUTOPT.shipped(n, init=(1 << n) - 1, device=-1)                     # rnghunt starts from all ones (:27)
    bits = [u.state_at(t + 1) & 1 for t in range(2 * n)]
    poly = gf2.berlekamp_massey(bits)
    assert poly[0] == n and poly[-1] == 0
