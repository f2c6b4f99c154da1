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


def test_dot_rnghunt_kat(golden_gf2):
    g = golden_gf2["test_dot"]                                                  # binary_matrix.rs:133-180 (64 x 128)
    out = gf2.dot(g["nrows"], g["ncols"], [int(w, 16) for w in g["col_words_hex"]], g["x_bits"])
    assert out.tolist() == g["out_bits"]
    r = golden_gf2["test_recur"]                                                # recur = repeated dot, first bit
    x, bits = r["x_bits"], []
    for _ in range(r["n"]):
        x = gf2.dot(r["nrows"], r["ncols"], [int(w, 16) for w in r["col_words_hex"]], x).tolist()
        bits.append(x[0])
    assert bits == r["out_bits"]


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


# ---- polynomial arithmetic and the search's acceptance test (binary_polynomial.rs, rnghunt.rs) ----------

P200 = [1] + [0] * 194 + [1, 0, 1, 1, 0, 1]            # x^200 + x^5 + x^3 + x^2 + 1 (binary_polynomial.rs:337-339)
P33 = [1, 1, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1]   # :361-364


def test_is_primitive_rnghunt_kats():
    """binary_polynomial.rs:352-371, literally."""
    assert gf2.poly_is_primitive([1, 1, 0, 0, 1])                       # x^4 + x^3 + 1
    assert not gf2.poly_is_primitive([1, 0, 1, 1, 1])                   # x^4 + x^2 + x + 1 = (x+1)(...)
    assert gf2.poly_is_primitive(P33)
    assert gf2.poly_is_primitive(P200)


def test_modexp_and_check_integer_rnghunt_kats():
    """binary_polynomial.rs:308-350: x^k mod p."""
    big = [1] + [0] * 255                                               # x^255: "degree(p) > k so mod doesn't come into it"
    for k in (0, 1, 2, 20, 100):
        out = gf2.poly_modexp(big, k)
        assert [i for i, c in enumerate(reversed(out)) if c] == [k]
    p = [1, 1, 0, 0, 1]
    assert gf2.poly_modexp(p, 15) == [0, 0, 0, 0, 1]
    for k in range(1, 15):                                              # primitive: no smaller power of x is 1
        assert gf2.poly_modexp(p, k) != [0, 0, 0, 0, 1]
    one = [0] * 200 + [1]
    for k in range(1, 100):
        assert gf2.poly_modexp(P200, k) != one
    assert gf2.poly_modexp(P200, 2 ** 200 - 1) == one


def test_prbs_polynomials_are_primitive():
    """x^k + x^tap + 1 for the seven PRBS orders (gateware/bbb/prbs.py:12-14): maximal length."""
    for k, tap in bbb.TAPS.items():
        c = [0] * (k + 1)
        c[0] = c[k - tap] = c[k] = 1
        assert gf2.poly_is_primitive(c), k


def test_product_arithmetic_equals_the_oracle(tmp_path):
    from oracle import gf2poly as og
    rng = np.random.default_rng(3)
    for n in (4, 16, 33, 64, 200, 256):
        for _ in range(4):
            c = rng.integers(0, 2, size=n + 1).tolist()
            c[0] = 1
            p = og.from_coefficients(c)
            assert gf2.poly_is_primitive(c) == og.is_primitive(p)
            e = int.from_bytes(rng.bytes(n // 8 + 1), "little")
            got = og.from_coefficients(gf2.poly_modexp(c, e))
            assert got == og.modexp(p, e)
    with pytest.raises(bbb._lib.BbbError, match="no factorisation"):
        gf2.poly_is_primitive([1] + [0] * 16 + [1, 1])                  # degree 18: not in the table


@pytest.mark.parametrize("n", (16, 32, 64, 128, 192, 256, 512))
def test_the_references_found_matrices_are_accepted(n):
    """The shipped matrices ARE rnghunt's results (software/rnghunt/matrices/N): its own acceptance test,
    restated, must pass on every one of them; and a spoiled one must fail."""
    from basebandboard_amd import recurrences
    rows = recurrences.load_packed(recurrences.matrix_path(n))
    coeffs, deg = gf2.lutopt_charpoly(rows)
    assert deg == n and coeffs[0] == 1 and coeffs[-1] == 1
    assert gf2.is_full_period(rows)
    bad = [list(r) for r in rows]
    bad[3] = bad[3][:-1]
    assert not gf2.is_full_period(bad) or n == 16


@pytest.mark.parametrize("n", (16, 32, 64, 128))
def test_charpoly_equals_the_oracle(n):
    from oracle import gf2poly as og
    from basebandboard_amd import recurrences
    rows = recurrences.load_packed(recurrences.matrix_path(n))
    coeffs, deg = gf2.lutopt_charpoly(rows)
    p, L = og.lutopt_charpoly(rows)
    assert deg == L and og.from_coefficients(coeffs) == p and og.is_primitive(p)


@pytest.mark.parametrize("k,seed,cand", [(16, 1, 0), (16, 1, 228), (32, 9, 4), (64, 2, 77), (192, 5, 3), (256, 1, 0), (512, 123456789, 2 ** 40)])
def test_search_candidates(k, seed, cand):
    """Candidate matrices: the C++ construction equals the restatement; 3 or 4 distinct taps per row,
    column weights within one of each other (BinaryMatrix::random, binary_matrix.rs:81-101)."""
    from oracle import gf2poly as og
    rows = gf2.search_candidate(k, seed, cand)
    assert rows == og.search_candidate(k, seed, cand)
    assert all(len(r) in (3, 4) and len(set(r)) == len(r) for r in rows)
    colw = np.bincount([c for r in rows for c in r], minlength=k)
    assert colw.max() - colw.min() <= 1
    assert 0.02 < sum(len(r) == 3 for r in rows) / k < 0.3 or k <= 32


def test_candidate_acceptance_equals_the_oracle():
    from oracle import gf2poly as og
    hits = [c for c in range(600) if gf2.is_full_period(gf2.search_candidate(16, 1, c))]
    assert hits == [c for c in range(600) if og.is_full_period(og.search_candidate(16, 1, c))] == [228, 234, 427, 544]


def test_save_matrix_roundtrip(tmp_path):
    from basebandboard_amd import recurrences
    rows = gf2.search_candidate(64, 4, 2)
    gf2.save_matrix(tmp_path / "out", rows)                             # rnghunt writes its result to `out` (rnghunt.rs:46)
    text = (tmp_path / "out").read_text().splitlines()
    assert len(text) == 64 and all(len(l) == 64 and set(l) <= {"0", "1"} for l in text)
    assert [sorted(r) for r in recurrences.load_packed(tmp_path / "out")] == [sorted(r) for r in rows]


def test_host_arithmetic_under_sanitizers(tmp_path):
    """The product's host-side GF(2) headers (gf2.hpp, gf2poly.hpp, search_rng.hpp) under AddressSanitizer +
    UndefinedBehaviorSanitizer, on the reference's polynomial KATs, candidate matrices of every order and a jump."""
    import subprocess
    from conftest import ROOT
    gen = ROOT / "basebandboard_amd" / "csrc" / "gen" / "mersenne_factors.inc"
    if not gen.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "basebandboard_amd" / "csrc"), "gen/mersenne_factors.inc"])
    exe = tmp_path / "san_gf2"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        str(ROOT / "tests" / "san_gf2.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("no sanitizer runtime for g++ here")
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env={"ASAN_OPTIONS": "detect_leaks=0"})
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stderr[-2000:]
