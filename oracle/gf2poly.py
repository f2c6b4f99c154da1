"""GF(2) polynomial checks restated with Python integers -- TEST INFRASTRUCTURE ONLY (see oracle/bbb_oracle.h).

Restates, for the parity tests of the GF(2) toolkit and of the GPU recurrence search:
  modmult / modexp / is_primitive   software/rnghunt/src/binary_polynomial.rs:87-216
  berlekamp_massey                  software/rnghunt/src/berlekamp_massey.rs:5-31
  the search's acceptance test      software/rnghunt/src/bin/rnghunt.rs:23-47
  the candidate matrices            basebandboard_amd/csrc/search_rng.hpp (build-defined: the reference's
                                    generator is unseeded, binary_matrix.rs:84)
Pinned by the literal known answers of binary_polynomial.rs:250-371 and berlekamp_massey.rs:36-65
(tests/test_gf2.py) and by the reference's own found matrices, which must all be accepted.
A polynomial is a Python int, bit i = coefficient of x^i.  The prime factors of 2^n - 1 are derived here
(tools/make_factors.py's derivation, not the product's data file).
"""
import importlib.util
import pathlib

_mf = None


def _factors_mod():
    global _mf
    if _mf is None:
        p = pathlib.Path(__file__).resolve().parent.parent / "tools" / "make_factors.py"
        spec = importlib.util.spec_from_file_location("make_factors", p)
        _mf = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mf)
    return _mf


def from_coefficients(coeffs):
    """[c0, c1, ..., cN] = c0 x^N + ... + cN (BinaryPolynomial::from_coefficients, :48-53)."""
    n = len(coeffs)
    return sum((int(c) & 1) << (n - 1 - i) for i, c in enumerate(coeffs))


def degree(p):
    return p.bit_length() - 1


def modmult(f, g, p):
    """f*g mod p by shift and add (:87-133)."""
    n = degree(p)
    r = 0
    while f:
        if f & 1:
            r ^= g
        f >>= 1
        g <<= 1
        if (g >> n) & 1:
            g ^= p
    return r


def modexp(p, e):
    """x^e mod p (:135-163)."""
    if e == 0:
        return 1
    n = degree(p)
    if n == 0:
        return 0
    if n == 1:
        return p & 1
    f = 2
    for b in range(e.bit_length() - 2, -1, -1):
        f = modmult(f, f, p)
        if (e >> b) & 1:
            f <<= 1
            if (f >> n) & 1:
                f ^= p
    return f


def is_primitive(p):
    """:178-216"""
    n = degree(p)
    if n == -1:
        return True
    if not p & 1:
        return False
    if bin(p).count("1") % 2 != 1:
        return False
    r = 2 ** n - 1
    if modexp(p, r) != 1:
        return False
    for q in _factors_mod().primes_of(n):
        if modexp(p, r // q) == 1:
            return False
    return True


def berlekamp_massey(bits):
    """Connection polynomial C (bit j = c_j, c_0 = 1) and linear complexity L of a 0/1 sequence."""
    c, b, L, m = 1, 1, 0, -1
    for i, s in enumerate(bits):
        d = s & 1
        for j in range(1, L + 1):
            d ^= ((c >> j) & 1) & bits[i - j]
        if not d:
            continue
        t = c
        c ^= b << (i - m)
        if 2 * L <= i:
            L, m, b = i + 1 - L, i, t
    return c, L


def lutopt_charpoly(rows):
    """The polynomial rnghunt examines for a recurrence given as tap lists (rnghunt.rs:27-38): bit 0 of 2k
    successive states from all-ones, reversed, Berlekamp-Massey; its index 0 is the leading coefficient, so
    the returned int is the reciprocal of the connection polynomial.  Returns (poly, degree)."""
    k = len(rows)
    x = [1] * k
    seq = []
    for _ in range(2 * k):
        x = [sum(x[c] for c in row) & 1 for row in rows]
        seq.append(x[0])
    seq.reverse()
    c, L = berlekamp_massey(seq)
    p = sum(((c >> i) & 1) << (L - i) for i in range(L + 1))
    return p, L


def is_full_period(rows):
    p, L = lutopt_charpoly(rows)
    return L == len(rows) and is_primitive(p)


# ---- candidate matrices (csrc/search_rng.hpp) --------------------------------------------------
_M = 2 ** 64 - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & _M
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & _M
    return z ^ (z >> 31)


def _hash(seed, cand, ctr):
    return _mix64((_mix64((seed + 0x9e3779b97f4a7c15 * (cand + 1)) & _M) + ctr) & _M) >> 32


def search_candidate(k, seed, cand):
    w = [3 if _hash(seed, cand, r) & 7 == 0 else 4 for r in range(k)]
    P = [0]
    for x in w:
        P.append(P[-1] + x)
    rounds = (P[k] + k - 1) // k
    perm = []
    for j in range(rounds):
        keys = sorted(((_hash(seed, cand, 0x10000 + j * k + c) & ~1023) | c) for c in range(k))
        perm.append([key & 1023 for key in keys])
    for j in range(1, rounds):
        edge = j * k
        rs = [r for r in range(k) if P[r] < edge < P[r] + w[r]]
        if not rs:
            continue
        r = rs[0]
        ntail, h = edge - P[r], P[r] + w[r] - edge
        tail = perm[j - 1][k - ntail:]
        t = 0
        for hp in range(h):
            while perm[j][hp] in tail:
                perm[j][hp], perm[j][h + t] = perm[j][h + t], perm[j][hp]
                t += 1
    stream = [c for pj in perm for c in pj]
    return [stream[P[r]:P[r] + w[r]] for r in range(k)]
