"""GF(2) toolkit -- the reference's Rust crate software/rnghunt behind the C ABI: Berlekamp-Massey
(berlekamp_massey.rs), the matrix recurrence (binary_matrix.rs `recur`), polynomial modexp and the
primitivity test (binary_polynomial.rs), the search tool's acceptance test and output format
(src/bin/rnghunt.rs) and the search itself, which runs on the GPU (`search`).  Everything else is
host-side algebra in libbbb_hip.so."""
import ctypes as C

import numpy as np

from . import _lib


def berlekamp_massey(bits):
    """Exponents (descending) of the minimal polynomial of a 0/1 sequence, e.g. [9, 5, 0]."""
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(len(b) + 1, dtype=np.uint8)
    deg = C.c_int64()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    _lib.check(_lib.lib().bbb_gf2_berlekamp_massey(p(b), len(b), p(out), C.byref(deg)), "bbb_gf2_berlekamp_massey")
    return [i for i in range(deg.value, -1, -1) if out[i]]


def poly_str(exponents):
    """Format as the reference prints polynomials (binary_polynomial.rs Display): 'x^9 + x^5 + 1'."""
    if not exponents:
        return "0"
    return " + ".join("1" if e == 0 else "x" if e == 1 else f"x^{e}" for e in exponents)


def recur(nrows, ncols, col_words, x_bits, nsteps):
    """BinaryMatrix::recur on rnghunt's column-major, MSbit-first words (binary_matrix.rs:68-76)."""
    cw = np.array(col_words, dtype=np.uint64)
    xb = np.array(x_bits, dtype=np.uint8)
    out = np.zeros(nsteps, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    _lib.check(_lib.lib().bbb_gf2_recur(nrows, ncols, cw.ctypes.data_as(C.POINTER(C.c_uint64)), p(xb), nsteps, p(out)),
               "bbb_gf2_recur")
    return out


def dot(nrows, ncols, col_words, x_bits):
    """BinaryMatrix::dot on rnghunt's column-major, MSbit-first words (binary_matrix.rs:52-63)."""
    cw = np.array(col_words, dtype=np.uint64)
    xb = np.array(x_bits, dtype=np.uint8)
    out = np.zeros(nrows, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    _lib.check(_lib.lib().bbb_gf2_dot(nrows, ncols, cw.ctypes.data_as(C.POINTER(C.c_uint64)), p(xb), p(out)), "bbb_gf2_dot")
    return out


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def poly_is_primitive(coeffs):
    """BinaryPolynomial.from_coefficients(coeffs).is_primitive() -- coeffs = [c0, ..., cN] for
    c0 x^N + ... + cN (binary_polynomial.rs:48-53, :178-216)."""
    c = np.ascontiguousarray(coeffs, dtype=np.uint8)
    r = C.c_int()
    _lib.check(_lib.lib().bbb_gf2_poly_is_primitive(_u8(c), len(c), C.byref(r)), "bbb_gf2_poly_is_primitive")
    return bool(r.value)


def poly_modexp(coeffs, e):
    """x^e mod p as a coefficient list in the same order (binary_polynomial.rs:135-163)."""
    c = np.ascontiguousarray(coeffs, dtype=np.uint8)
    nw = max(1, (int(e).bit_length() + 63) // 64)
    ew = np.array([(int(e) >> (64 * i)) & (2 ** 64 - 1) for i in range(nw)], dtype=np.uint64)
    out = np.zeros(len(c), dtype=np.uint8)
    _lib.check(_lib.lib().bbb_gf2_poly_modexp(_u8(c), len(c), ew.ctypes.data_as(C.POINTER(C.c_uint64)), nw, _u8(out)),
               "bbb_gf2_poly_modexp")
    return out.tolist()


def _flat(rows):
    k = len(rows)
    taps = np.array([c for r in rows for c in r], dtype=np.uint16)
    off = np.zeros(k + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(r) for r in rows])
    return k, taps, off


def _rows(k, taps, off):
    return [[int(x) for x in taps[off[r]:off[r + 1]]] for r in range(k)]


def lutopt_charpoly(rows):
    """(coefficients highest power first, degree) of the polynomial rnghunt's search examines for the
    recurrence `rows` (tap lists): rnghunt.rs:27-38."""
    k, taps, off = _flat(rows)
    out = np.zeros(2 * k + 1, dtype=np.uint8)
    deg = C.c_int()
    _lib.check(_lib.lib().bbb_lutopt_charpoly(k, taps.ctypes.data_as(C.POINTER(C.c_uint16)), off.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              _u8(out), C.byref(deg)), "bbb_lutopt_charpoly")
    return out[:deg.value + 1].tolist(), deg.value


def is_full_period(rows):
    """The search's acceptance test (rnghunt.rs:40-46): degree k and primitive, i.e. period 2^k - 1."""
    k, taps, off = _flat(rows)
    r = C.c_int()
    _lib.check(_lib.lib().bbb_lutopt_is_full_period(k, taps.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                    off.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(r)), "bbb_lutopt_is_full_period")
    return bool(r.value)


def save_matrix(path, rows):
    """Write the reference's 0/1 text matrix (rnghunt.rs:51-53), readable by recurrences.load_packed."""
    k, taps, off = _flat(rows)
    _lib.check(_lib.lib().bbb_lutopt_save_matrix_file(str(path).encode(), k, taps.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                      off.ctypes.data_as(C.POINTER(C.c_uint32))), "bbb_lutopt_save_matrix_file")


def search_candidate(k, seed, candidate):
    """Candidate matrix number `candidate` of `seed` as tap lists (csrc/search_rng.hpp)."""
    taps = np.zeros(4 * k, dtype=np.uint16)
    off = np.zeros(k + 1, dtype=np.uint32)
    _lib.check(_lib.lib().bbb_lutopt_search_candidate(k, int(seed), int(candidate), taps.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                      off.ctypes.data_as(C.POINTER(C.c_uint32))), "bbb_lutopt_search_candidate")
    return _rows(k, taps, off)


def search(k, seed=1, first=0, count=1 << 16, device=0):
    """rnghunt's search loop on the GPU over candidates first .. first+count-1.  Returns
    (index of the smallest accepted candidate or None, its tap lists or None, stats dict)."""
    import torch
    taps = np.zeros(4 * k, dtype=np.uint16)
    off = np.zeros(k + 1, dtype=np.uint32)
    found = C.c_uint64()
    st = _lib.SearchStats()
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(_lib.lib().bbb_lutopt_search(k, int(seed), int(first), int(count), C.byref(found),
                                            taps.ctypes.data_as(C.POINTER(C.c_uint16)), off.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            C.byref(st), int(device), stream), "bbb_lutopt_search")
    stats = {n: int(getattr(st, n)) for n, _ in _lib.SearchStats._fields_}
    if found.value == 2 ** 64 - 1:
        return None, None, stats
    return int(found.value), _rows(k, taps, off), stats
