"""GF(2) helpers -- the pieces of the reference's Rust crate software/rnghunt this path leans on:
Berlekamp-Massey (berlekamp_massey.rs) and the matrix recurrence (binary_matrix.rs `recur`).
Host-side algebra in libbbb_hip.so; used to cross-check the generators."""
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
