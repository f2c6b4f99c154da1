// gf2_api.hip -- host-side GF(2) helpers behind the C ABI (no kernels): the parts of the
// reference's Rust crate software/rnghunt that touch this path.
//   berlekamp_massey   software/rnghunt/src/berlekamp_massey.rs:5-31 (minimal polynomial of a bit
//                      sequence; the search tool applies it to bit 0 of 2n recurrence steps,
//                      src/bin/rnghunt.rs:28-36)
//   recur              software/rnghunt/src/binary_matrix.rs:68-76 (bit 0 of successive A x)
// Used here to cross-check the generators: PRBS-k bits must have minimal polynomial
// x^k + x^tap + 1, bit 0 of a LUTOPT-n state sequence must have linear complexity n.
#include "bbb_common.hpp"
#include "gf2.hpp"

#include <vector>

using namespace bbb;

extern "C" {

// Minimal LFSR (connection polynomial C(x) = 1 + c1 x + ... + cL x^L with
// s[t] = c1 s[t-1] ^ ... ^ cL s[t-L]) of bits[0..n).  coeffs_out[i] = coefficient of x^i,
// i = 0..*degree (needs n+1 bytes).
int bbb_gf2_berlekamp_massey(const uint8_t *bits, uint64_t n, uint8_t *coeffs_out, int64_t *degree) {
    if ((n && !bits) || !coeffs_out || !degree) return fail(BBB_EINVAL, "null argument");
    std::vector<uint8_t> c(n + 1, 0), b(n + 1, 0), t;
    c[0] = b[0] = 1;
    uint64_t L = 0;
    int64_t m = -1;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t d = bits[i] & 1;                               // discrepancy
        for (uint64_t j = 1; j <= L; j++) d ^= c[j] & bits[i - j];
        if (!d) continue;
        t = c;
        const uint64_t shift = (uint64_t)((int64_t)i - m);
        for (uint64_t j = 0; j + shift <= n; j++) c[j + shift] ^= b[j];
        if (2 * L <= i) {
            L = i + 1 - L;
            m = (int64_t)i;
            b = t;
        }
    }
    for (uint64_t j = 0; j <= L; j++) coeffs_out[j] = c[j];
    *degree = (int64_t)L;
    return BBB_OK;
}

// BinaryMatrix::recur with rnghunt's storage (binary_matrix.rs:15-19): column-major u64 words, the
// first row in the MOST significant bit; x as one bit per byte.  out_bits[s] = bit 0 of A^(s+1) x.
int bbb_gf2_recur(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits, int nsteps,
                  uint8_t *out_bits) {
    if (!col_words || !x_bits || !out_bits) return fail(BBB_EINVAL, "null argument");
    if (nrows != ncols || nrows < 1 || nrows > BBB_MAX_K) return fail(BBB_EINVAL, "matrix must be square, at most 512 wide");
    const int wpc = (nrows + 63) / 64;
    GF2Mat A(nrows);
    for (int c = 0; c < ncols; c++)
        for (int r = 0; r < nrows; r++)
            if ((col_words[(size_t)c * wpc + r / 64] >> (63 - (r % 64))) & 1ull) A.set(r, c);
    uint64_t x[8] = {0};
    for (int c = 0; c < ncols; c++)
        if (x_bits[c] & 1) x[c >> 6] |= 1ull << (c & 63);
    for (int s = 0; s < nsteps; s++) {
        A.matvec(x, x);
        out_bits[s] = (uint8_t)(x[0] & 1ull);
    }
    return BBB_OK;
}

}  // extern "C"
