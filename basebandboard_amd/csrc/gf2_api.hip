// gf2_api.hip -- host-side GF(2) helpers behind the C ABI (no kernels): the parts of the
// reference's Rust crate software/rnghunt that touch this path.
//   berlekamp_massey   software/rnghunt/src/berlekamp_massey.rs:5-31 (minimal polynomial of a bit
//                      sequence; the search tool applies it to bit 0 of 2n recurrence steps,
//                      src/bin/rnghunt.rs:28-36)
//   recur              software/rnghunt/src/binary_matrix.rs:68-76 (bit 0 of successive A x)
//   dot                binary_matrix.rs:52-63 (A x)
//   poly_modexp / poly_is_primitive   src/binary_polynomial.rs:135-216 (arithmetic in gf2poly.hpp)
//   lutopt_charpoly / is_full_period  the acceptance test of the search tool, src/bin/rnghunt.rs:27-46
//   lutopt_save_matrix_file           its output format, src/bin/rnghunt.rs:51-53
//   lutopt_search_candidate           candidate c of seed s (search_rng.hpp), as the GPU search draws it
// Used here to cross-check the generators: PRBS-k bits must have minimal polynomial
// x^k + x^tap + 1, bit 0 of a LUTOPT-n state sequence must have linear complexity n.
#include "bbb_common.hpp"
#include "gf2.hpp"
#include "gf2poly.hpp"
#include "search_rng.hpp"

#include <cstdio>
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

// BinaryMatrix::dot (binary_matrix.rs:52-63) on the same storage: y = A x over GF(2), any shape up to 512 x 512.
int bbb_gf2_dot(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits, uint8_t *out_bits) {
    if (!col_words || !x_bits || !out_bits) return fail(BBB_EINVAL, "null argument");
    if (nrows < 1 || ncols < 1 || nrows > BBB_MAX_K || ncols > BBB_MAX_K) return fail(BBB_EINVAL, "shape must be within 512 x 512");
    const int wpc = (nrows + 63) / 64;
    uint64_t acc[8] = {0};
    for (int c = 0; c < ncols; c++)
        if (x_bits[c] & 1)
            for (int w = 0; w < wpc; w++) acc[w] ^= col_words[(size_t)c * wpc + w];
    for (int r = 0; r < nrows; r++) out_bits[r] = (uint8_t)((acc[r / 64] >> (63 - (r % 64))) & 1ull);
    return BBB_OK;
}

// coefficient bytes "highest power first" (BinaryPolynomial::from_coefficients, binary_polynomial.rs:48-53)
static int poly_from_bytes(const uint8_t *coeffs, int ncoeffs, GF2Poly &p) {
    if (!coeffs || ncoeffs < 1 || ncoeffs > 513) return fail(BBB_EINVAL, "1..513 coefficients expected");
    p = GF2Poly();
    for (int i = 0; i < ncoeffs; i++)
        if (coeffs[i] & 1) p.set(ncoeffs - 1 - i);
    return BBB_OK;
}

int bbb_gf2_poly_is_primitive(const uint8_t *coeffs, int ncoeffs, int *result) {
    if (!result) return fail(BBB_EINVAL, "null argument");
    GF2Poly p;
    int rc = poly_from_bytes(coeffs, ncoeffs, p);
    if (rc) return rc;
    const int r = gf2_is_primitive(p);
    if (r < 0) return fail(BBB_EUNSUP, "no factorisation of 2^" + std::to_string(p.degree()) + " - 1 in the table (data/mersenne_factors.txt)");
    *result = r;
    return BBB_OK;
}

int bbb_gf2_poly_modexp(const uint8_t *coeffs, int ncoeffs, const uint64_t *exponent_words, int nwords, uint8_t *out_coeffs) {
    if (!exponent_words || nwords < 1 || !out_coeffs) return fail(BBB_EINVAL, "null argument");
    GF2Poly p;
    int rc = poly_from_bytes(coeffs, ncoeffs, p);
    if (rc) return rc;
    const int n = p.degree();
    if (n < 0) return fail(BBB_EINVAL, "modulus is the zero polynomial");
    const GF2Poly f = gf2_modexp(exponent_words, nwords, p, n);
    for (int i = 0; i < ncoeffs; i++) out_coeffs[i] = (uint8_t)f.bit(ncoeffs - 1 - i);
    return BBB_OK;
}

static int check_taps(int k, const uint16_t *taps, const uint32_t *row_off) {
    if (!taps || !row_off) return fail(BBB_EINVAL, "null argument");
    if (k < 1 || k > BBB_MAX_K) return fail(BBB_EINVAL, "k must be 1..512");
    for (int r = 0; r < k; r++) {
        if (row_off[r + 1] < row_off[r]) return fail(BBB_EINVAL, "row_off must be non-decreasing");
        for (uint32_t q = row_off[r]; q < row_off[r + 1]; q++)
            if (taps[q] >= k) return fail(BBB_EINVAL, "tap index out of range");
    }
    return BBB_OK;
}

int bbb_lutopt_charpoly(int k, const uint16_t *taps, const uint32_t *row_off, uint8_t *coeffs_out, int *degree) {
    if (!coeffs_out || !degree) return fail(BBB_EINVAL, "null argument");
    int rc = check_taps(k, taps, row_off);
    if (rc) return rc;
    GF2Poly P;
    const int L = lutopt_charpoly(k, taps, row_off, P);
    for (int i = 0; i <= 2 * k; i++) coeffs_out[i] = 0;
    for (int i = 0; i <= L; i++) coeffs_out[i] = (uint8_t)P.bit(L - i);             // highest power first; L <= 2k
    *degree = L;
    return BBB_OK;
}

int bbb_lutopt_is_full_period(int k, const uint16_t *taps, const uint32_t *row_off, int *result) {
    if (!result) return fail(BBB_EINVAL, "null argument");
    int rc = check_taps(k, taps, row_off);
    if (rc) return rc;
    GF2Poly P;
    const int L = lutopt_charpoly(k, taps, row_off, P);
    if (L != k) { *result = 0; return BBB_OK; }                                       // rnghunt.rs:40-42
    const int r = gf2_is_primitive(P);
    if (r < 0) return fail(BBB_EUNSUP, "no factorisation of 2^" + std::to_string(k) + " - 1 in the table");
    *result = r;
    return BBB_OK;
}

int bbb_lutopt_save_matrix_file(const char *path, int k, const uint16_t *taps, const uint32_t *row_off) {
    if (!path) return fail(BBB_EINVAL, "null argument");
    int rc = check_taps(k, taps, row_off);
    if (rc) return rc;
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(BBB_EIO, std::string("cannot create ") + path);
    std::string line((size_t)k, '0');
    for (int r = 0; r < k; r++) {
        std::fill(line.begin(), line.end(), '0');
        for (uint32_t q = row_off[r]; q < row_off[r + 1]; q++) line[taps[q]] = line[taps[q]] == '0' ? '1' : '0';
        std::fprintf(f, "%s\n", line.c_str());
    }
    if (std::fclose(f) != 0) return fail(BBB_EIO, std::string("write failed: ") + path);
    return BBB_OK;
}

int bbb_lutopt_search_candidate(int k, uint64_t seed, uint64_t candidate, uint16_t *taps_out, uint32_t *row_off_out) {
    if (!taps_out || !row_off_out) return fail(BBB_EINVAL, "null argument");
    if (k < 8 || k > BBB_MAX_K) return fail(BBB_EINVAL, "k must be 8..512");
    std::vector<uint16_t> t4;
    std::vector<uint8_t> w;
    search_candidate_host(k, seed, candidate, t4, w);
    uint32_t n = 0;
    for (int r = 0; r < k; r++) {
        row_off_out[r] = n;
        for (int q = 0; q < w[r]; q++) taps_out[n++] = t4[(size_t)4 * r + q];
    }
    row_off_out[k] = n;
    return BBB_OK;
}

}  // extern "C"
