// gf2.hpp -- dense GF(2) matrices up to 512 x 512 for jump-ahead.
//
// The reference keeps this algebra in its Rust crate (software/rnghunt/src/binary_matrix.rs:
// `dot` 53-64, `recur` 68-76; polynomial form of the same jump in binary_polynomial.rs:136-163).
// Here it serves one purpose: split ONE sequential generator stream across many GPU lanes,
// lane g starting at A^(g*L) * x0, so that the concatenated output equals the stream a single
// LUTOPT / PRBS instance emits.
//
// Layout: row-major bit rows, bit c of row r at rows[r*W + c/64] >> (c%64); vectors use the
// same LSB-first bit order as the HDL integer (gateware/bbb/rng.py:135).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace bbb {

struct GF2Mat {
    int n = 0;   // square n x n
    int W = 0;   // words per row
    std::vector<uint64_t> rows;

    GF2Mat() = default;
    explicit GF2Mat(int n_) : n(n_), W((n_ + 63) / 64), rows((size_t)n_ * ((n_ + 63) / 64), 0) {}

    static GF2Mat identity(int n) {
        GF2Mat m(n);
        for (int i = 0; i < n; i++) m.set(i, i);
        return m;
    }
    void set(int r, int c) { rows[(size_t)r * W + (c >> 6)] |= 1ull << (c & 63); }
    bool get(int r, int c) const { return (rows[(size_t)r * W + (c >> 6)] >> (c & 63)) & 1ull; }
    const uint64_t *row(int r) const { return &rows[(size_t)r * W]; }
    uint64_t *row(int r) { return &rows[(size_t)r * W]; }

    // y = M x
    void matvec(const uint64_t *x, uint64_t *y) const {
        uint64_t out[8] = {0};
        for (int r = 0; r < n; r++) {
            const uint64_t *a = row(r);
            uint64_t acc = 0;
            for (int w = 0; w < W; w++) acc ^= a[w] & x[w];
            out[r >> 6] |= (uint64_t)(__builtin_popcountll(acc) & 1) << (r & 63);
        }
        std::memcpy(y, out, sizeof(uint64_t) * (size_t)W);
    }

    // y = M^T x, i.e. the XOR of the rows of this matrix selected by the bits of x: for y = B x keep B's TRANSPOSE and call
    // this -- about n/2 row XORs instead of n dot products (4x fewer word operations at n = 256; the per-call host work in
    // front of a seeding is fifteen of these)
    void matvec_t(const uint64_t *x, uint64_t *y) const {
        uint64_t out[8] = {0};
        for (int w = 0; w < W; w++) {
            uint64_t bits = x[w];
            while (bits) {
                const int c = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                if (c >= n) break;
                const uint64_t *b = row(c);
                for (int q = 0; q < W; q++) out[q] ^= b[q];
            }
        }
        std::memcpy(y, out, sizeof(uint64_t) * (size_t)W);
    }

    // C = this * B : row r of C is the XOR of the rows c of B for which this[r][c] = 1
    GF2Mat mul(const GF2Mat &B) const {
        GF2Mat C(n);
        for (int r = 0; r < n; r++) {
            uint64_t *out = C.row(r);
            const uint64_t *a = row(r);
            for (int w = 0; w < W; w++) {
                uint64_t bits = a[w];
                while (bits) {
                    int c = (w << 6) + __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const uint64_t *b = B.row(c);
                    for (int q = 0; q < W; q++) out[q] ^= b[q];
                }
            }
        }
        return C;
    }

    GF2Mat transpose() const {
        GF2Mat T(n);
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++)
                if (get(r, c)) T.set(c, r);
        return T;
    }
};

// Powers A^(2^i), i = 0..63, computed lazily and cached: x_t = A^t x_0 needs at most 64
// matrix-vector products, A^L at most 64 matrix-matrix products.
struct GF2Powers {
    std::vector<GF2Mat> p2;
    explicit GF2Powers(const GF2Mat &A) { p2.push_back(A); }
    const GF2Mat &pow2(int i) {
        while ((int)p2.size() <= i) p2.push_back(p2.back().mul(p2.back()));
        return p2[(size_t)i];
    }
    // y = A^e x
    void apply(uint64_t e, const uint64_t *x, uint64_t *y) {
        uint64_t v[8];
        const int W = p2[0].W;
        std::memcpy(v, x, sizeof(uint64_t) * (size_t)W);
        for (int i = 0; e; i++, e >>= 1)
            if (e & 1) pow2(i).matvec(v, v);
        std::memcpy(y, v, sizeof(uint64_t) * (size_t)W);
    }
    GF2Mat power(uint64_t e) {
        GF2Mat R = GF2Mat::identity(p2[0].n);
        for (int i = 0; e; i++, e >>= 1)
            if (e & 1) R = pow2(i).mul(R);
        return R;
    }
};

// Companion matrix of the PRBS-k Fibonacci LFSR (gateware/bbb/prbs.py:32-35):
// s'[0] = s[k-1] ^ s[tap-1]; s'[i] = s[i-1].
inline GF2Mat prbs_matrix(int k, int tap) {
    GF2Mat T(k);
    T.set(0, k - 1);
    T.set(0, tap - 1);
    for (int i = 1; i < k; i++) T.set(i, i - 1);
    return T;
}

inline int prbs_tap(int k) {
    switch (k) {   // TAPS, gateware/bbb/prbs.py:14
    case 7: return 6;  case 9: return 5;  case 11: return 9; case 15: return 14;
    case 20: return 3; case 23: return 18; case 31: return 28;
    default: return 0;
    }
}

}  // namespace bbb
