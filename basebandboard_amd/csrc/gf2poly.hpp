// gf2poly.hpp -- polynomials over GF(2) up to degree 512, host side: the arithmetic of the
// reference's software/rnghunt/src/binary_polynomial.rs (modmult :87-133, modexp :135-163,
// check_integer :165-176, is_primitive :178-216) and the way its search tool derives a recurrence's
// characteristic polynomial (src/bin/rnghunt.rs:27-38).  The GPU search (search_kernels.hip) does
// the same arithmetic cooperatively per wave; this is the host form behind bbb_gf2_poly_* and the
// re-verification of every matrix the search returns.
//
// Layout: bit i of the word array = coefficient of x^i (LSB first), 9 words.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace bbb {

constexpr int kPolyWords = 9;     // degree <= 512

struct GF2Poly {
    uint64_t w[kPolyWords];
    GF2Poly() { std::memset(w, 0, sizeof w); }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
    void set(int i) { w[i >> 6] |= 1ull << (i & 63); }
    void flip(int i) { w[i >> 6] ^= 1ull << (i & 63); }
    int degree() const {
        for (int q = kPolyWords - 1; q >= 0; q--)
            if (w[q]) return 64 * q + 63 - __builtin_clzll(w[q]);
        return -1;
    }
    int weight() const {
        int c = 0;
        for (int q = 0; q < kPolyWords; q++) c += __builtin_popcountll(w[q]);
        return c;
    }
    bool operator==(const GF2Poly &o) const { return std::memcmp(w, o.w, sizeof w) == 0; }
    void xor_in(const GF2Poly &o) {
        for (int q = 0; q < kPolyWords; q++) w[q] ^= o.w[q];
    }
    void shl1() {
        for (int q = kPolyWords - 1; q > 0; q--) w[q] = (w[q] << 1) | (w[q - 1] >> 63);
        w[0] <<= 1;
    }
};

// f <- f * x mod p, deg p = n, deg f < n
inline void gf2_mulx_mod(GF2Poly &f, const GF2Poly &p, int n) {
    f.shl1();
    if (f.bit(n)) f.xor_in(p);
}

// f * g mod p (shift and add, binary_polynomial.rs:87-133), deg f, deg g < n = deg p
inline GF2Poly gf2_mulmod(const GF2Poly &f, const GF2Poly &g, const GF2Poly &p, int n) {
    GF2Poly r, gs = g;
    const int df = f.degree();
    for (int i = 0; i <= df; i++) {
        if (f.bit(i)) r.xor_in(gs);
        gf2_mulx_mod(gs, p, n);
    }
    return r;
}

// x^e mod p; e = `ewords` little-endian words (binary_polynomial.rs:135-163: square, then multiply
// by x where the exponent bit is set, from the bit below the leading one downwards)
inline GF2Poly gf2_modexp(const uint64_t *ewords, int nwords, const GF2Poly &p, int n) {
    GF2Poly f;
    int top = -1;
    for (int q = nwords - 1; q >= 0 && top < 0; q--)
        if (ewords[q]) top = 64 * q + 63 - __builtin_clzll(ewords[q]);
    if (top < 0) { f.set(0); return f; }                    // x^0 = 1
    if (n == 0) return f;                                   // everything is 0 mod a constant
    if (n == 1) { if (p.bit(0)) f.set(0); return f; }       // x = p(0) mod (x + p0)
    f.set(1);                                               // x
    for (int b = top - 1; b >= 0; b--) {
        f = gf2_mulmod(f, f, p, n);
        if ((ewords[b >> 6] >> (b & 63)) & 1ull) gf2_mulx_mod(f, p, n);
    }
    return f;
}

#include "gen/mersenne_factors.inc"

inline const MersenneEntry *mersenne_entry(int n) {
    for (const MersenneEntry &e : kMersenne)
        if (e.n == n) return &e;
    return nullptr;
}

// 1 primitive, 0 not, -1 the factorisation of 2^deg - 1 is not in the table.
// Steps as in binary_polynomial.rs:178-216.
inline int gf2_is_primitive(const GF2Poly &p) {
    const int n = p.degree();
    if (n == -1) return 1;                          // :181-183 (the reference's choice for the zero polynomial)
    if (!p.bit(0)) return 0;                        // :186-188 nonzero constant term
    if (p.weight() % 2 != 1) return 0;              // :191-193 odd number of terms
    const MersenneEntry *e = mersenne_entry(n);
    if (!e) return -1;
    const int W = (n + 63) / 64;
    GF2Poly one;
    one.set(0);
    if (!(gf2_modexp(kMersenneWords + e->offset, W, p, n) == one)) return 0;          // :199-201
    for (int i = 1; i < e->nexp; i++)                                                   // :205-209
        if (gf2_modexp(kMersenneWords + e->offset + (size_t)i * W, W, p, n) == one) return 0;
    return 1;
}

// Berlekamp-Massey on packed bits s[0..len): connection polynomial C (C[0] = 1), returns L
inline int gf2_berlekamp_massey(const std::vector<uint8_t> &s, GF2Poly &C) {
    const int len = (int)s.size();
    GF2Poly B, T;
    C = GF2Poly();
    C.set(0);
    B.set(0);
    int L = 0, m = -1;
    for (int i = 0; i < len; i++) {
        int d = s[i] & 1;
        for (int j = 1; j <= L; j++) d ^= (int)C.bit(j) & s[i - j];
        if (!d) continue;
        T = C;
        const int shift = i - m;
        for (int j = 0; j + shift <= 512 && j <= 512; j++)
            if (B.bit(j)) C.flip(j + shift);
        if (2 * L <= i) {
            L = i + 1 - L;
            m = i;
            B = T;
        }
    }
    return L;
}

// The polynomial rnghunt's search examines (src/bin/rnghunt.rs:27-38): bit 0 of 2k successive
// states from the all-ones vector, reversed, through Berlekamp-Massey.  Returns its degree; P has
// the connection polynomial's coefficient i at power (degree - i), i.e. index 0 = leading term as
// in the reference's BinaryPolynomial.
inline int lutopt_charpoly(int k, const uint16_t *taps, const uint32_t *row_off, GF2Poly &P) {
    std::vector<uint8_t> x(k, 1), y(k), seq(2 * k);
    for (int s = 0; s < 2 * k; s++) {
        for (int r = 0; r < k; r++) {
            uint8_t v = 0;
            for (uint32_t q = row_off[r]; q < row_off[r + 1]; q++) v ^= x[taps[q]];
            y[r] = v;
        }
        x.swap(y);
        seq[2 * k - 1 - s] = x[0];
    }
    GF2Poly C;
    const int L = gf2_berlekamp_massey(seq, C);
    P = GF2Poly();
    for (int i = 0; i <= L; i++)
        if (C.bit(i)) P.set(L - i);
    return L;
}

}  // namespace bbb
