// bitslice_util.hpp -- small bit-matrix helpers shared by the HIP kernels (and compiled on the
// host by tests/test_codegen_cpu.py to unit-test them without a GPU).
#pragma once
#include <cstdint>

#ifndef BBB_HD
#ifdef __HIPCC__
#define BBB_HD __host__ __device__ __forceinline__
#else
#define BBB_HD inline
#endif
#endif

namespace bbb {

// In-place 32x32 bit-matrix transpose: afterwards q[i] bit j == (old q[j]) bit i.
// Five block-swap stages; stage s exchanges the (row bit s = 0, column bit s = 1) blocks with
// their mirror images.
BBB_HD void transpose32(uint32_t (&q)[32]) {
#define BBB_T32_STAGE(S, M)                                         \
    _Pragma("unroll") for (int i = 0; i < 32; i++) {                \
        if ((i & (S)) == 0) {                                       \
            const uint32_t t = ((q[i] >> (S)) ^ q[i + (S)]) & (M);  \
            q[i + (S)] ^= t;                                        \
            q[i] ^= t << (S);                                       \
        }                                                           \
    }
    BBB_T32_STAGE(16, 0x0000FFFFu)
    BBB_T32_STAGE(8, 0x00FF00FFu)
    BBB_T32_STAGE(4, 0x0F0F0F0Fu)
    BBB_T32_STAGE(2, 0x33333333u)
    BBB_T32_STAGE(1, 0x55555555u)
#undef BBB_T32_STAGE
}

// Byte permute with the V_PERM_B32 convention: the 8 source bytes are {hi, lo} (lo = bytes 0..3),
// byte i of sel picks the source byte for result byte i.
BBB_HD uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t in = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((in >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
#endif
}

// (mask & a) | (~mask & b)  -- one V_BFI_B32
BBB_HD uint32_t bit_select(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }

// 8 bit-plane words -> 8 byte-sliced words.  In: c[b] bit j = bit b of sample j (j = 0..31).
// Out: c[i] byte q = the whole sample (8 bits) of generator j = 8q + i.
// Three block-swap stages on the 8x8 bit blocks (one block per byte column).
BBB_HD void planes8_to_bytes(uint32_t (&c)[8]) {
#define BBB_B8_STAGE(S, M)                                              \
    _Pragma("unroll") for (int i = 0; i < 8; i++) {                     \
        if ((i & (S)) == 0) {                                           \
            const uint32_t x = c[i], y = c[i + (S)];                    \
            c[i] = bit_select((M), x, y << (S));                        \
            c[i + (S)] = bit_select((M), x >> (S), y);                  \
        }                                                               \
    }
    BBB_B8_STAGE(4, 0x0F0F0F0Fu)
    BBB_B8_STAGE(2, 0x33333333u)
    BBB_B8_STAGE(1, 0x55555555u)
#undef BBB_B8_STAGE
}

// 4x4 byte transpose: in r[t] byte q  ->  out r[q] byte t.   8 V_PERM_B32.
BBB_HD void transpose4x4_bytes(uint32_t (&r)[4]) {
    const uint32_t a = byte_perm(r[1], r[0], 0x05010400u);   // r0.b0 r1.b0 r0.b1 r1.b1
    const uint32_t b = byte_perm(r[1], r[0], 0x07030602u);   // r0.b2 r1.b2 r0.b3 r1.b3
    const uint32_t c = byte_perm(r[3], r[2], 0x05010400u);
    const uint32_t d = byte_perm(r[3], r[2], 0x07030602u);
    r[0] = byte_perm(c, a, 0x05040100u);                     // a.b0 a.b1 c.b0 c.b1
    r[1] = byte_perm(c, a, 0x07060302u);
    r[2] = byte_perm(d, b, 0x05040100u);
    r[3] = byte_perm(d, b, 0x07060302u);
}

// Generator numbering shared by host and kernels.  A lane holds 32 generators (one per bit of
// every plane register); lane-global index LG = wave*64 + lane.  Generator
//     g = (wave*32 + j)*64 + lane            (j = bit position)
// so that, for fixed j, the 64 lanes of a wave own 64 CONSECUTIVE output segments.
BBB_HD uint64_t gen_index(uint64_t wave, unsigned lane, unsigned j) { return (wave * 32 + j) * 64 + lane; }

// Thue-Morse sign mask of one 64-bit state word: bit j set when popcount(j) is even, i.e. when
// the CLT adder tree (gateware/bbb/rng.py:96-105) gives state bit 64w+j weight +1 for even
// popcount(w).  For odd popcount(w) the complement applies.
constexpr uint64_t kThueMorse64 = 0x9669699669969669ull;

}  // namespace bbb
