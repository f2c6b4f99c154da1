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
