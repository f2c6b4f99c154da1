// awgn512.hip -- generated-network sample kernel for the shipped n512 matrix (software/rnghunt/matrices/512; the
// reference's gateware stops at n256, the matrix exists as rnghunt output).  LUTOPT gateware/bbb/rng.py:14-55,
// CLTGRNG rng.py:58-108 (9-bit signed output for n = 512: rng.py:78).
//
// A 512-plane state does not fit a lane's 512 registers next to its successor, so the state is PACKED: 16 generators
// per lane, register p = plane p in bits 0..15 | plane 256 + (p ^ 1) in bits 16..31.  basebandboard_amd/gen_lutopt_kernel.py
// (Packed512Emitter) emits the step: per new register the i-th taps of its two rows are brought into one word by one
// V_PERM_B32, then XORed; the carry-save counter runs on both halves at once (9 planes each, same sign in both halves
// by the p ^ 1 pairing).  2018 VALU ops per 16 samples per lane (n256: 918 per 32) + 256 AGPR moves: the generator parks
// the state registers that have no room in the 256 VGPRs between their birth and their first reader (budget 230, as for n256).
//
// Per pair of steps: the halves' counts are added bit-sliced (T = number of +1 terms, sample = T - 256 as 9 bits
// signed, sign-extended to 16), the two steps merged into one word per plane (V_PERM), a 16 x 16 bit transpose on both
// halves gives per generator one word = two consecutive int16 samples; four such words per round of 8 steps go out as
// one 16-byte store per generator.
#include "bbb_common.hpp"
#include "bitslice_util.hpp"
#include "awgn_launch.hpp"
#include "gen/lutopt512_gen.inc"

namespace bbb {

// 16 generators per lane: for a fixed bit j the 64 lanes of a wave own 64 consecutive segments
__host__ __device__ __forceinline__ unsigned long long gen_index16(unsigned long long wave, unsigned lane, unsigned j) {
    return (wave * 16 + j) * 64 + lane;
}

// word-major states S[w * stride + g] (16 words of 32 bits) -> packed planes [256][nlanes]
__global__ void __launch_bounds__(256)
bitslice512p_kernel(const uint32_t *__restrict S, unsigned long long G, unsigned long long stride, unsigned nlanes,
                    uint32_t *__restrict planes) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long LG = t % nlanes;
    const unsigned wq = (unsigned)(t / nlanes);
    if (wq >= 8) return;
    const unsigned long long wave = LG >> 6;
    const unsigned lane = (unsigned)(LG & 63);
    uint32_t q[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const unsigned long long g = gen_index16(wave, lane, j);
        q[j] = g < G ? S[wq * stride + g] : 0u;                 // planes 32 wq .. 32 wq + 31
        q[16 + j] = g < G ? S[(8 + wq) * stride + g] : 0u;      // planes 256 + 32 wq ..
    }
    transpose32(q);        // q[b] bit j = plane 32 wq + b of generator j, bit 16 + j = plane 256 + 32 wq + b
#pragma unroll
    for (int b = 0; b < 32; b++)
        planes[(size_t)(32 * wq + b) * nlanes + LG] = (q[b] & 0x0000ffffu) | (q[b ^ 1] & 0xffff0000u);
}

// 16 plane words (bit j of each half = generator j) -> 16 words whose halves are generator j's 16-bit values
__device__ __forceinline__ void transpose16_halves(uint32_t (&m)[16]) {
#define BBB_T16_STAGE(S, M)                                             \
    _Pragma("unroll") for (int i = 0; i < 16; i++) {                    \
        if ((i & (S)) == 0) {                                           \
            const uint32_t x = m[i], y = m[i + (S)];                    \
            m[i] = bit_select((M), x, y << (S));                        \
            m[i + (S)] = bit_select((M), x >> (S), y);                  \
        }                                                               \
    }
    BBB_T16_STAGE(8, 0x00FF00FFu)
    BBB_T16_STAGE(4, 0x0F0F0F0Fu)
    BBB_T16_STAGE(2, 0x33333333u)
    BBB_T16_STAGE(1, 0x55555555u)
#undef BBB_T16_STAGE
}

// sample planes of one step from the two half counters: s[q], q = 0..8 (valid in bits 0..15)
__device__ __forceinline__ void add_halves(const uint32_t (&cnt)[9], uint32_t (&s)[9]) {
    uint32_t carry = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const uint32_t lo = cnt[q], hi = cnt[q] >> 16;
        s[q] = __builtin_amdgcn_bitop3_b32(lo, hi, carry, 0x96);
        carry = __builtin_amdgcn_bitop3_b32(lo, hi, carry, 0xe8);
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
awgn512p_kernel(const uint32_t *__restrict planes, int16_t *__restrict dst, unsigned long long nsamples, unsigned L,
                unsigned long long G, unsigned nlanes) {
    __shared__ uint32_t Z[4 * 16 * 64];          // [pair of steps u][generator j][lane]
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    // (the state registers in LUTOPT512_PARKED travel in AGPRs between their birth and their first reader of the next step,
    // placed by the generator: gen_lutopt_kernel.py, Packed512Emitter)
    uint32_t a[256], b[256], pa[256], pb[256], c0[9], c1[9];
#pragma unroll
    for (int p = 0; p < 256; p++) {
        const uint32_t v = planes[(size_t)p * nlanes + LG];
        if (lutopt512_is_parked(p)) BBB_ACC_WRITE(pa[p], v); else a[p] = v;
    }
    const unsigned rounds = L / 8;
#pragma unroll 1
    for (unsigned r = 0; r < rounds; r++) {
#pragma unroll 1
        for (unsigned u = 0; u < 4; u++) {
            lutopt512p_step_new_parked(a, pa, b, pb, c0);       // the step yields the sample of the NEW state
            lutopt512p_step_new_parked(b, pb, a, pa, c1);
            uint32_t s0[9], s1[9], m[16];
            add_halves(c0, s0);
            add_halves(c1, s1);
#pragma unroll
            for (int q = 0; q < 9; q++) m[q] = __builtin_amdgcn_perm(s1[q], s0[q], 0x05040100u);   // step 2u | step 2u+1 << 16
            m[8] = ~m[8];                        // (T - 256) mod 512: flip bit 8 ...
#pragma unroll
            for (int q = 9; q < 16; q++) m[q] = m[8];                                               // ... and extend the sign
            transpose16_halves(m);
#pragma unroll
            for (int j = 0; j < 16; j++) Z[(u * 16 + j) * 64 + lane] = m[j];
        }
#pragma unroll 1
        for (unsigned j = 0; j < 16; j++) {
            const unsigned long long g = gen_index16(wave, lane, j);
            const unsigned long long off = g * L + (unsigned long long)r * 8;
            if (!(g < G && off < nsamples)) continue;
            const u32x4 v = {Z[(0 * 16 + j) * 64 + lane], Z[(1 * 16 + j) * 64 + lane], Z[(2 * 16 + j) * 64 + lane],
                             Z[(3 * 16 + j) * 64 + lane]};
            if (off + 8 <= nsamples) {
                *reinterpret_cast<u32x4 *>(dst + off) = v;
            } else {
                const unsigned n = (unsigned)(nsamples - off);
                for (unsigned e = 0; e < n; e++) dst[off + e] = (int16_t)((v[e >> 1] >> (16 * (e & 1))) & 0xffff);
            }
        }
    }
}

int bitslice512p_launch(const uint32_t *d_states, uint64_t G, uint64_t stride, unsigned nlanes, uint32_t *d_planes, hipStream_t st) {
    const uint64_t threads = (uint64_t)nlanes * 8;
    hipLaunchKernelGGL(bitslice512p_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_states,
                       (unsigned long long)G, (unsigned long long)stride, nlanes, d_planes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int awgn512p_fill_launch(const uint32_t *d_planes, int16_t *dst, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                         hipStream_t st) {
    hipLaunchKernelGGL(awgn512p_kernel, dim3(nlanes / 64), dim3(64), 0, st, d_planes, dst, (unsigned long long)nsamples, L,
                       (unsigned long long)G, nlanes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

bool awgn512p_matches(int k, const uint16_t *taps, const uint32_t *row_off) {
    if (k != 512) return false;
    uint32_t e = 0;
    for (int r = 0; r < 512; r++) {
        if (row_off[r + 1] - row_off[r] != LUTOPT512_NTAPS[r]) return false;
        for (uint32_t j = row_off[r]; j < row_off[r + 1]; j++)
            if (taps[j] != LUTOPT512_TAPS[e++]) return false;
    }
    return true;
}

}  // namespace bbb
