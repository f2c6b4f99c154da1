// awgn_small.hip -- the generated bit-sliced step for the other shipped recurrence matrices with a power-of-two
// order below 256 (n16, n32, n64, n128 of gateware/bbb/rng_recurrences.py; the same matrices as
// software/rnghunt/matrices/N).  Same formulation as awgn256_kernel (awgn_kernels.hip): 32 generators per lane,
// straight-line network from tools/gen_lutopt_kernel.py, log2(n) count planes per step sign-extended to an
// int8 sample (CLTGRNG.x is a log2(n)-bit signed Signal, rng.py:78), staged in LDS, one 16-byte store per
// generator per 16 steps.  No explicit AGPR placement is needed here: the state is at most 128 planes.
// The reference's transmitter only instantiates n256 (tx.py:68-71); these serve matrices found by the search.
#include "bbb_common.hpp"
#include "bitslice_util.hpp"
#include "awgn_launch.hpp"

#include "gen/lutopt16_gen.inc"
#include "gen/lutopt32_gen.inc"
#include "gen/lutopt64_gen.inc"
#include "gen/lutopt128_gen.inc"

namespace bbb {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> struct SmallNet;
#define BBB_SMALL_NET(N, LOG)                                                                                          \
    template <> struct SmallNet<N> {                                                                                   \
        static constexpr int kLog = LOG;                                                                               \
        static __device__ __forceinline__ void advance(const uint32_t (&a)[N], uint32_t (&b)[N]) { lutopt##N##_advance(a, b); } \
        static __device__ __forceinline__ void step(const uint32_t (&a)[N], uint32_t (&b)[N], uint32_t (&c)[LOG]) { lutopt##N##_step(a, b, c); } \
        static const uint16_t *ntaps() { return LUTOPT##N##_NTAPS; }                                                   \
        static const uint16_t *taps() { return LUTOPT##N##_TAPS; }                                                     \
    };
BBB_SMALL_NET(16, 4)
BBB_SMALL_NET(32, 5)
BBB_SMALL_NET(64, 6)
BBB_SMALL_NET(128, 7)

template <int N>
__global__ void __launch_bounds__(64)
awgn_small_kernel(const uint32_t *__restrict planes, int8_t *__restrict dst, unsigned long long nsamples, unsigned L,
                  unsigned long long G, unsigned nlanes) {
    constexpr int LOG = SmallNet<N>::kLog;
    __shared__ uint32_t Z[16 * 8 * 64];
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;

    // the step yields the sample of the state it is GIVEN (and its successor): the planes hold the state before
    // the first sample, so advance once
    uint32_t a[N], b[N], cnt[LOG];
#pragma unroll
    for (int p = 0; p < N; p++) b[p] = planes[(size_t)p * nlanes + LG];
    SmallNet<N>::advance(b, a);

    auto stage = [&](unsigned t) {
        uint32_t c8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) c8[q] = cnt[q < LOG ? q : LOG - 1];      // sign extension of the log2(n)-bit value
        planes8_to_bytes(c8);
#pragma unroll
        for (int i = 0; i < 8; i++) Z[(t * 8 + i) * 64 + lane] = c8[i];
    };

    const unsigned rounds = L / 16;
#pragma unroll 1
    for (unsigned r = 0; r < rounds; r++) {
#pragma unroll 1
        for (unsigned tt = 0; tt < 8; tt++) {
            SmallNet<N>::step(a, b, cnt);
            stage(2 * tt);
            SmallNet<N>::step(b, a, cnt);
            stage(2 * tt + 1);
        }
#pragma unroll 1
        for (unsigned i = 0; i < 8; i++) {
            uint32_t o[4][4];                 // o[w][q] after the transposes
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t z[4];
#pragma unroll
                for (int t = 0; t < 4; t++) z[t] = Z[((4 * w + t) * 8 + i) * 64 + lane];
                transpose4x4_bytes(z);        // z[q] = bytes (t = 4w..4w+3) of generator 8q+i
#pragma unroll
                for (int q = 0; q < 4; q++) o[w][q] = z[q];
            }
#pragma unroll
            for (unsigned q = 0; q < 4; q++) {
                const unsigned long long g = gen_index(wave, lane, 8 * q + i);
                const unsigned long long off = g * L + (unsigned long long)r * 16;
                if (g < G && off < nsamples) {
                    const u32x4 v = {o[0][q], o[1][q], o[2][q], o[3][q]};
                    if (off + 16 <= nsamples) {
                        *reinterpret_cast<u32x4 *>(dst + off) = v;
                    } else {
                        const unsigned n = (unsigned)(nsamples - off);
                        for (unsigned e = 0; e < n; e++) dst[off + e] = (int8_t)((o[e >> 2][q] >> (8 * (e & 3))) & 0xff);
                    }
                }
            }
        }
    }
}

template <int N>
static bool small_matches(const uint16_t *taps, const uint32_t *row_off) {
    uint32_t e = 0;
    for (int r = 0; r < N; r++) {
        if (row_off[r + 1] - row_off[r] != SmallNet<N>::ntaps()[r]) return false;
        for (uint32_t j = row_off[r]; j < row_off[r + 1]; j++)
            if (taps[j] != SmallNet<N>::taps()[e++]) return false;
    }
    return true;
}

// the order if (k, taps) is one of the matrices a generated kernel exists for, else 0
int awgn_small_matches(int k, const uint16_t *taps, const uint32_t *row_off) {
    switch (k) {
    case 16: return small_matches<16>(taps, row_off) ? 16 : 0;
    case 32: return small_matches<32>(taps, row_off) ? 32 : 0;
    case 64: return small_matches<64>(taps, row_off) ? 64 : 0;
    case 128: return small_matches<128>(taps, row_off) ? 128 : 0;
    default: return 0;
    }
}

int awgn_small_fill_launch(int k, const uint32_t *d_planes, int8_t *dst, uint64_t nsamples, unsigned L, uint64_t G,
                           unsigned nlanes, hipStream_t st) {
    const unsigned nwaves = nlanes / 64;
#define BBB_SMALL_LAUNCH(N)                                                                                  \
    case N:                                                                                                  \
        hipLaunchKernelGGL(awgn_small_kernel<N>, dim3(nwaves), dim3(64), 0, st, d_planes, dst,               \
                           (unsigned long long)nsamples, L, (unsigned long long)G, nlanes);                  \
        break;
    switch (k) {
        BBB_SMALL_LAUNCH(16)
        BBB_SMALL_LAUNCH(32)
        BBB_SMALL_LAUNCH(64)
        BBB_SMALL_LAUNCH(128)
    default: return fail(BBB_EINVAL, "no generated kernel for this order");
    }
#undef BBB_SMALL_LAUNCH
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

}  // namespace bbb
