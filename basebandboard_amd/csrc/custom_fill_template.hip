// custom_fill_template.hip -- a sample kernel for ONE recurrence matrix that is not among the shipped ones (for
// instance a matrix found by bbb_lutopt_search), built at run time by basebandboard_amd.LUTOPT.specialise():
//   basebandboard_amd/gen_lutopt_kernel.py <taps> custom_gen.inc
//   hipcc --offload-arch=gfx950 -shared -fPIC -DBBB_N=<n> -DBBB_LOG=<log2 n> -I<dir of custom_gen.inc> -I<csrc> \
//         custom_fill_template.hip -o libbbb_custom_<hash>.so
// and attached to a handle with bbb_lutopt_set_custom_fill.  Same formulation as awgn_small.hip / awgn256_kernel.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "bitslice_util.hpp"
#include "custom_abi.hpp"
#include "custom_gen.inc"

#define BBB_CAT2(a, b, c) a##b##c
#define BBB_CAT(a, b, c) BBB_CAT2(a, b, c)
#define BBB_STEP BBB_CAT(lutopt, BBB_N, _step)
#define BBB_ADVANCE BBB_CAT(lutopt, BBB_N, _advance)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
using bbb::gen_index;
using bbb::planes8_to_bytes;
using bbb::transpose4x4_bytes;

__global__ void __launch_bounds__(64)
#if BBB_N == 256
__attribute__((amdgpu_waves_per_eu(1, 1)))
#endif
bbb_custom_kernel(const uint32_t *__restrict planes, int8_t *__restrict dst, unsigned long long nsamples, unsigned L,
                  unsigned long long G, unsigned nlanes) {
    constexpr int N = BBB_N, LOG = BBB_LOG;
    __shared__ uint32_t Z[16 * 8 * 64];
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    uint32_t a[N], b[N], cnt[LOG];
#pragma unroll
    for (int p = 0; p < N; p++) b[p] = planes[(size_t)p * nlanes + LG];
    BBB_ADVANCE(b, a);                      // the step yields the sample of the state it is given
#if BBB_N == 256
    // 256 planes need the whole register file: the generator's explicit AGPR placement, as in awgn256_kernel
    uint32_t pa[N], pb[N];
#define BBB_PARK(p) BBB_ACC_WRITE(pa[p], a[p]);
    LUTOPT256_FOR_PARKED(BBB_PARK)
#undef BBB_PARK
    __builtin_amdgcn_s_setprio(3);
#define BBB_STEP_AB lutopt256_step_parked(a, pa, b, pb, cnt)
#define BBB_STEP_BA lutopt256_step_parked(b, pb, a, pa, cnt)
#else
#define BBB_STEP_AB BBB_STEP(a, b, cnt)
#define BBB_STEP_BA BBB_STEP(b, a, cnt)
#endif
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
            BBB_STEP_AB;
            stage(2 * tt);
            BBB_STEP_BA;
            stage(2 * tt + 1);
        }
#pragma unroll 1
        for (unsigned i = 0; i < 8; i++) {
            uint32_t o[4][4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t z[4];
#pragma unroll
                for (int t = 0; t < 4; t++) z[t] = Z[((4 * w + t) * 8 + i) * 64 + lane];
                transpose4x4_bytes(z);
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

// the signature of bbb_custom_fill_fn (include/bbb.h); 0 on success, the hipError_t otherwise
extern "C" int bbb_custom_fill(const uint32_t *planes_dev, int8_t *dst_dev, uint64_t nsamples, uint32_t L, uint64_t G,
                               uint32_t nlanes, void *hip_stream) {
    hipLaunchKernelGGL(bbb_custom_kernel, dim3(nlanes / 64), dim3(64), 0, (hipStream_t)hip_stream, planes_dev, dst_dev,
                       (unsigned long long)nsamples, L, (unsigned long long)G, nlanes);
    return (int)hipGetLastError();
}
extern "C" int bbb_custom_order(void) { return BBB_N; }
// the layout contract between this library and libbbb_hip.so (plane layout, TrialDev, launch geometry): custom_abi.hpp
extern "C" int bbb_custom_abi(void) { return BBB_CUSTOM_ABI; }

#if BBB_N == 256
// the fused BER trial kernels over this matrix's network (the signature of bbb_custom_ber_fn, include/bbb.h)
#include "bbb_common.hpp"
#include "awgn_launch.hpp"
namespace bbb {
std::string &last_error() {
    static thread_local std::string s;
    return s;
}
}  // namespace bbb
// (one translation unit, built at run time: the short list of group sizes)
#define BBB_BER_PART 0
#define BBB_BER_FEW_INSTANCES 1
#include "ber_kernels_impl.hpp"
extern "C" int bbb_custom_ber(uint32_t *planes_dev, uint32_t *prbs_planes_dev, const void *trials, int ncfg,
                              uint32_t nlanes, uint64_t *counters_dev, void *hip_stream) {
    return bbb::ber256_launch(planes_dev, prbs_planes_dev, (const bbb::TrialDev *)trials, ncfg, nlanes,
                              (unsigned long long *)counters_dev, (hipStream_t)hip_stream);
}
extern "C" const char *bbb_custom_last_error(void) { return bbb::last_error().c_str(); }
#endif
