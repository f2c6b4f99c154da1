// ber_kernels.hip -- fused Monte-Carlo trial: PRBS-k bit -> BPSK level + scaled CLT noise ->
// sign slicer -> compare -> count, entirely in bit-sliced form (no sample stream is written).
//
// Reference pieces being chained (paths relative to the reference checkout):
//   PRBS bit            gateware/bbb/prbs.py:32-35
//   LUTOPT + CLTGRNG    gateware/bbb/rng.py:38-40, 96-108
//   noise scaling, sum  gateware/bbb/tx.py:75-81   (12-bit signed registers)
//   slicer              gateware/bbb/rx.py:29      (bit^ = sample >= 0)
// The error/bit counters and the sharding are build-defined (the reference only raises an
// unconsumed `err` pulse: prbs.py:79, rx.py:46).
//
// The channel is a function of (sample, bit) only, so the host tabulates it: for each bit value
// the error indicator over T = sample + 128 (mod 256) is a parity of at most four threshold tests
// [T >= thr] (one for ordinary parameters; 12-bit wrap-around can add more).  On the GPU a
// threshold test on the 8 bit planes of T is 8 V_BITOP3 per 32 samples.
//
// Per wave: 2048 generators.  PRBS state is a second bit-sliced GF(2) recurrence (k planes, kept in
// lane-private LDS as a circular buffer).  Error counts: per-lane u32 -> wave reduction ->
// LDS-free single 64-bit atomic per wave (one wave per block).
#include "bbb_common.hpp"
#include "bitslice_util.hpp"
#include "awgn_launch.hpp"
#include "gen/lutopt256_gen.inc"

namespace bbb {

struct TrialK {              // kernel-argument form of TrialDev (uniform -> SGPRs)
    int32_t k, tap;
    int32_t nthr0, nthr1, inv0, inv1;
    uint32_t thrmask0[4][8];   // thrmask[i][q] = all-ones when bit q of threshold i is 1
    uint32_t thrmask1[4][8];
    uint32_t L, last_len;
    unsigned long long G;
};

// [T >= thr] for 32 samples: scan from the LSB; where the threshold bit is 1 the running
// result must AND with T's bit, where it is 0 it ORs (tm = 0 / ~0 selects).
__device__ __forceinline__ uint32_t ge_thr(const uint32_t (&T)[8], const uint32_t (&tm)[8]) {
    uint32_t ge = ~0u;
#pragma unroll
    for (int q = 0; q < 8; q++) ge = (tm[q] & (T[q] & ge)) | (~tm[q] & (T[q] | ge));
    return ge;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
ber256_kernel(const uint32_t *__restrict planes, const uint32_t *__restrict prbs_planes, TrialK tk, unsigned nlanes,
              unsigned long long *__restrict counters) {
    __shared__ uint32_t PR[32 * 64];          // PRBS state planes, circular: slot (head + i) % k = state bit i
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;

    uint32_t a[256], b[256], cnt[8];
#pragma unroll
    for (int p = 0; p < 256; p++) a[p] = planes[(size_t)p * nlanes + LG];
    for (int i = 0; i < tk.k; i++) PR[i * 64 + lane] = prbs_planes[(size_t)i * nlanes + LG];

    // which of this lane's 32 generators exist, and which one is the (possibly short) last one
    uint32_t vm_all = 0, vm_last = 0;
    for (unsigned j = 0; j < 32; j++) {
        const unsigned long long g = gen_index(wave, lane, j);
        if (g < tk.G) vm_all |= 1u << j;
        if (g + 1 == tk.G) vm_last |= 1u << j;
    }
    uint32_t nerr = 0, nbit = 0;
    int head = 0;

    auto consume = [&](unsigned t) {
        // PRBS: bit = s[k-1] ^ s[tap-1]; shift in at position 0 (prbs.py:34-35)
        int i1 = head + tk.k - 1;   if (i1 >= tk.k) i1 -= tk.k;
        int i2 = head + tk.tap - 1; if (i2 >= tk.k) i2 -= tk.k;
        const uint32_t pb = PR[i1 * 64 + lane] ^ PR[i2 * 64 + lane];
        head = head == 0 ? tk.k - 1 : head - 1;       // new slot 0 overwrites the old s[k-1]
        PR[head * 64 + lane] = pb;
        // T = sample + 128: flip the int8 sign plane back
        uint32_t T[8];
#pragma unroll
        for (int q = 0; q < 7; q++) T[q] = cnt[q];
        T[7] = ~cnt[7];
        uint32_t e0 = tk.inv0 ? ~0u : 0u, e1 = tk.inv1 ? ~0u : 0u;
        for (int i = 0; i < tk.nthr0; i++) e0 ^= ge_thr(T, tk.thrmask0[i]);
        for (int i = 0; i < tk.nthr1; i++) e1 ^= ge_thr(T, tk.thrmask1[i]);
        const uint32_t valid = t < tk.last_len ? vm_all : (vm_all & ~vm_last);
        const uint32_t e = ((pb & e1) | (~pb & e0)) & valid;
        nerr += __builtin_popcount(e);
        nbit += __builtin_popcount(valid);
    };

    const unsigned pairs = (tk.L + 1) / 2;
#pragma unroll 1
    for (unsigned it = 0; it < pairs; it++) {
        lutopt256_step(a, b, cnt);
        consume(2 * it);
        lutopt256_step(b, a, cnt);
        if (2 * it + 1 < tk.L) consume(2 * it + 1);
    }
    unsigned long long e64 = nerr, b64 = nbit;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e64 += __shfl_xor(e64, off, 64);
        b64 += __shfl_xor(b64, off, 64);
    }
    if (lane == 0) {
        atomicAdd(&counters[0], b64);
        if (e64) atomicAdd(&counters[1], e64);
    }
}

int ber256_launch(const uint32_t *d_planes, const uint32_t *d_prbs_planes, const TrialDev &t, unsigned nlanes,
                  unsigned long long *d_counters, hipStream_t st) {
    TrialK tk{};
    tk.k = t.prbs_k;
    tk.tap = t.prbs_tap;
    tk.L = t.L;
    tk.G = t.G;
    tk.last_len = (uint32_t)(t.nbits - (t.G - 1) * (uint64_t)t.L);
    int nth[2] = {0, 0}, inv[2] = {0, 0};
    for (int bv = 0; bv < 2; bv++) {
        for (int i = 0; i < t.nthr[bv]; i++) {
            const int th = t.thr[bv][i];
            if (th <= 0) { inv[bv] ^= 1; continue; }     // [T >= 0] is always true
            if (th >= 256) continue;                     // [T >= 256] never
            uint32_t(*dst)[8] = bv ? tk.thrmask1 : tk.thrmask0;
            for (int q = 0; q < 8; q++) dst[nth[bv]][q] = ((th >> q) & 1) ? ~0u : 0u;
            nth[bv]++;
        }
    }
    tk.nthr0 = nth[0]; tk.nthr1 = nth[1];
    tk.inv0 = inv[0];  tk.inv1 = inv[1];
    hipLaunchKernelGGL(ber256_kernel, dim3(nlanes / 64), dim3(64), 0, st, d_planes, d_prbs_planes, tk, nlanes,
                       d_counters);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

}  // namespace bbb
