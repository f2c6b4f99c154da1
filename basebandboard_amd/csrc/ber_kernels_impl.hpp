// ber_kernels_impl.hpp -- the fused trial kernels and their launcher (see ber_kernels.hip for the formulation).
// Textually included AFTER a generated network lutopt256_* (gen/lutopt256_gen.inc for the shipped matrix, the
// custom_gen.inc of custom_fill_template.hip for any other k = 256 matrix), so that the same kernels exist for
// a matrix found by bbb_lutopt_search.
#pragma once
namespace bbb {

constexpr int kMaxCfg = BBB_BER_MAX_GROUP;   // channel settings evaluated per launch on ONE noise / PRBS stream

// kBerGeneral: one channel setting per launch, any number (<= 4) of thresholds per bit value
//              (12-bit wrap-around cases).
// kBerPair   : up to kMaxCfg settings per launch, each with exactly ONE threshold per bit value plus an
//              optional complement: two 8-instruction scans per setting.
// kBerFast   : the same for every non-wrapping channel (bit 0 errs when T >= thr0, bit 1 when T < thr1,
//              thr0 + thr1 = 256 or 257): ONE scan per setting on X = bit ? ~T : T.
// [T >= thr] for 32 samples: scan from the LSB; where the threshold bit is 1 the running result ANDs with T's bit, where
// it is 0 it ORs: tm ? (T & ge) : (T | ge) = one V_BITOP3 (truth table 0xD4 over a = T, b = ge, c = tm) with the per-bit
// mask tm a scalar (0 or ~0).
enum { kBerFast = 0, kBerPair = 1, kBerGeneral = 2 };

// ---------------------------------------------------------------------------------------------
// The fused trial kernel, PLANES generation (round 4).  Its predecessor (rounds 1-3: docs/DESIGN_rounds_1_2.md) used the
// budget-180 step (242 AGPR moves), staged eight steps of count planes through LDS for a comparator loop of its own per round
// and kept the PRBS planes in a circular LDS buffer with scalar index arithmetic: 1368 issued VALU instructions per step and
// wave (SQ_INSTS_VALU, profiles/r04_ber_old_pmc.json) where the sample kernel of the stream needs 1061.  This one is the
// sample kernel's loop -- lutopt256_step_parked_ber, no round end, no LDS staging -- with the
// comparators run on the count planes while they are still in registers:
//   X[q] = (T[q] ^ bit) & valid                      8 V_BITOP3 per step (T[7] = ~cnt[7]: folded into the truth table)
//   per setting  ge = (bit | eq) & valid             1
//                ge = tm[q] ? X[q] & ge : X[q] | ge  8   (LSB-first scan of [X >= thr]; tm[q] = bit q of thr as a 0 / ~0 SCALAR)
//                nerr += popcount(ge)                1   (V_BCNT_U32_B32 accumulates)
// The scalar masks of twelve settings (9 each) do not fit the SGPR file beside the loop's own scalars, and hipcc spills
// scalars to VGPR lanes (a V_READLANE per use).  They are therefore STREAMED: the kernel-argument segment holds the
// table, each step re-reads it with scalar loads (s_load_dwordx8 + s_load_dword per setting, ~6 cycles each for the one
// wave of a SIMD: experiments/ubench6.hip) behind an opaque pointer that keeps hipcc from hoisting them.
// PRBS: b[t] = b[t-k] ^ b[t-tap] (prbs.py:32-35) on a time-indexed ring in LDS, slot t & 31, every slot written twice
// (s and s + 32) so that the reads of one iteration -- slots u + 32 - k, u + 33 - k, u + 32 - tap, u + 33 - tap,
// u = t & 31 even -- never wrap: two ds_read2st64_b32, two XOR, two ds_write2st64_b32 per two steps.
// The bit counters are not counted: a trial's number of bits is known to the host (wave 0 adds it).
// ---------------------------------------------------------------------------------------------
struct BerMasks { uint32_t m[kMaxCfg][32]; };   // fast: [c][0..7] tm, [c][8] eq; pair: [c][0..7] thr0, [8..15] thr1, [16] inv0, [17] inv1;
                                                // general (one setting): [bv * 4 + i][0..7] tm, [..][8] enable, m[8][bv] inv
struct TrialF {
    int32_t k, tap, ncfg, save;     // save: leave the generator's and the PRBS state behind (a continued trial)
    uint32_t L, last_len;
    unsigned long long G, nbits;
};

template <int MODE, int NC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
ber256_fused_kernel(BerMasks mk, uint32_t *__restrict planes, uint32_t *__restrict prbs_planes, TrialF tk, unsigned nlanes,
                    unsigned long long *__restrict counters) {
    __shared__ uint32_t ring[64 * 64];
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    __builtin_amdgcn_s_setprio(3);

    // `planes` = the state OF the first sample (the host seeds one clock past the stream position: no advance in front of the
    // loop, and a continued trial -- bbb_ber_run_* -- finds the state this kernel left); the parked planes go straight to
    // their AGPRs (scalar base per plane + ONE 32-bit lane offset)
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
    {
        const uint32_t voff = (uint32_t)LG * 4u;
#define BBB_PLANE(p) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(planes + (size_t)(p) * nlanes) + (unsigned long long)voff))
#define BBB_PARK(p) { const uint32_t v_ = BBB_PLANE(p); BBB_ACC_WRITE(pa[p], v_); }
        LUTOPT256_FOR_PARKED_BER(BBB_PARK)
#undef BBB_PARK
#pragma unroll
        for (int p = 0; p < 256; p++)
            if (!lutopt256_ber_is_parked(p)) a[p] = BBB_PLANE(p);
#undef BBB_PLANE
    }
    // ring slot 31 - i (and its copy 63 - i) = b[-1 - i] = LFSR state bit i (prbs.py:34-35: the new bit enters at position 0)
    for (int i = 0; i < tk.k; i++) {
        const uint32_t v = prbs_planes[(size_t)i * nlanes + LG];
        ring[(31 - i) * 64 + lane] = v;
        ring[(63 - i) * 64 + lane] = v;
    }
    // which of this lane's 32 generators exist, and which one is the (possibly short) last one
    uint32_t vm_all = 0, vm_last = 0;
    for (unsigned j = 0; j < 32; j++) {
        const unsigned long long g = gen_index(wave, lane, j);
        if (g < tk.G) vm_all |= 1u << j;
        if (g + 1 == tk.G) vm_last |= 1u << j;
    }
    uint32_t nerr[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) nerr[c] = 0;
    const unsigned ka = (unsigned)(32 - tk.k) * 64, ta = (unsigned)(32 - tk.tap) * 64;

    // Scalar masks.  Fast mode: the eq masks of all settings and the threshold masks of the first kRes settings live in SGPRs
    // for the whole kernel; the threshold masks of the others are streamed through two 8-SGPR windows by s_load_dwordx8 --
    // issued in front of the generator step (settings kRes, kRes + 1: the step hides their latency) and right after a window's
    // setting is done (the resident settings' scans hide it).  Written as inline asm so that the loads stay where they are put:
    // left to hipcc they sit in front of their first use and the one wave of the SIMD waits out every one of them (1.20 ms per
    // 1e9 steps, 85 % VALU busy: profiles/r04_ber_new1_pmc.json).  A compiler-inserted s_waitcnt lgkmcnt(N) that does not know
    // of these loads only waits longer than it had to.
    typedef const uint32_t __attribute__((address_space(4))) *kptr_t;
    typedef uint32_t s8_t __attribute__((ext_vector_type(8)));
    const kptr_t mp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();       // the table is the FIRST kernel argument: offset 0
    #ifndef BBB_BER_KRES
#define BBB_BER_KRES 7
#endif
    constexpr int kRes = MODE == kBerFast ? (NC < BBB_BER_KRES ? NC : BBB_BER_KRES) : 0;
    uint32_t eqm[MODE == kBerFast ? NC : 1], tmr[kRes ? kRes : 1][8];
    if constexpr (MODE == kBerFast) {
#pragma unroll
        for (int c = 0; c < NC; c++) eqm[c] = mp[c * 32 + 8];
#pragma unroll
        for (int c = 0; c < kRes; c++)
#pragma unroll
            for (int q = 0; q < 8; q++) tmr[c][q] = mp[c * 32 + q];
    }
    s8_t w0 = {0, 0, 0, 0, 0, 0, 0, 0}, w1 = w0;
#define BBB_SLOAD8(w, c) asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(w) : "s"(mp), "n"((c) * 128))
    // the same, pinned between the scans around it: `dep` (a value the NEXT scan reads) passes through the statement, `after` (the
    // counter the PREVIOUS scan wrote) is an input -- hipcc otherwise lets the load sink to just in front of its wait
#define BBB_SLOAD8_AT(w, c, after, dep) asm volatile("s_load_dwordx8 %0, %2, %3" : "=&s"(w), "+v"(dep) : "s"(mp), "n"((c) * 128), "v"(after))
    // a wait names EVERY window with a load in flight as "+s": whatever reads a window afterwards -- the scans' V_BITOP3 are
    // ordinary code hipcc is free to move -- depends on the wait's output and cannot be hoisted above it (nor can a copy or a
    // spill of the window be placed in front of it)
#define BBB_SWAIT2(wa, wb) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wa), "+s"(wb))
#define BBB_SWAIT2_AFTER(wa, wb, after) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wa), "+s"(wb) : "v"(after))      // not before `after` is computed
#define BBB_SWAIT_AFTER(w, after) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w) : "v"(after))
    auto prefetch = [&]() {
        if constexpr (MODE == kBerFast && NC > kRes) BBB_SLOAD8(w0, kRes);
        if constexpr (MODE == kBerFast && NC > kRes + 1) BBB_SLOAD8(w1, kRes + 1);
    };

    auto compare = [&](const uint32_t pbit, const uint32_t valid) {
        if constexpr (MODE == kBerFast) {
            uint32_t X[8];
#pragma unroll
            for (int q = 0; q < 7; q++) X[q] = __builtin_amdgcn_bitop3_b32(cnt[q], pbit, valid, 0x28);     // (a ^ b) & c
            X[7] = __builtin_amdgcn_bitop3_b32(cnt[7], pbit, valid, 0x82);                                 // ~(a ^ b) & c
            auto scan = [&](const int c, auto tm) {
                uint32_t ge = __builtin_amdgcn_bitop3_b32(pbit, eqm[c], valid, 0xA8);                      // (a | b) & c
#pragma unroll
                for (int q = 0; q < 8; q++) ge = __builtin_amdgcn_bitop3_b32(X[q], ge, tm[q], 0xD4);
                nerr[c] += __builtin_popcount(ge);
            };
            if constexpr (NC > kRes) {
                if constexpr (NC > kRes + 1) BBB_SWAIT2(w0, w1); else BBB_SWAIT_AFTER(w0, valid);
                scan(kRes, w0);
                if constexpr (NC > kRes + 2) BBB_SLOAD8_AT(w0, kRes + 2, nerr[kRes], X[0]);
            }
            if constexpr (NC > kRes + 1) {
                scan(kRes + 1, w1);                     // (landed: the wait above named both windows)
                if constexpr (NC > kRes + 3) BBB_SLOAD8_AT(w1, kRes + 3, nerr[kRes + 1], X[0]);
            }
#pragma unroll
            for (int c = 0; c < kRes; c++) scan(c, tmr[c]);
            if constexpr (NC > kRes + 2) {
                if constexpr (NC > kRes + 3) BBB_SWAIT2_AFTER(w0, w1, nerr[kRes - 1]);     // (w1: setting kRes + 3's load, issued behind scan kRes + 1)
                else BBB_SWAIT_AFTER(w0, nerr[kRes - 1]);
                scan(kRes + 2, w0);
                if constexpr (NC > kRes + 4) BBB_SLOAD8_AT(w0, kRes + 4, nerr[kRes + 2], X[0]);
            }
            if constexpr (NC > kRes + 3) scan(kRes + 3, w1);     // (landed and pinned by the two-window wait above)
            if constexpr (NC > kRes + 4) {
                BBB_SWAIT_AFTER(w0, nerr[kRes + 3]);          // (the only load in flight here is w0's)
                scan(kRes + 4, w0);
            }
            static_assert(NC <= kRes + 5, "two windows serve five streamed settings");
        } else {
            kptr_t mo = mp;
            asm volatile("" : "+s"(mo));            // (opaque: the scalar loads below stay inside the step)
            uint32_t T[8];
#pragma unroll
            for (int q = 0; q < 7; q++) T[q] = cnt[q];
            T[7] = ~cnt[7];
            if constexpr (MODE == kBerPair) {
                const uint32_t pv1 = pbit & valid, pv0 = ~pbit & valid;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    uint32_t g0 = ~0u, g1 = ~0u;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        g0 = __builtin_amdgcn_bitop3_b32(T[q], g0, mo[c * 32 + q], 0xD4);
                        g1 = __builtin_amdgcn_bitop3_b32(T[q], g1, mo[c * 32 + 8 + q], 0xD4);
                    }
                    g0 ^= mo[c * 32 + 16];
                    g1 ^= mo[c * 32 + 17];
                    nerr[c] += __builtin_popcount((pv1 & g1) | (pv0 & g0));
                }
            } else {
                uint32_t e[2] = {mo[8 * 32 + 0], mo[8 * 32 + 1]};
#pragma unroll
                for (int bv = 0; bv < 2; bv++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        uint32_t g = ~0u;
#pragma unroll
                        for (int q = 0; q < 8; q++) g = __builtin_amdgcn_bitop3_b32(T[q], g, mo[(bv * 4 + i) * 32 + q], 0xD4);
                        e[bv] ^= g & mo[(bv * 4 + i) * 32 + 8];
                    }
                nerr[0] += __builtin_popcount(((pbit & e[1]) | (~pbit & e[0])) & valid);
            }
        }
    };

#pragma unroll 1
    for (unsigned t = 0; t < tk.L; t += 2) {
        const unsigned u = (t & 31u) * 64 + lane;
        const uint32_t x0 = ring[u + ka], x1 = ring[u + ka + 64], y0 = ring[u + ta], y1 = ring[u + ta + 64];
        const uint32_t p0 = x0 ^ y0, p1 = x1 ^ y1;
        ring[u] = p0; ring[u + 64] = p1;
        ring[u + 32 * 64] = p0; ring[u + 33 * 64] = p1;
        const uint32_t s0 = t >= tk.last_len ? ~0u : 0u, s1 = t + 1 >= tk.last_len ? ~0u : 0u;
        prefetch();
        lutopt256_step_parked_ber(a, pa, b, pb, cnt);
        compare(p0, __builtin_amdgcn_bitop3_b32(vm_all, vm_last, s0, 0x70));      // a & ~(b & c)
        prefetch();
        lutopt256_step_parked_ber(b, pb, a, pa, cnt);
        compare(p1, __builtin_amdgcn_bitop3_b32(vm_all, vm_last, s1, 0x70));
    }
#undef BBB_SLOAD8
#undef BBB_SLOAD8_AT
#undef BBB_SWAIT2
#undef BBB_SWAIT2_AFTER
#undef BBB_SWAIT_AFTER
#pragma unroll
    for (int c = 0; c < NC; c++) {
        unsigned long long e64 = nerr[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e64 += __shfl_xor(e64, off, 64);
        if (lane == 0 && c < tk.ncfg) {
            if (wave == 0) atomicAdd(&counters[2 * c], tk.nbits);
            if (e64) atomicAdd(&counters[2 * c + 1], e64);
        }
    }
    if (tk.save) {
        // L is even: the state of the NEXT sample is in a / pa.  Every lane rewrites its own words of the buffers it read.
#define BBB_SAVE(p) { uint32_t v_; BBB_ACC_READ(v_, pa[p]); planes[(size_t)(p) * nlanes + LG] = v_; }
        LUTOPT256_FOR_PARKED_BER(BBB_SAVE)
#undef BBB_SAVE
#pragma unroll
        for (int p = 0; p < 256; p++)
            if (!lutopt256_ber_is_parked(p)) planes[(size_t)p * nlanes + LG] = a[p];
        for (int i = 0; i < tk.k; i++) prbs_planes[(size_t)i * nlanes + LG] = ring[((tk.L - 1u - (unsigned)i) & 31u) * 64 + lane];
    }
    (void)mk;
}

// Host side of the fused kernel.  The instances are spread over several translation units (ber_kernels*.hip: hipcc spends
// about a minute on each) and reached through ber_fused_go<MODE, NC>; a custom-matrix library (custom_fill_template.hip, ONE
// unit, built at run time) instantiates a shorter list.
template <int MODE, int NC>
void ber_fused_go(const BerMasks &mk, uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialF &tf, unsigned nlanes,
                  unsigned long long *d_counters, hipStream_t st);
#define BBB_BER_DEFINE_GO(MODE, NC)                                                                                              \
    template <>                                                                                                                  \
    void ber_fused_go<MODE, NC>(const BerMasks &mk, uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialF &tf,   \
                                unsigned nlanes, unsigned long long *d_counters, hipStream_t st) {                               \
        hipLaunchKernelGGL((ber256_fused_kernel<MODE, NC>), dim3(nlanes / 64), dim3(64), 0, st, mk, d_planes, d_prbs_planes, tf, \
                           nlanes, d_counters);                                                                                  \
    }

#define BBB_BER_DECLARE_GO(MODE, NC)                                                                                              \
    template <>                                                                                                                   \
    void ber_fused_go<MODE, NC>(const BerMasks &mk, uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialF &tf,    \
                                unsigned nlanes, unsigned long long *d_counters, hipStream_t st);
BBB_BER_DECLARE_GO(kBerFast, 1) BBB_BER_DECLARE_GO(kBerFast, 4) BBB_BER_DECLARE_GO(kBerFast, 12)
BBB_BER_DECLARE_GO(kBerPair, 1) BBB_BER_DECLARE_GO(kBerPair, 4) BBB_BER_DECLARE_GO(kBerPair, 12) BBB_BER_DECLARE_GO(kBerGeneral, 1)
#ifndef BBB_BER_FEW_INSTANCES
BBB_BER_DECLARE_GO(kBerFast, 2) BBB_BER_DECLARE_GO(kBerFast, 6) BBB_BER_DECLARE_GO(kBerFast, 8) BBB_BER_DECLARE_GO(kBerFast, 10)
BBB_BER_DECLARE_GO(kBerFast, 11)
#endif

#if BBB_BER_PART == 0
// counters: [ncfg][2] contiguous.  All trials of the group share t[0]'s stream geometry.
int ber256_launch(uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialDev *t, int ncfg, unsigned nlanes,
                  unsigned long long *d_counters, hipStream_t st) {
    if (ncfg < 1 || ncfg > kMaxCfg) return fail(BBB_EINVAL, "bad trial group size");
    if (t[0].L & 1) return fail(BBB_EINVAL, "segment length must be even");
    // thresholds strictly inside (0, 256) and the constant term of each bit value's parity
    int nthr[kMaxCfg][2], inv[kMaxCfg][2], thr[kMaxCfg][2][4];
    for (int c = 0; c < ncfg; c++)
        for (int bv = 0; bv < 2; bv++) {
            int n = 0, iv = 0;
            for (int i = 0; i < t[c].nthr[bv]; i++) {
                const int th = t[c].thr[bv][i];
                if (th <= 0) { iv ^= 1; continue; }     // [T >= 0] is always true
                if (th >= 256) continue;                // [T >= 256] never
                thr[c][bv][n++] = th;
            }
            nthr[c][bv] = n;
            inv[c][bv] = iv;
        }
    // the straight-line multi-setting kernel needs exactly one threshold per bit value
    bool simple = true;
    for (int c = 0; c < ncfg; c++)
        for (int bv = 0; bv < 2; bv++) simple = simple && nthr[c][bv] == 1;
    // fast path: bit 0 errs when T >= thr0, bit 1 when T < thr1, and thr0 - (256 - thr1) is 0 or 1 (the two
    // thresholds mirror each other up to the rounding of amp / noise_var): one comparison on X = bit ? ~T : T,
    // X >= thrx (bit = 1) / X >= thrx + strict0 (bit = 0)
    bool fast = simple;
    int thrx[kMaxCfg], strict0[kMaxCfg];
    for (int c = 0; c < ncfg && fast; c++) {
        const int d = thr[c][0][0] - (256 - thr[c][1][0]);
        fast = inv[c][0] == 0 && inv[c][1] == 1 && (d == 0 || d == 1);
        thrx[c] = 256 - thr[c][1][0];
        strict0[c] = d;
    }
    static const bool no_fast = env_knob("BBB_BER_NO_FAST", 0) != 0;     // (A/B timing of the two forms; -DBBB_EXPERIMENTS only)
    if (!simple && ncfg != 1) return fail(BBB_EINVAL, "grouped trials must be single-threshold");
    const int mode = fast && !no_fast ? kBerFast : simple ? kBerPair : kBerGeneral;
    // the scalar masks of the comparators: bit q of a threshold as 0 / ~0
    BerMasks mk{};
    auto bits = [](uint32_t *dst, int th) {
        for (int q = 0; q < 8; q++) dst[q] = (th >> q) & 1 ? ~0u : 0u;
    };
    TrialF tf{};
    tf.k = t[0].prbs_k; tf.tap = t[0].prbs_tap; tf.ncfg = ncfg; tf.L = t[0].L; tf.G = t[0].G; tf.nbits = t[0].nbits;
    tf.save = (t[0].flags & kTrialSaveState) != 0;
    tf.last_len = (t[0].flags & kTrialLastLen) ? t[0].last_len : (uint32_t)(t[0].nbits - (t[0].G - 1) * (uint64_t)t[0].L);
    if (mode == kBerGeneral) {
        for (int bv = 0; bv < 2; bv++) {
            for (int i = 0; i < 4; i++)
                if (i < nthr[0][bv]) {
                    bits(mk.m[bv * 4 + i], thr[0][bv][i]);
                    mk.m[bv * 4 + i][8] = ~0u;
                }
            mk.m[8][bv] = inv[0][bv] ? ~0u : 0u;
        }
        ber_fused_go<kBerGeneral, 1>(mk, d_planes, d_prbs_planes, tf, nlanes, d_counters, st);
    } else {
        // (instances exist for some group sizes: the next larger one runs copies of the last setting, whose counters are never read)
        for (int c = 0; c < kMaxCfg; c++) {
            const int s = c < ncfg ? c : ncfg - 1;
            if (mode == kBerFast) {
                bits(mk.m[c], thrx[s]);
                mk.m[c][8] = strict0[s] == 0 ? ~0u : 0u;
            } else {
                bits(mk.m[c], thr[s][0][0]);
                bits(mk.m[c] + 8, thr[s][1][0]);
                mk.m[c][16] = inv[s][0] ? ~0u : 0u;
                mk.m[c][17] = inv[s][1] ? ~0u : 0u;
            }
        }
#define BBB_GO(MODE, NC) ber_fused_go<MODE, NC>(mk, d_planes, d_prbs_planes, tf, nlanes, d_counters, st)
        if (mode == kBerFast) {
#ifdef BBB_BER_FEW_INSTANCES
            if (ncfg <= 1) BBB_GO(kBerFast, 1); else if (ncfg <= 4) BBB_GO(kBerFast, 4); else BBB_GO(kBerFast, 12);
#else
            if (ncfg <= 1) BBB_GO(kBerFast, 1); else if (ncfg <= 2) BBB_GO(kBerFast, 2); else if (ncfg <= 4) BBB_GO(kBerFast, 4);
            else if (ncfg <= 6) BBB_GO(kBerFast, 6); else if (ncfg <= 8) BBB_GO(kBerFast, 8); else if (ncfg <= 10) BBB_GO(kBerFast, 10);
            else if (ncfg <= 11) BBB_GO(kBerFast, 11); else BBB_GO(kBerFast, 12);
#endif
        } else {
            if (ncfg <= 1) BBB_GO(kBerPair, 1); else if (ncfg <= 4) BBB_GO(kBerPair, 4); else BBB_GO(kBerPair, 12);
        }
#undef BBB_GO
    }
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}
#endif   // BBB_BER_PART == 0

// the instances of this translation unit
#ifdef BBB_BER_FEW_INSTANCES
BBB_BER_DEFINE_GO(kBerFast, 1) BBB_BER_DEFINE_GO(kBerFast, 4) BBB_BER_DEFINE_GO(kBerFast, 12)
BBB_BER_DEFINE_GO(kBerPair, 1) BBB_BER_DEFINE_GO(kBerPair, 4) BBB_BER_DEFINE_GO(kBerPair, 12) BBB_BER_DEFINE_GO(kBerGeneral, 1)
#elif BBB_BER_PART == 0
BBB_BER_DEFINE_GO(kBerFast, 11) BBB_BER_DEFINE_GO(kBerFast, 1) BBB_BER_DEFINE_GO(kBerGeneral, 1)
#elif BBB_BER_PART == 1
BBB_BER_DEFINE_GO(kBerFast, 12) BBB_BER_DEFINE_GO(kBerFast, 2) BBB_BER_DEFINE_GO(kBerPair, 1)
#elif BBB_BER_PART == 2
BBB_BER_DEFINE_GO(kBerFast, 10) BBB_BER_DEFINE_GO(kBerFast, 4) BBB_BER_DEFINE_GO(kBerPair, 4)
#elif BBB_BER_PART == 3
BBB_BER_DEFINE_GO(kBerFast, 8) BBB_BER_DEFINE_GO(kBerFast, 6) BBB_BER_DEFINE_GO(kBerPair, 12)
#endif

}  // namespace bbb
