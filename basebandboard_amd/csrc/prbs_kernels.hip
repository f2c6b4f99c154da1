// prbs_kernels.hip -- PRBS-k generation / checking / error-detector kernels for gfx950.
//
// Reference semantics: gateware/bbb/prbs.py
//   PRBS              :17-35   bit = s[k-1] ^ s[tap-1]; s = (s << 1 | bit) mod 2^k; reset s = 1
//   TAPS              :14
//   PRBSErrorDetector :38-99
//
// GPU formulation (not a translation: the reference emits one bit per clock).
// The emitted stream obeys b[t] = b[t-k] ^ b[t-tap]; squaring over GF(2) gives
// b[t] = b[t - k*2^m] ^ b[t - tap*2^m] for every m.  With m = 13 the lags are multiples of
// 128 64-bit words, so if a wave lays the stream out in ROWS of 128 words (lane l owns words
// 2l, 2l+1 of every row = one 16-byte store per lane, 1 KiB per wave instruction) then
//     row[q] = row[q-k] ^ row[q-tap]            -- purely lane-local, two 64-bit XORs per 16 B.
// Each wave owns a contiguous region of rows, keeps a k-row window in registers and streams.
// The first k rows of a region are built in LDS by the same identity at m = 6..12 (word lags
// k*2^j, tap*2^j), starting from k words that lanes derive from the LFSR state jumped to the
// region start with precomputed powers T^(2^i) of the LFSR companion matrix.
//
// Roofline: HBM.  fill writes 1/8 B per bit, check reads 1/8 B per bit; ~0.4 VALU op per byte.
#include "bbb_common.hpp"
#include "gf2.hpp"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace bbb {

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int tap_of(int k) {
    return k == 7 ? 6 : k == 9 ? 5 : k == 11 ? 9 : k == 15 ? 14 : k == 20 ? 3 : k == 23 ? 18 : k == 31 ? 28 : 0;
}
static int k_index(int k) {
    switch (k) { case 7: return 0; case 9: return 1; case 11: return 2; case 15: return 3;
                 case 20: return 4; case 23: return 5; case 31: return 6; default: return -1; }
}

// T^(2^i) for i = 0..63 as k rows of 32-bit masks (row r: new bit r = parity(row & s)).
struct PrbsPowTable { uint32_t rows[64][32]; };
__device__ PrbsPowTable d_prbs_pow[7];

static const PrbsPowTable &host_pow_table(int k) {
    static PrbsPowTable tabs[7];
    static std::once_flag once[7];
    const int ki = k_index(k);
    std::call_once(once[ki], [&] {
        GF2Powers pw(prbs_matrix(k, tap_of(k)));
        for (int i = 0; i < 64; i++) {
            const GF2Mat &m = pw.pow2(i);
            for (int r = 0; r < 32; r++) tabs[ki].rows[i][r] = r < k ? (uint32_t)m.row(r)[0] : 0u;
        }
    });
    return tabs[ki];
}

static int upload_pow_table(int k) {
    static std::mutex mu;
    static bool done[64][7];
    int dev = 0;
    BBB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    const int ki = k_index(k);
    if (dev < 64 && done[dev][ki]) return BBB_OK;
    const PrbsPowTable &t = host_pow_table(k);
    BBB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_prbs_pow), &t, sizeof t, sizeof(PrbsPowTable) * (size_t)ki));
    if (dev < 64) done[dev][ki] = true;
    return BBB_OK;
}

int prbs_state_at_host(int k, uint64_t init_state, uint64_t nbits, uint64_t *state) {
    if (!tap_of(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    const PrbsPowTable &t = host_pow_table(k);
    uint32_t s = (uint32_t)init_state;
    for (int i = 0; nbits; i++, nbits >>= 1) {
        if (!(nbits & 1)) continue;
        uint32_t ns = 0;
        for (int r = 0; r < k; r++) ns |= (uint32_t)(__builtin_popcount(t.rows[i][r] & s) & 1) << r;
        s = ns;
    }
    *state = s;
    return BBB_OK;
}

// ---------------------------------------------------------------------------------------------
// Streaming generator / checker.  One wave per block; block b owns rows [b*rpw, (b+1)*rpw).
// A row is RW = 64*WPL words; lane l owns words WPL*l .. WPL*l+WPL-1 of every row (8- or 16-byte
// accesses).  Row lag identity: row[q] = row[q-K] ^ row[q-TAP]  (bit lags K*RW*64, TAP*RW*64).
// ---------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ uint32_t lfsr_matvec(const uint32_t *__restrict rows, uint32_t s) {
    uint32_t ns = 0;
#pragma unroll
    for (int r = 0; r < K; r++) ns |= (uint32_t)(__builtin_popcount(rows[r] & s) & 1) << r;
    return ns;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int WPL> struct LaneWords;
template <> struct LaneWords<1> { typedef u32x2 type; };
template <> struct LaneWords<2> { typedef u32x4 type; };
__device__ __forceinline__ uint32_t popc_words(u32x2 v) { return (uint32_t)(__builtin_popcount(v.x) + __builtin_popcount(v.y)); }
__device__ __forceinline__ uint32_t popc_words(u32x4 v) {
    return (uint32_t)(__builtin_popcount(v.x) + __builtin_popcount(v.y) + __builtin_popcount(v.z) + __builtin_popcount(v.w));
}
// a ^= b, in place: the tied operand stops the compiler from double-buffering the K-row window
// (which would need 2 x 4K registers and spill)
__device__ __forceinline__ void xor_inplace(u32x2 &a, const u32x2 &b) {
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.x) : "v"(b.x));
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.y) : "v"(b.y));
}
__device__ __forceinline__ void xor_inplace(u32x4 &a, const u32x4 &b) {
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.x) : "v"(b.x));
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.y) : "v"(b.y));
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.z) : "v"(b.z));
    asm("v_xor_b32 %0, %0, %1" : "+v"(a.w) : "v"(b.w));
}

// The checker's row loads are NON-TEMPORAL: every byte is read once, and with the hint the read stream runs at 6.3-6.4 TB/s
// instead of 5.5-5.6 on a clean buffer, and right after a fill it trades fewer of the fill's dirty lines in the
// memory-side cache for lines it will never read again (0.244 against 0.277 ms for 1.25 GB; profiles/README.md, round 2).
// -DBBB_PRBS_CHECK_PLAIN (experiments build) restores plain loads for the A/B; -DBBB_PRBS_FILL_NT makes the generator's
// stores non-temporal (no gain in round 1).
template <typename T>
__device__ __forceinline__ T check_load(const T *p) {
    return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void fill_store(T *p, const T &v) {
    *p = v;
}

// LW = true keeps the K-row window in (lane-private) LDS instead of registers: the checker then
// only holds its in-flight loads in registers and can use 16-byte accesses without spilling.
template <int K, bool CHECK, int WPL, bool LW>
__global__ void __launch_bounds__(64, 2)   // >= 2 waves per SIMD: at most 256 registers
prbs_stream_kernel(int ki, u64 init_state, u64 first_bit, u64 nbits, u64 nwords, u64 rows_per_wave,
                   u64 *__restrict buf, u64 *__restrict nerr, int nt_stores) {
    typedef typename LaneWords<WPL>::type lw_t;
    constexpr int TAP = tap_of(K);
    constexpr int RW = 64 * WPL;                 // words per row
    constexpr int LEVELS = WPL == 2 ? 7 : 6;     // RW = 2^LEVELS
    constexpr uint32_t SMASK = (uint32_t)((1ull << K) - 1ull);
    __shared__ __attribute__((aligned(16))) u64 X[K * RW];
    const int lane = threadIdx.x;
    const u64 row0 = (u64)blockIdx.x * rows_per_wave;
    const u64 word0 = row0 * RW;
    if (word0 >= nwords) return;
    const PrbsPowTable &pw = d_prbs_pow[ki];

    // 1. LFSR state at the first bit of the K rows that PRECEDE this region (so that every pass of
    //    the main loop, including the first, is "advance the window by K rows, then emit").  The
    //    sequence has period 2^K - 1 (maximal length), which extends it to negative positions.
    //    s = T^(t0) * init, lane r evaluating row r; the candidate rows of T^(2^i) are fetched 16 at
    //    a time (four memory round trips, not one per set bit).
    {
    constexpr u64 PERIOD = (1ull << K) - 1ull;
    constexpr u64 BACK = (u64)K * RW * 64;                       // bits in K rows
    constexpr u64 WRAP = ((BACK + PERIOD - 1) / PERIOD) * PERIOD;  // multiple of the period >= BACK
    const u64 t0 = (first_bit % PERIOD) + (word0 * 64) % PERIOD + (WRAP - BACK);
    uint32_t s = (uint32_t)init_state;
#pragma unroll 1
    for (int i0 = 0; i0 < 64; i0 += 16) {
        if (((t0 >> i0) & 0xffffull) == 0) continue;
        uint32_t myrow[16];
#pragma unroll
        for (int i = 0; i < 16; i++) myrow[i] = pw.rows[i0 + i][lane & 31];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if ((t0 >> (i0 + i)) & 1) s = (uint32_t)__ballot(lane < K && (__builtin_popcount(myrow[i] & s) & 1));
        }
    }
    // 2. lane i < K forms word i of the region: jump a further 64*i bits, then clock 64 times.
    if (lane < K) {
        uint32_t si = s;
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const uint32_t sj = lfsr_matvec<K>(pw.rows[6 + m], si);
            si = ((lane >> m) & 1) ? sj : si;
        }
        u64 w = 0;
        for (int j = 0; j < 64; j++) {
            const uint32_t bit = ((si >> (K - 1)) ^ (si >> (TAP - 1))) & 1u;
            si = ((si << 1) | bit) & SMASK;
            w |= (u64)bit << j;
        }
        X[lane] = w;
    }
    }
    __syncthreads();
    // 3. grow the known prefix K -> K*RW words: level j uses word lags K*2^j and TAP*2^j.
    {
        int known = K;
#pragma unroll 1
        for (int j = 0; j < LEVELS; j++) {
            const int lagk = K << j, lagt = TAP << j, target = K << (j + 1);
            while (known < target) {
                const int cnt = min(lagt, target - known);
                for (int base = 0; base < cnt; base += 64) {
                    const int o = base + lane;
                    if (o < cnt) X[known + o] = X[known + o - lagk] ^ X[known + o - lagt];
                }
                known += cnt;
                __syncthreads();
            }
        }
    }
    // 4. register window: V[q] = this lane's words of row q, q < K (LW: stays in LDS, row q of
    //    this lane at Xv[q*64 + lane]).
    lw_t *Xv = reinterpret_cast<lw_t *>(X);
    lw_t V[LW ? 1 : K];
    if (!LW) {
#pragma unroll
        for (int q = 0; q < K; q++) V[q] = Xv[q * 64 + lane];
    }

    const u64 rows_total = (nwords - word0 + RW - 1) / RW;
    const u64 nrows = rows_total < rows_per_wave ? rows_total : rows_per_wave;
    const u64 last_word = nwords - 1;
    const u64 last_mask = (nbits & 63) ? ((1ull << (nbits & 63)) - 1ull) : ~0ull;
    u64 errs = 0;

    constexpr int DB = WPL == 2 ? 8 : 16;               // rows per batch of loads (checker)
    constexpr int NB = (K + DB - 1) / DB;
    for (u64 q0 = 0; q0 < nrows; q0 += K) {
        const u64 wbase = word0 + q0 * RW + (u64)(WPL * lane);
        u64 *rowp = buf + (word0 + q0 * RW);      // wave-uniform row base: scalar base + lane offset addressing
        // fast path: all K rows of this pass lie strictly below the stream's last word --
        // no bounds or tail-mask logic
        if (q0 + K <= nrows && word0 + (q0 + K) * RW <= last_word) {
            if (CHECK) {
                // loads run one batch ahead of the compares (two register batches, static indices)
                lw_t D[2][DB];
#pragma unroll
                for (int i = 0; i < DB && i < K; i++) D[0][i] = check_load(reinterpret_cast<const lw_t *>(rowp + (u64)i * RW) + lane);
                // LW: the window makes one LDS round trip per pass (K reads issued back to back, the
                // in-place recurrence in registers, K writes) so that no register state is carried
                // around the loop
                if (!LW) {
#pragma unroll
                    for (int i = 0; i < K; i++) xor_inplace(V[i], V[(i - TAP + K) % K]);   // row[q] = row[q-K] ^ row[q-TAP]
                }
                uint32_t e32 = 0;
                // LW: the window streams through registers.  New row i = old row i ^ (i < TAP ? old row
                // i+K-TAP : new row i-TAP); old row j is fetched from LDS PD iterations before its first
                // use (iteration j-(K-TAP), or j), new rows are kept while a later row needs them.
                constexpr int PD = 6;
                constexpr int LAG = K - TAP;
                lw_t O[LW ? K : 1], N[LW ? K : 1];
                if (LW) {
#pragma unroll
                    for (int j = 0; j < K; j++)
                        if ((j >= LAG ? j - LAG : j) < PD) O[j] = Xv[j * 64 + lane];
                }
#pragma unroll
                for (int bidx = 0; bidx < NB; bidx++) {
                    if (bidx + 1 < NB) {
#pragma unroll
                        for (int i = 0; i < DB; i++)
                            if ((bidx + 1) * DB + i < K)
                                D[(bidx + 1) & 1][i] = check_load(reinterpret_cast<const lw_t *>(rowp + (u64)((bidx + 1) * DB + i) * RW) + lane);
                    }
#pragma unroll
                    for (int i = 0; i < DB; i++) {
                        const int r = bidx * DB + i;
                        if (r < K) {
                            if (LW) {
#pragma unroll
                                for (int j = 0; j < K; j++)          // rows whose first use is PD iterations ahead
                                    if ((j >= LAG ? j - LAG : j) == r + PD) O[j] = Xv[j * 64 + lane];
                                N[r] = O[r] ^ (r < TAP ? O[(r + LAG) % K] : N[(r - TAP + K) % K]);
                                Xv[r * 64 + lane] = N[r];
                                e32 += popc_words(D[bidx & 1][i] ^ N[r]);
                            } else {
                                e32 += popc_words(D[bidx & 1][i] ^ V[r]);
                            }
                        }
                    }
                }
                errs += e32;
            } else if (LW) {
                lw_t W[K];
#pragma unroll
                for (int i = 0; i < K; i++) W[i] = Xv[i * 64 + lane];
#pragma unroll
                for (int i = 0; i < K; i++) W[i] ^= W[(i - TAP + K) % K];
#pragma unroll
                for (int i = 0; i < K; i++) Xv[i * 64 + lane] = W[i];
                if (nt_stores) {
#pragma unroll
                    for (int i = 0; i < K; i++) __builtin_nontemporal_store(W[i], reinterpret_cast<lw_t *>(rowp + (u64)i * RW) + lane);
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) fill_store(reinterpret_cast<lw_t *>(rowp + (u64)i * RW) + lane, W[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) xor_inplace(V[i], V[(i - TAP + K) % K]);
                // nt_stores (bbb_prbs_fill_hint, BBB_PRBS_WILL_READ_BACK): non-temporal stores leave the memory-side cache
                // clean, so that a check right behind this fill does not pay for the write-backs of its last 256 MiB (the
                // fill itself is ~15 % slower that way, which is why it is the caller's choice: DESIGN.md, PRBS)
                if (nt_stores) {
#pragma unroll
                    for (int i = 0; i < K; i++) __builtin_nontemporal_store(V[i], reinterpret_cast<lw_t *>(rowp + (u64)i * RW) + lane);
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) fill_store(reinterpret_cast<lw_t *>(rowp + (u64)i * RW) + lane, V[i]);
                }
            }
            continue;
        }
        // tail pass: per-word bounds and the mask of the final partial word; the window goes back
        // to LDS so that rows can be indexed at run time
        if (LW) {
#pragma unroll
            for (int i = 0; i < K; i++) Xv[i * 64 + lane] = Xv[i * 64 + lane] ^ Xv[((i - TAP + K) % K) * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) xor_inplace(V[i], V[(i - TAP + K) % K]);
#pragma unroll
            for (int q = 0; q < K; q++) Xv[q * 64 + lane] = V[q];
        }
#pragma unroll 1
        for (int i = 0; i < K && q0 + i < nrows; i++) {
#pragma unroll
            for (int e = 0; e < WPL; e++) {
                const u64 w = wbase + (u64)i * RW + e;
                if (w > last_word) continue;
                u64 v = X[i * RW + WPL * lane + e];
                if (w == last_word) v &= last_mask;
                if (CHECK) {
                    u64 d = buf[w] ^ v;
                    if (w == last_word) d &= last_mask;
                    errs += (u64)__builtin_popcountll(d);
                } else {
                    buf[w] = v;
                }
            }
        }
    }
    if (CHECK) {
        // wave-level reduction, then ONE 64-bit atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) errs += __shfl_xor(errs, off, 64);
        if (lane == 0 && errs) atomicAdd(nerr, errs);
    }
}

// ---------------------------------------------------------------------------------------------
// The checker, regions read from their END to their start.  Why: the generator's waves write their
// regions in ascending order, so when a fill ends the 256 MiB memory-side cache holds the TAIL of
// every region as dirty lines.  A checker that starts at the heads misses on every line, and each miss
// evicts one of those dirty lines -- 0.25 GB of write-backs on top of the 1.25 GB it reads (measured:
// 3.7-3.9 TB/s instead of 5.5).  Started at the tails, the first 256 MiB are hits on exactly those
// lines and the write-backs replace reads instead of adding to them.  The count of mismatches does not
// depend on the order.
// The row recurrence runs backwards as well: row[p] = row[p+K] ^ row[p+K-TAP]; the window starts as
// the K rows that FOLLOW the region and retreats K rows per pass (same in-place update, descending).
// 8-byte lanes, register window.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(64, 2)
prbs_check_rev_kernel(int ki, u64 init_state, u64 first_bit, u64 nbits, u64 nwords, u64 rows_per_wave,
                      const u64 *__restrict buf, u64 *__restrict nerr) {
    constexpr int TAP = tap_of(K);
    constexpr int RW = 64;
    constexpr int LEVELS = 6;
    constexpr uint32_t SMASK = (uint32_t)((1ull << K) - 1ull);
    __shared__ __attribute__((aligned(16))) u64 X[K * RW];
    const int lane = threadIdx.x;
    const u64 row0 = (u64)blockIdx.x * rows_per_wave;
    const u64 word0 = row0 * RW;
    if (word0 >= nwords) return;
    const PrbsPowTable &pw = d_prbs_pow[ki];
    const u64 rows_total = (nwords - word0 + RW - 1) / RW;
    const long long nrows = (long long)(rows_total < rows_per_wave ? rows_total : rows_per_wave);

    {
    // 1. LFSR state at the first bit of the K rows that FOLLOW this region
    constexpr u64 PERIOD = (1ull << K) - 1ull;
    const u64 wtop = word0 + (u64)nrows * RW;
    const u64 t0 = (first_bit % PERIOD) + ((wtop % PERIOD) * 64) % PERIOD;
    uint32_t s = (uint32_t)init_state;
#pragma unroll 1
    for (int i0 = 0; i0 < 64; i0 += 16) {
        if (((t0 >> i0) & 0xffffull) == 0) continue;
        uint32_t myrow[16];
#pragma unroll
        for (int i = 0; i < 16; i++) myrow[i] = pw.rows[i0 + i][lane & 31];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if ((t0 >> (i0 + i)) & 1) s = (uint32_t)__ballot(lane < K && (__builtin_popcount(myrow[i] & s) & 1));
        }
    }
    // 2. lane i < K forms word i of the window, 3. the prefix grows K -> K*RW words (as in the ascending kernel)
    if (lane < K) {
        uint32_t si = s;
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const uint32_t sj = lfsr_matvec<K>(pw.rows[6 + m], si);
            si = ((lane >> m) & 1) ? sj : si;
        }
        u64 w = 0;
        for (int j = 0; j < 64; j++) {
            const uint32_t bit = ((si >> (K - 1)) ^ (si >> (TAP - 1))) & 1u;
            si = ((si << 1) | bit) & SMASK;
            w |= (u64)bit << j;
        }
        X[lane] = w;
    }
    }
    __syncthreads();
    {
        int known = K;
#pragma unroll 1
        for (int j = 0; j < LEVELS; j++) {
            const int lagk = K << j, lagt = TAP << j, target = K << (j + 1);
            while (known < target) {
                const int cnt = min(lagt, target - known);
                for (int base = 0; base < cnt; base += 64) {
                    const int o = base + lane;
                    if (o < cnt) X[known + o] = X[known + o - lagk] ^ X[known + o - lagt];
                }
                known += cnt;
                __syncthreads();
            }
        }
    }
    u32x2 *Xv = reinterpret_cast<u32x2 *>(X);
    u32x2 V[K];
#pragma unroll
    for (int q = 0; q < K; q++) V[q] = Xv[q * 64 + lane];

    const u64 last_word = nwords - 1;
    const u64 last_mask = (nbits & 63) ? ((1ull << (nbits & 63)) - 1ull) : ~0ull;
    u64 errs = 0;
    constexpr int DB = 16;
    constexpr int NB = (K + DB - 1) / DB;
    for (long long q0 = nrows - K; q0 > -(long long)K; q0 -= K) {       // the pass covers rows [q0, q0 + K)
        // retreat the window: V[i] = row q0 + i
#pragma unroll
        for (int i = K - 1; i >= 0; i--) xor_inplace(V[i], V[(i - TAP + K) % K]);
        if (q0 >= 0 && word0 + (u64)(q0 + K) * RW <= last_word) {
            const u64 *rowp = buf + (word0 + (u64)q0 * RW);
            u32x2 D[2][DB];
#pragma unroll
            for (int i = 0; i < DB && i < K; i++) D[0][i] = check_load(reinterpret_cast<const u32x2 *>(rowp + (u64)i * RW) + lane);
            uint32_t e32 = 0;
#pragma unroll
            for (int bidx = 0; bidx < NB; bidx++) {
                if (bidx + 1 < NB) {
#pragma unroll
                    for (int i = 0; i < DB; i++)
                        if ((bidx + 1) * DB + i < K)
                            D[(bidx + 1) & 1][i] = check_load(reinterpret_cast<const u32x2 *>(rowp + (u64)((bidx + 1) * DB + i) * RW) + lane);
                }
#pragma unroll
                for (int i = 0; i < DB; i++) {
                    const int r = bidx * DB + i;
                    if (r < K) e32 += popc_words(D[bidx & 1][i] ^ V[r]);
                }
            }
            errs += e32;
            continue;
        }
        // partial pass (the bottom of the region, or rows at the very end of the stream): per-word bounds
        __syncthreads();
#pragma unroll
        for (int q = 0; q < K; q++) Xv[q * 64 + lane] = V[q];
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            const long long r = q0 + i;
            if (r < 0 || r >= nrows) continue;
            const u64 w = word0 + (u64)r * RW + (u64)lane;
            if (w > last_word) continue;
            u64 d = buf[w] ^ X[i * RW + lane];
            if (w == last_word) d &= last_mask;
            errs += (u64)__builtin_popcountll(d);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) errs += __shfl_xor(errs, off, 64);
    if (lane == 0 && errs) atomicAdd(nerr, errs);
}

static int launch_check_rev(int k, int ki, u64 init_state, u64 first_bit, u64 nbits, const u64 *buf, u64 *nerr, hipStream_t st) {
    const u64 nwords = (nbits + 63) / 64;
    const u64 RW = 64;
    const u64 rows = (nwords + RW - 1) / RW;
    int dev = 0, ncu = 256, per_cu = 4;
    BBB_HIP(hipGetDevice(&dev));
    BBB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const void *fn = nullptr;
#define BBB_PRBS_FN(KK) case KK: fn = (const void *)prbs_check_rev_kernel<KK>; break;
    switch (k) { BBB_PRBS_FN(7) BBB_PRBS_FN(9) BBB_PRBS_FN(11) BBB_PRBS_FN(15) BBB_PRBS_FN(20) BBB_PRBS_FN(23) BBB_PRBS_FN(31) }
#undef BBB_PRBS_FN
    {
        static std::mutex mu;
        static int cached_per_cu[8] = {0};
        std::lock_guard<std::mutex> g(mu);
        int &slot = cached_per_cu[ki];
        if (slot == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
            slot = per_cu;
        }
        per_cu = slot;
    }
    static const int cap_env = env_knob("BBB_PRBS_WAVES_PER_CU", -1);
    // the SAME partition as the generator (4 waves per CU): only then is the end of a checker's region the end of a
    // generator's region, i.e. what the memory-side cache still holds (same-box sweep, gpurun_out/r02_prbs_rev_sweep.log:
    // check after fill 0.277 ms at 4 per CU, 0.287-0.293 at 3, 5, 6, 8; ascending 0.290-0.301)
    const int cap = cap_env > 0 ? cap_env : 4;
    if (per_cu > cap) per_cu = cap;
    const u64 target_waves = (u64)ncu * (u64)per_cu;           // exactly one resident generation of waves
    u64 rpw = (rows + target_waves - 1) / target_waves;
    const u64 min_rpw = 8 * (u64)k;
    if (rpw < min_rpw) rpw = min_rpw;
    const u64 nblocks = (rows + rpw - 1) / rpw;
    if (nblocks > 0x7fffffffull) return fail(BBB_EINVAL, "nbits too large");
    dim3 grid((unsigned)nblocks), block(64);
#define BBB_PRBS_CASE(KK)                                                                                          \
    case KK:                                                                                                       \
        hipLaunchKernelGGL((prbs_check_rev_kernel<KK>), grid, block, 0, st, ki, init_state, first_bit, nbits, nwords, \
                           rpw, buf, nerr);                                                                        \
        break;
    switch (k) {
        BBB_PRBS_CASE(7) BBB_PRBS_CASE(9) BBB_PRBS_CASE(11) BBB_PRBS_CASE(15)
        BBB_PRBS_CASE(20) BBB_PRBS_CASE(23) BBB_PRBS_CASE(31)
    }
#undef BBB_PRBS_CASE
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// words per lane: 1 (8-byte accesses, a 2K-register window, 15.5 KiB of LDS for K = 31) measured
// faster than 2 for both directions (profiles/r01_prbs_sweep.log); overridable for experiments
static int prbs_wpl(bool check) {
    static const int fill = env_knob("BBB_PRBS_FILL_WPL", 1);      // knobs exist only in -DBBB_EXPERIMENTS builds
    static const int chk = env_knob("BBB_PRBS_CHECK_WPL", 1);
    const int v = check ? chk : fill;
    return v == 1 ? 1 : 2;
}

template <bool CHECK, int WPL, bool LW>
static int launch_stream_w(int k, int ki, u64 init_state, u64 first_bit, u64 nbits, u64 *buf, u64 *nerr, hipStream_t st, int nt_stores) {
    const u64 nwords = (nbits + 63) / 64;
    const u64 RW = 64 * WPL;
    const u64 rows = (nwords + RW - 1) / RW;
    // exactly one resident generation of waves: every wave pays the bootstrap once, and a second,
    // partially filled generation would double the run time
    int dev = 0, ncu = 256, per_cu = 4;
    BBB_HIP(hipGetDevice(&dev));
    BBB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const void *fn = nullptr;
#define BBB_PRBS_FN(KK) case KK: fn = (const void *)prbs_stream_kernel<KK, CHECK, WPL, LW>; break;
    switch (k) { BBB_PRBS_FN(7) BBB_PRBS_FN(9) BBB_PRBS_FN(11) BBB_PRBS_FN(15) BBB_PRBS_FN(20) BBB_PRBS_FN(23) BBB_PRBS_FN(31) }
#undef BBB_PRBS_FN
    {   // occupancy of this instantiation, per K, cached under a lock (the same for every gfx950 device)
        static std::mutex mu;
        static int cached_per_cu[8] = {0};
        std::lock_guard<std::mutex> g(mu);
        int &slot = cached_per_cu[ki];
        if (slot == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
            slot = per_cu;
        }
        per_cu = slot;
    }
    // the generator is fastest with 4 waves per CU (stores need little latency hiding and fewer,
    // longer regions amortise the bootstrap); the checker takes every wave it can get
    static const int cap_env = env_knob("BBB_PRBS_WAVES_PER_CU", -1);
    const int cap = cap_env >= 0 ? cap_env : (CHECK ? 0 : 4);
    if (cap > 0 && per_cu > cap) per_cu = cap;
    const u64 target_waves = (u64)ncu * (u64)per_cu;
    u64 rpw = (rows + target_waves - 1) / target_waves;
    const u64 min_rpw = 8 * (u64)k;
    if (rpw < min_rpw) rpw = min_rpw;
    const u64 nblocks = (rows + rpw - 1) / rpw;
    if (nblocks > 0x7fffffffull) return fail(BBB_EINVAL, "nbits too large");
    dim3 grid((unsigned)nblocks), block(64);
#define BBB_PRBS_CASE(KK)                                                                                       \
    case KK:                                                                                                    \
        hipLaunchKernelGGL((prbs_stream_kernel<KK, CHECK, WPL, LW>), grid, block, 0, st, ki, init_state, first_bit, \
                           nbits, nwords, rpw, buf, nerr, nt_stores);                                           \
        break;
    switch (k) {
        BBB_PRBS_CASE(7) BBB_PRBS_CASE(9) BBB_PRBS_CASE(11) BBB_PRBS_CASE(15)
        BBB_PRBS_CASE(20) BBB_PRBS_CASE(23) BBB_PRBS_CASE(31)
    }
#undef BBB_PRBS_CASE
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

template <bool CHECK>
static int launch_stream(int k, u64 init_state, u64 first_bit, u64 nbits, u64 *buf, u64 *nerr, hipStream_t st, int nt_stores = 0) {
    const int ki = k_index(k);
    if (ki < 0) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (init_state == 0 || init_state >> k) return fail(BBB_EINVAL, "PRBS state must be in [1, 2^k)");
    if (nbits == 0) return BBB_OK;
    if ((uintptr_t)buf & 15) return fail(BBB_EINVAL, "packed PRBS buffer must be 16-byte aligned");
    if (first_bit + nbits < first_bit) return fail(BBB_EINVAL, "first_bit + nbits overflows");
    int rc = upload_pow_table(k);
    if (rc) return rc;
    // both directions: register window, 8-byte accesses (same-box A/B in profiles/README.md); the
    // LDS-window variants stay selectable for experiments
    // (both measured: profiles/r01_prbs_sweep.log)
    static const int lw_fill = env_knob("BBB_PRBS_FILL_LW", 0);
    static const int lw_chk = env_knob("BBB_PRBS_CHECK_LW", 0);
    const bool lw = CHECK ? lw_chk != 0 : lw_fill != 0;
    // the checker reads every region from its end (see prbs_check_rev_kernel); the ascending form stays for A/B
    static const int rev = env_knob("BBB_PRBS_CHECK_REV", 1);
    if (CHECK && rev && !lw && prbs_wpl(true) == 1) return launch_check_rev(k, ki, init_state, first_bit, nbits, buf, nerr, st);
    if (prbs_wpl(CHECK) == 1)
        return lw ? launch_stream_w<CHECK, 1, true>(k, ki, init_state, first_bit, nbits, buf, nerr, st, nt_stores)
                  : launch_stream_w<CHECK, 1, false>(k, ki, init_state, first_bit, nbits, buf, nerr, st, nt_stores);
    return lw ? launch_stream_w<CHECK, 2, true>(k, ki, init_state, first_bit, nbits, buf, nerr, st, nt_stores)
              : launch_stream_w<CHECK, 2, false>(k, ki, init_state, first_bit, nbits, buf, nerr, st, nt_stores);
}

int prbs_fill_launch(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits, uint64_t *dst, hipStream_t st, int nt_stores) {
    return launch_stream<false>(k, init_state, first_bit, nbits, (u64 *)dst, nullptr, st, nt_stores);
}
int prbs_check_launch(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits, const uint64_t *src,
                      uint64_t *nerr_dev, hipStream_t st) {
    return launch_stream<true>(k, init_state, first_bit, nbits, (u64 *)const_cast<uint64_t *>(src), (u64 *)nerr_dev, st);
}

// ---------------------------------------------------------------------------------------------
// PRBSErrorDetector, cycle exact (prbs.py:61-99): one independent detector per lane.
// Registers: bit_in (:66), prbs reset 1 (:62,:68), err_sr reset all-ones (:80-81), reload_ctr.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
prbs_detector_kernel(int k, int tap, const uint8_t *__restrict bits, u64 nstreams, u64 n,
                     uint8_t *__restrict err, uint8_t *__restrict reload) {
    const u64 sidx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    const u64 mask = (1ull << k) - 1ull;
    u64 prbs = 1, err_sr = mask;
    int bit_in = 0, reload_ctr = 0;
    const uint8_t *in = bits + sidx * n;
    for (u64 i = 0; i < n; i++) {
        const int feedback = (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1ull);
        const int rl = reload_ctr != 0;
        const int prbs_in = rl ? bit_in : feedback;              // :75-76
        const int e = bit_in != feedback;                        // :79
        const int err_count = __builtin_popcountll(err_sr);      // :86-87
        prbs = ((prbs << 1) | (u64)prbs_in) & mask;              // :68
        u64 esr = ((err_sr << 1) | (u64)e) & mask;               // :81
        if (err_count > k / 2) {                                 // :92-94 (overrides the shift)
            reload_ctr = k + k / 2;
            esr = 0;
        } else if (rl) {                                         // :95-97
            reload_ctr -= 1;
        }
        err_sr = esr;
        bit_in = in[i] & 1;                                      // :66
        if (err) err[sidx * n + i] = (uint8_t)(bit_in != (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1ull));
        if (reload) reload[sidx * n + i] = (uint8_t)(reload_ctr != 0);
    }
}

int prbs_detector_launch(int k, const uint8_t *bits, uint64_t nstreams, uint64_t n, uint8_t *err,
                         uint8_t *reload, hipStream_t st) {
    const int tap = tap_of(k);
    if (!tap) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (nstreams == 0 || n == 0) return BBB_OK;
    const u64 nblocks = (nstreams + 255) / 256;
    if (nblocks > 0x7fffffffull) return fail(BBB_EINVAL, "too many streams");
    hipLaunchKernelGGL(prbs_detector_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, k, tap, bits,
                       (u64)nstreams, (u64)n, err, reload);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

}  // namespace bbb
