// awgn_kernels.hip -- LUTOPT uniform generator + CLT Gaussian adder tree on gfx950.
//
// Reference semantics (paths relative to the reference checkout):
//   LUTOPT   gateware/bbb/rng.py:14-55    x'[r] = XOR of x[taps[r]], all rows from the old state
//   CLTGRNG  gateware/bbb/rng.py:58-108   log2(n)-level tree of y[j] = x[2j] - x[2j+1]
//            software/clt-grng/clt-grng-evaluate.py:8-16  (same tree in numpy)
//
// GPU formulation.  The reference is ONE generator emitting one sample per clock.  Here the
// sequential stream is cut into G contiguous segments of L samples; generator g starts from
// A^(first_step + g*L) * init (GF(2) jump-ahead), so the concatenation is bit-identical to the
// reference stream.  32 generators are packed per lane (bit-slicing: VGPR p = state bit p of
// 32 generators), a wave therefore advances 2048 generators per step with ~1000 V_BITOP3/XOR
// instructions (basebandboard_amd/gen_lutopt_kernel.py emits the straight-line network for the matrix).
//
// Kernels
//   seed_first/level_kernel  start states, radix 16: S[j*16^e + i] = (A^L)^(j*16^e) * S[i]
//   bitslice_kernel      [G][k bits] -> planes [k][lanes] (32x32 bit transposes)
//   awgn256_kernel       the hot kernel (n256 matrix of gateware/bbb/rng_recurrences.py:172-259)
//   awgn_generic_kernel  any k <= 512 / any taps, table driven, planes in global scratch
//   clt_tree_kernel      adder tree of caller-supplied words (clt-grng-evaluate.py loop body)
//
// Roofline of awgn256_kernel: integer VALU (about 33 lane-ops per 1-byte sample); the sample
// stream is 1 B/sample of HBM writes, issued as 16-byte stores, one per generator per 16 steps.
#include "bbb_common.hpp"
#include "bitslice_util.hpp"
#include "awgn_launch.hpp"

#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "gen/lutopt256_gen.inc"

namespace bbb {

// ---------------------------------------------------------------------------------------------
// Start states: S[g] = B^g * s0 with B = A^L, built radix 16 (radix 4 until round 2: twice the launches, and the
// chain of small levels -- 17 us each -- was half of an un-overlapped seeding).  Level e maps the first 16^e states
// through B^(j*16^e), j = 1..15:   S[j*16^e + i] = B^(j*16^e) * S[i].
// y = M x is evaluated four state bits at a time: a table holds, per nibble position n, the 16
// XOR-combinations of columns 4n..4n+3 of M ([k/4][chunks][16][C] words, built on the host, one
// table per (level, j)).  The big levels stage their table in LDS (32 KiB for k = 256) and every
// lane does k/4 lookups of W32 words; the first 256 states are produced by one block straight from
// the tables in global memory (each lane composes up to four jumps from the base-4 digits of its
// index).  States are stored word-major, S[w * stride + g], so that this kernel and the
// bit-slicing pass touch consecutive addresses from consecutive lanes.
//
// Table layout: an entry of W32 words is cut into chunks of C = 4, 2 or 1 words (the largest that
// divides W32; one 16-byte access each for C = 4); chunk zc of all 16 entries of nibble n is contiguous:
//   index(n, v, zc, zz) = ((n * (W32/C) + zc) * 16 + v) * C + zz
// so the 16 possible 16-byte LDS reads of one (n, zc) cover 256 consecutive bytes = every bank once:
// lanes reading different entries never conflict, lanes reading the same entry broadcast.
// ---------------------------------------------------------------------------------------------

// accumulates the nibbles [nlo, nhi) of x into y; `tab` points at the table slice of nibble nlo
template <int W32, typename TabPtr>
__device__ __forceinline__ void nibble_matvec_part(TabPtr tab, int nlo, int nhi, const uint32_t (&x)[W32],
                                                   uint32_t (&y)[W32]) {
    constexpr int C = (W32 % 4 == 0) ? 4 : (W32 % 2 == 0 ? 2 : 1);
    constexpr int NC = W32 / C;
    typedef uint32_t chunk_t __attribute__((ext_vector_type(C)));
    // two lookups are folded per accumulate: y ^= e0 ^ e1 is ONE V_BITOP3 (0x96) per word
#pragma unroll
    for (int w = 0; w < W32; w++) {
        uint32_t xw = x[w];
        asm volatile("" : "+v"(xw));            // keeps hipcc from extracting all k/4 nibbles up front (64 live registers)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const int n0 = w * 8 + q, n1 = n0 + 1;
            if (n0 < nlo || n0 >= nhi) continue;
            const uint32_t v0 = (xw >> (4 * q)) & 15u;
            const uint32_t v1 = (xw >> (4 * q + 4)) & 15u;
#pragma unroll
            for (int zc = 0; zc < NC; zc++) {
                const chunk_t e0 = *reinterpret_cast<const chunk_t *>(tab + (((n0 - nlo) * NC + zc) * 16 + v0) * C);
                chunk_t e1 = (chunk_t)(0);
                if (n1 < nhi) e1 = *reinterpret_cast<const chunk_t *>(tab + (((n1 - nlo) * NC + zc) * 16 + v1) * C);
#pragma unroll
                for (int zz = 0; zz < C; zz++)
                    y[zc * C + zz] = __builtin_amdgcn_bitop3_b32(y[zc * C + zz], e0[zz], e1[zz], 0x96);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep at most one state word's lookups (16 x 16 B) in flight
    }
}

// states 0..15 come from the host (15 sequential products with B: microseconds, overlapped with
// the previous launch); this kernel only stores them
struct Seed16 { uint32_t w[16][16]; };     // [state][word]; only the first W32 words of each are used
template <int W32>
__global__ void __launch_bounds__(64)
seed_store16_kernel(Seed16 s, unsigned long long G, unsigned long long stride, uint32_t *__restrict S) {
    const unsigned i = threadIdx.x;
    if (i >= 16 || i >= G) return;
#pragma unroll
    for (int w = 0; w < W32; w++) S[w * stride + i] = s.w[i][w];
}

// nibbles per staged piece of the table: ceil(nnib / parts), made even (lookups are folded in pairs)
__host__ __device__ inline int seed_part_nibbles(int nnib, int parts) {
    const int h = (nnib + parts - 1) / parts;
    return h + (h & 1);
}

// level e: blockIdx.y = j - 1
template <int W32>
__global__ void __launch_bounds__(256, 5)   // <= 96 registers: a block must fit beside the sample kernel's waves
seed_level_kernel(const uint32_t *__restrict tabs, int k, int e, unsigned long long G, unsigned long long stride,
                  uint32_t *__restrict S, int parts) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    const unsigned long long lo = 1ull << (4 * e);
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned j = blockIdx.y + 1;
    const unsigned long long dst = (unsigned long long)j * lo + i;
    const unsigned long long blk0 = (unsigned long long)j * lo + (unsigned long long)blockIdx.x * blockDim.x;
    if (blk0 >= G) return;                       // whole block beyond the last generator
    const int nnib = (k + 3) / 4;
    const int nt = nnib * 16 * W32;              // words; a multiple of 4 for every supported k
    const bool active = i < lo && dst < G;
    uint32_t x[W32], y[W32];
#pragma unroll
    for (int w = 0; w < W32; w++) { x[w] = active ? S[w * stride + i] : 0u; y[w] = 0u; }   // in flight while the table is staged
    // The table is staged in `parts` pieces (2: 16 KiB each for k = 256; 4: 8 KiB): a block of this kernel then fits
    // beside the four resident blocks of the sample kernel on a CU (4 x 32 KiB of the 160; the transmitter variant
    // holds 4 x 36 KiB, beside which only the 8 KiB form fits), so that prefetched seeding (bbb_awgn_prefetch) can
    // overlap the previous fill.
    if (parts == 0) {
        // no LDS at all: the lookups go to the table in global memory (32 KiB per level and digit: the vector cache and
        // L2 hold it) -- for seeding beside a kernel whose waves wait on the CU's LDS pipeline
        if (active) nibble_matvec_part<W32>(tabs + ((size_t)e * 15 + (j - 1)) * nt, 0, nnib, x, y);
    } else {
    const int half = seed_part_nibbles(nnib, parts);                  // even number of nibbles
    for (int part = 0; part < parts; part++) {
        const int nlo = part * half, nhi = (part + 1) * half < nnib ? (part + 1) * half : nnib;
        if (nlo >= nhi) break;
        if (part) __syncthreads();
        {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 *src = reinterpret_cast<const u4 *>(tabs + ((size_t)e * 15 + (j - 1)) * nt + (size_t)nlo * 16 * W32);
            u4 *dstp = reinterpret_cast<u4 *>(tab);
            const int n4 = (nhi - nlo) * 16 * W32 / 4;
            for (int base = 0; base < n4; base += 4 * 256) {     // 16 KiB = 4 x 16 B per thread
                u4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = base + u * 256 + (int)threadIdx.x;
                    if (q < n4) v[u] = src[q];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = base + u * 256 + (int)threadIdx.x;
                    if (q < n4) dstp[q] = v[u];
                }
            }
        }
        __syncthreads();
        if (active) nibble_matvec_part<W32>(tab, nlo, nhi, x, y);
    }
    }
    if (active) {
#pragma unroll
        for (int w = 0; w < W32; w++) S[w * stride + dst] = y[w];
    }
}

// The same start states without LDS and in ONE launch (round 5).  The 32 generators of a consumer lane are 64 L bits apart
// from each other, so ONE table-composed state per lane -- 65 408 instead of 2.09 M, from the jump tables in place (global memory:
// the vector cache holds the few KiB a wave touches) -- and 31 steps with the 31 x 31 matrix Q = B^64 (its columns are scalars:
// V_BFE_I32 + V_BITOP3 per column) give a lane's 32 states, a 32 x 32 bit transpose the planes.  ~2300 VALU instructions per
// lane, no LDS: beside the generators' seeding kernels, which live on the LDS pipeline, it costs them nothing (the table-driven
// form above took 65 us there and lengthened the tail kernel from 61 to 85: profiles/r05_ber_c_timeline.txt).
struct PrbsLaneJump { uint32_t qcol[32]; uint32_t first[16]; };      // qcol[c]: column c of Q (bit r = Q[r][c]); first[i] = B^i p0
// (a block of 256 consumer lanes; `block` = its index: the body of prbs_seed_lanes_kernel and of the extra blocks of seed_head_kernel)
__device__ __forceinline__ void prbs_seed_lanes_block(const uint32_t *__restrict tabs, const PrbsLaneJump &jp, int k, int levels,
                                                      unsigned long long G, unsigned nlanes, uint32_t *__restrict planes, unsigned block) {
    __shared__ uint32_t first[16];
    if (threadIdx.x < 16) first[threadIdx.x] = jp.first[threadIdx.x];
    __syncthreads();
    const unsigned long long LG = (unsigned long long)block * 256 + threadIdx.x;
    if (LG >= nlanes) return;
    const unsigned long long wave = LG >> 6;
    const unsigned lane = (unsigned)(LG & 63);
    const unsigned long long g0 = gen_index(wave, lane, 0);
    const int nnib = (k + 3) / 4, nt = nnib * 16;
    uint32_t x = first[g0 & 15];
    for (int e = 1; e < levels; e++) {
        const unsigned d = (unsigned)(g0 >> (4 * e)) & 15u;
        if (d) {
            const uint32_t *t = tabs + ((size_t)e * 15 + (d - 1)) * nt;
            uint32_t y = 0;
#pragma unroll
            for (int n = 0; n < 8; n++)
                if (n < nnib) y ^= t[n * 16 + ((x >> (4 * n)) & 15u)];
            x = y;
        }
    }
    uint32_t q[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        q[j] = g0 + 64ull * (unsigned)j < G ? x : 0u;
        uint32_t y = 0;
#pragma unroll
        for (int c = 0; c < 31; c++) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)x, c, 1);      // 0 or ~0: bit c of x
            y = __builtin_amdgcn_bitop3_b32(y, jp.qcol[c], m, 0x78);                // a ^ (b & c)
        }
        x = y;
    }
    transpose32(q);
#pragma unroll
    for (int p = 0; p < 32; p++)
        if (p < k) planes[(size_t)p * nlanes + LG] = q[p];
}

__global__ void __launch_bounds__(256)
prbs_seed_lanes_kernel(const uint32_t *__restrict tabs, PrbsLaneJump jp, int k, int levels, unsigned long long G, unsigned nlanes,
                       uint32_t *__restrict planes) {
    prbs_seed_lanes_block(tabs, jp, k, levels, G, nlanes, planes, blockIdx.x);
}

// the PRBS seeding that rides on a head launch (blocks [nhead, nhead + nblocks) of seed_head_kernel): nblocks = 0 -- none
struct PrbsRide {
    const uint32_t *tabs;
    uint32_t *planes;
    PrbsLaneJump jp;
    int k, levels;
    unsigned nlanes, nblocks;
};

// ---------------------------------------------------------------------------------------------
// Start states AND bit planes in TWO launches (round 5; the BER trial's own seeding: an isolated trial waits for this chain, and
// the chain above is seven launches -- store16, five levels of which three are a few thousand mat-vecs each and cost a launch
// and a table staging apiece, then a bit-slicing pass over 134 MB: 131 us, profiles/r05_base_ber_timeline.txt).
//   seed_head_kernel         S[0 .. 65535], packed.  Thread g composes the jumps of its three upper digits itself: digit 1 from
//                            the tables IN PLACE (a wave reads four of them, each lookup two lines of one), digits 2 and 3 -- the
//                            same for the whole block -- from LDS, both tables staged while the first mat-vec runs.  256 blocks,
//                            each redoes the 256 small mat-vecs in front of its own: a few microseconds of redundant work against
//                            two launches and two drains.
//   seed_tail_planes_kernel  one block per WAVE of the consumer = 2048 consecutive generators, eight per thread.  Above the first
//                            65536 it derives them, S[d 65536 + i] = B^(d 65536) S[i], d = 1 .. 31: digits 4 and 5 merged into
//                            ONE level of "top" tables (JumpPlan::d_top), so that everything there is one launch of independent
//                            mat-vecs and a block stages its table once for 2048 states (an eighth of the staging traffic of the
//                            256-state blocks above); below, it reads them.  Then it bit-slices them itself: the eight states of a
//                            thread become, per state word, 32 bytes (planes8_to_bytes: byte p = bit p of the eight), which meet
//                            the bytes of the lane's other three threads in LDS (the table's memory, dead by then) as whole
//                            plane words, stored by rows of 64 lanes.  The packed states above 65536 never reach memory and the
//                            bit-slicing launch with its 134 MB is gone.
// ---------------------------------------------------------------------------------------------
template <int W32>
__global__ void __launch_bounds__(256)
seed_head_kernel(const uint32_t *__restrict tabs, Seed16 s, int k, unsigned long long G, unsigned long long stride,
                 uint32_t *__restrict S, unsigned nhead, PrbsRide pr) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];      // two tables: digit 2's, digit 3's
    // The trial's PRBS start states ride on this launch as extra blocks (VALU only, no LDS to speak of: they share the CUs with the
    // head blocks, which wait on LDS and L2): no second stream, no event and no barrier packet between the seeding and the trial kernel
    if (blockIdx.x >= nhead) {
        prbs_seed_lanes_block(pr.tabs, pr.jp, pr.k, pr.levels, G, pr.nlanes, pr.planes, blockIdx.x - nhead);
        return;
    }
    const unsigned t = threadIdx.x, d0 = t & 15u, d1 = t >> 4, d2 = blockIdx.x & 15u, d3 = blockIdx.x >> 4;
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + t;
    if ((unsigned long long)blockIdx.x * 256 >= G) return;       // whole block beyond the last generator
    const int nnib = (k + 3) / 4;
    const int nt = nnib * 16 * W32;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    // the block's two tables on their way to LDS (registers first: digit 1's mat-vec runs while the loads are in flight)
    constexpr int kMaxPieces = (64 * 16 * W32 / 4 + 255) / 256;   // 16-byte pieces per thread and table for k <= 256
    u4 v2[kMaxPieces], v3[kMaxPieces];
    const int n4 = nt / 4;
    if (d2) {
        const u4 *src = reinterpret_cast<const u4 *>(tabs + ((size_t)2 * 15 + (d2 - 1)) * nt);
#pragma unroll
        for (int u = 0; u < kMaxPieces; u++) {
            const int q = u * 256 + (int)t;
            if (q < n4) v2[u] = src[q];
        }
    }
    if (d3) {
        const u4 *src = reinterpret_cast<const u4 *>(tabs + ((size_t)3 * 15 + (d3 - 1)) * nt);
#pragma unroll
        for (int u = 0; u < kMaxPieces; u++) {
            const int q = u * 256 + (int)t;
            if (q < n4) v3[u] = src[q];
        }
    }
    uint32_t x[W32], y[W32];
#pragma unroll
    for (int w = 0; w < W32; w++) x[w] = s.w[d0][w];
    if (d1) {
#pragma unroll
        for (int w = 0; w < W32; w++) y[w] = 0u;
        nibble_matvec_part<W32>(tabs + ((size_t)1 * 15 + (d1 - 1)) * nt, 0, nnib, x, y);
#pragma unroll
        for (int w = 0; w < W32; w++) x[w] = y[w];
    }
    if (d2 || d3) {                                 // (block-uniform: every thread of the block reaches the barrier)
        u4 *dst2 = reinterpret_cast<u4 *>(tab), *dst3 = reinterpret_cast<u4 *>(tab + nt);
#pragma unroll
        for (int u = 0; u < kMaxPieces; u++) {
            const int q = u * 256 + (int)t;
            if (q < n4 && d2) dst2[q] = v2[u];
            if (q < n4 && d3) dst3[q] = v3[u];
        }
        __syncthreads();
    }
    if (d2) {
#pragma unroll
        for (int w = 0; w < W32; w++) y[w] = 0u;
        nibble_matvec_part<W32>(tab, 0, nnib, x, y);
#pragma unroll
        for (int w = 0; w < W32; w++) x[w] = y[w];
    }
    if (d3) {
#pragma unroll
        for (int w = 0; w < W32; w++) y[w] = 0u;
        nibble_matvec_part<W32>(tab + nt, 0, nnib, x, y);
#pragma unroll
        for (int w = 0; w < W32; w++) x[w] = y[w];
    }
    if (g < G) {
#pragma unroll
        for (int w = 0; w < W32; w++) S[w * stride + g] = x[w];
    }
}

// k = 256 only (W32 = 8: the 32 KiB table holds four state words' worth of plane rows at a time)
__global__ void __launch_bounds__(256, 4)       // <= 128 registers: the 1022 blocks of a 1e9-bit trial are resident together, four per CU
seed_tail_planes_kernel(const uint32_t *__restrict top, unsigned long long G, unsigned long long stride, const uint32_t *__restrict S,
                        unsigned nlanes, uint32_t *__restrict planes) {
    constexpr int W32 = 8, nnib = 64, nt = nnib * 16 * W32;
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];       // 32 KiB: the block's table, then the exchange buffer
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const unsigned t = threadIdx.x, lane = t & 63u, jq = t >> 6;
    const unsigned long long wv = blockIdx.x, d = wv >> 5;               // 32 waves of the consumer per 65536 generators
    // generator r of this thread: bit position j = 8 jq + r of consumer lane (wv, lane)
    const unsigned long long g0 = gen_index(wv, lane, 8 * jq);           // (+ 64 per r)
    // acc[wq][i] byte q, bit r = bit 8 q + i of word wq of the thread's state r: what planes8_to_bytes makes of the eight words,
    // built up state by state (the eight states of a thread never exist side by side: 64 registers of accumulators instead of
    // 64 of states + the transposition's temporaries, and the loop over r stays a loop)
    uint32_t acc[W32][8];
#pragma unroll
    for (int w = 0; w < W32; w++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[w][i] = 0u;
    auto deposit = [&](const uint32_t (&y)[W32], unsigned r) {
#pragma unroll
        for (int w = 0; w < W32; w++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[w][i] |= ((y[w] >> i) & 0x01010101u) << r;
    };
    uint32_t xn[W32];
#pragma unroll
    for (int w = 0; w < W32; w++) xn[w] = g0 < G ? S[w * stride + (g0 & 65535ull)] : 0u;       // in flight while the table is staged
    if (d == 0) {
#pragma unroll 1
        for (unsigned r = 0; r < 8; r++) {
            uint32_t x[W32];
#pragma unroll
            for (int w = 0; w < W32; w++) x[w] = xn[w];
            if (r + 1 < 8) {
                const unsigned long long g = g0 + 64ull * (r + 1);
#pragma unroll
                for (int w = 0; w < W32; w++) xn[w] = g < G ? S[w * stride + g] : 0u;
            }
            deposit(x, r);
        }
    } else {
        {
            const u4 *src = reinterpret_cast<const u4 *>(top + (size_t)(d - 1) * nt);
            u4 *dstp = reinterpret_cast<u4 *>(tab);
            constexpr int n4 = nt / 4;
#pragma unroll
            for (int base = 0; base < n4; base += 4 * 256) {
                u4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = src[base + u * 256 + (int)t];
#pragma unroll
                for (int u = 0; u < 4; u++) dstp[base + u * 256 + (int)t] = v[u];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (unsigned r = 0; r < 8; r++) {
            uint32_t x[W32], y[W32];
#pragma unroll
            for (int w = 0; w < W32; w++) { x[w] = xn[w]; y[w] = 0u; }
            if (r + 1 < 8) {
                const unsigned long long g = g0 + 64ull * (r + 1);
#pragma unroll
                for (int w = 0; w < W32; w++) xn[w] = g < G ? S[w * stride + (g & 65535ull)] : 0u;
            }
            // y = M x by nibble lookups (nibble_matvec_part's scheme, eight 16-byte lookups in flight instead of sixteen: registers)
#pragma unroll
            for (int w = 0; w < W32; w++) {
                uint32_t xw = x[w];
                asm volatile("" : "+v"(xw));
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    const int n0 = w * 8 + q, n1 = n0 + 1;
                    const uint32_t v0 = (xw >> (4 * q)) & 15u, v1 = (xw >> (4 * q + 4)) & 15u;
#pragma unroll
                    for (int zc = 0; zc < 2; zc++) {
                        const u4 e0 = *reinterpret_cast<const u4 *>(tab + ((n0 * 2 + zc) * 16 + v0) * 4);
                        const u4 e1 = *reinterpret_cast<const u4 *>(tab + ((n1 * 2 + zc) * 16 + v1) * 4);
#pragma unroll
                        for (int zz = 0; zz < 4; zz++) y[zc * 4 + zz] = __builtin_amdgcn_bitop3_b32(y[zc * 4 + zz], e0[zz], e1[zz], 0x96);
                    }
                    if (q & 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
            deposit(y, r);                                               // (a generator beyond G: x = 0 gives y = 0)
        }
        __syncthreads();                                                 // the table is dead: its memory becomes the exchange buffer
    }
    // byte jq of the plane word (plane 32 wq + 8 q + i, lane) = byte q of acc[wq][i].  Four state words (128 plane rows of 64 lanes:
    // 32 KiB) per round.
    uint8_t *xb = reinterpret_cast<uint8_t *>(tab);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();                                       // the first round's rows have been read
#pragma unroll
        for (int wl = 0; wl < 4; wl++)
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    xb[(((unsigned)(wl * 32 + 8 * q + i)) * 64u + lane) * 4u + jq] = (uint8_t)(acc[half * 4 + wl][i] >> (8 * q));
        __syncthreads();
        // 128 rows of 64 words: thread (lane, jq) takes rows jq, jq + 4, ...
#pragma unroll 8
        for (int m = 0; m < 32; m++) {
            const unsigned row = jq + 4u * (unsigned)m;
            planes[(size_t)(half * 128 + row) * nlanes + wv * 64 + lane] = tab[row * 64u + lane];
        }
    }
}

// PRBS start states for the BER kernels (round 4): k <= 31, so a state is one word and a jump table at most 512 bytes; all of
// them (levels 1..6 x 15 digits: 45 KiB) sit in LDS and thread g composes the jumps of generator g digit by digit -- one launch
// (plus bitslice_kernel<1>) where the generic chain took seven on the side stream and held CUs beside the generator's own seeding.
__global__ void __launch_bounds__(256)
prbs_seed_states_kernel(const uint32_t *__restrict tabs, Seed16 s, int k, int levels, unsigned long long G, uint32_t *__restrict S) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];       // [e - 1][d - 1][ceil(k / 4) nibbles][16]  (W32 = 1, C = 1)
    __shared__ uint32_t first[16];
    const int nnib = (k + 3) / 4, nt = nnib * 16;
    for (int i = threadIdx.x; i < (levels - 1) * 15 * nt; i += blockDim.x) tab[i] = tabs[15 * nt + i];
    if (threadIdx.x < 16) first[threadIdx.x] = s.w[threadIdx.x][0];
    __syncthreads();
    // a block takes a contiguous stretch of generators, thread-strided: neighbouring lanes differ in the LOW digits only
    const unsigned long long per = 256ull * 8;
    for (unsigned long long g = (unsigned long long)blockIdx.x * per + threadIdx.x; g < G && g < (blockIdx.x + 1ull) * per; g += 256) {
        uint32_t x = first[g & 15];
        for (int e = 1; e < levels; e++) {
            const unsigned d = (unsigned)(g >> (4 * e)) & 15u;
            if (d) {
                const uint32_t *t = tab + ((e - 1) * 15 + (d - 1)) * nt;
                uint32_t y = 0;
#pragma unroll
                for (int n = 0; n < 8; n++)
                    if (n < nnib) y ^= t[n * 16 + ((x >> (4 * n)) & 15u)];
                x = y;
            }
        }
        S[g] = x;
    }
}

// Word-major packed states -> bit planes.  Thread (LG, wq) gathers word wq of the 32 generators of
// lane LG and transposes: planes[(32*wq + p) * nlanes + LG] bit j = state bit 32*wq+p of g(LG, j).
template <int W32>
__global__ void __launch_bounds__(256)
bitslice_kernel(const uint32_t *__restrict S, unsigned long long G, unsigned long long stride, unsigned nlanes, int k,
                uint32_t *__restrict planes) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long LG = t % nlanes;       // consecutive threads -> consecutive lanes
    const unsigned wq = (unsigned)(t / nlanes);
    if (wq * 32 >= (unsigned)k) return;
    const unsigned long long wave = LG >> 6;
    const unsigned lane = (unsigned)(LG & 63);
    uint32_t q[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const unsigned long long g = gen_index(wave, lane, j);
        q[j] = g < G ? S[wq * stride + g] : 0u;
    }
    transpose32(q);
#pragma unroll
    for (int p = 0; p < 32; p++)
        if (wq * 32 + p < (unsigned)k) planes[(size_t)(wq * 32 + p) * nlanes + LG] = q[p];
}

// ---------------------------------------------------------------------------------------------
// The hot kernel: n256, one wave per block, one wave per SIMD (the 256-plane state plus the
// update's temporaries need the whole 512-register file).
//
// Per round of 16 steps a lane leaves 16 bytes for each of its 32 generators:
//   step t    -> 8 count planes -> planes8_to_bytes (12 shift/BFI butterflies): word i byte q is
//                the sample of generator 8q+i -> LDS Z[t][i][lane]
//   round end -> for i = 0..7: the 16 words Z[0..15][i]; four 4x4 byte transposes (V_PERM_B32)
//                give, per q, the 16 consecutive sample bytes of generator 8q+i -> one
//                global_store_dwordx4 each.
// All LDS traffic is lane-private (dword index = row*64 + lane: conflict-free, no barriers).
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


// The transmitter's output fused into the sample kernel (tx.py:60-81; bitshaper.py:25-86): instead of the
// int8 noise stream the round end writes x = wrap12(bit_en * shaped + wrap12(g * noise_var)) as int16, 32
// bytes per generator and round, and the noise never goes through HBM.
//   * shaped[n] = T[ph][q] with ph = (n-17) & 7, q = data bits M-7..M, M = (n-17) >> 3 (tx_kernels.hip).  A
//     generator's 16 samples of a round start at n = first_sample + g L + 16 r; L is a multiple of 16, so
//     c0 = (n-17) & 7 is the same for every generator and round: sample e has phase (c0+e) & 7 and window
//     shift (c0+e) >> 3 in {0,1,2}.  The LDS table TT[q][j] = T[(c0+j) & 7][q] (16 bytes per q) gives the 8
//     phases of one window by ONE ds_read_b128; rows q0, q1, q2 of the three shifts and a per-dword select
//     (V_BFI with wave-uniform masks) give the 16 shaped samples as 8 packed pairs.
//   * |g * noise_var| <= 128 * 15 < 2048: the inner wrap12 never acts.  The outer one is done in 16-bit
//     arithmetic scaled by 16: TT holds 16 * (shaped - 128 noise_var) mod 2^16, the sample enters as the
//     unsigned byte u = g + 128, and (u * 16 noise_var + TT) mod 2^16, shifted right arithmetically by 4, is
//     wrap12(shaped + g noise_var) sign-extended to int16: V_PK_MAD_U16 + V_PK_ASHRREV_I16 per pair.
//   * the data bits come packed from the PRBS generator (or the pulse pattern); a generator's 10-bit window
//     moves by two bits per round and is re-read from the L1/L2-resident bit buffer.
struct TxFuse {
    int16_t coeffs[64];
    const uint32_t *bits;     // packed data bits, bit 0 of the buffer = data bit m0v (64 zero bits lead when the stream starts)
    uint32_t rel_base;        // window bit offset of the sample at output position 0: (F >> 3) - 7 - m0v, F = first_sample - 17
    uint32_t c0;              // F & 7
    int32_t noise_var;
    int32_t bit_en;           // 0: shaped = 0 (tx.py:65-66)
    int32_t use_bits;         // 0: every data bit reads as 0
    uint32_t last_word;       // index of the last 32-bit word of the buffer that may be read
};

// fills a TxFuse from a call's arguments (the fused one-kernel transmitter and the shaping mover share it)
static TxFuse make_txfuse(const int16_t *coeffs, const uint32_t *d_bits, uint32_t nwords32, uint32_t rel_base, uint32_t c0, int noise_var,
                          int bit_en, int use_bits) {
    TxFuse tx;
    for (int i = 0; i < 64; i++) tx.coeffs[i] = coeffs[i];
    tx.bits = d_bits;
    tx.rel_base = rel_base;
    tx.c0 = c0;
    tx.noise_var = noise_var;
    tx.bit_en = bit_en;
    tx.use_bits = use_bits && nwords32 >= 2;
    tx.last_word = nwords32 ? nwords32 - 1 : 1;
    return tx;
}

// (mask & a) | (~mask & b) with a wave-uniform mask: ONE V_BFI_B32 (left to itself hipcc turns the uniform mask into
// s_not + v_and + v_and_or)
__device__ __forceinline__ uint32_t bfi_uniform(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));

// (The two-kernel "staged" form of the stream no longer goes through this kernel: awgn256_planes_kernel below.)
template <bool TX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
awgn256_kernel(const uint32_t *__restrict planes, void *__restrict dst_, unsigned long long nsamples,
               unsigned L, unsigned long long G, unsigned nlanes, TxFuse tx) {
    __shared__ uint32_t Z[16 * 8 * 64];
    __shared__ __attribute__((aligned(16))) uint16_t TT[TX ? 256 * 8 : 8];
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    // highest wave priority: when the seeding of the next fill (bbb_awgn_prefetch) shares the SIMD it gets the
    // issue slots this wave leaves free instead of every other one
    __builtin_amdgcn_s_setprio(3);

    uint32_t selmask[4] = {0, 0, 0, 0};
    if (TX) {
        // TT[q][j] = 16 * (T[(c0+j) & 7][q] - 128 nv) mod 2^16;  T[ph][q] = wrap12(sum_idx +-coeffs[8 idx + ph]), the
        // sign by data bit M - idx = bit (7 - idx) of q (bitshaper.py:52-58,74)
        for (int e = (int)lane; e < 256 * 8; e += 64) {
            const int q = e >> 3, j = e & 7, ph = (int)((tx.c0 + (unsigned)j) & 7u);
            int sum = 0;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {
                const int c = tx.coeffs[8 * idx + ph];
                sum += ((q >> (7 - idx)) & 1) ? c : -c;
            }
            const int shaped = tx.bit_en ? ((int)((unsigned)sum << 20) >> 20) : 0;
            TT[e] = (uint16_t)((unsigned)((shaped - 128 * tx.noise_var) * 16) & 0xffffu);
        }
        // pair d of a group of eight holds samples 2d, 2d+1: window shift 0 while c0 + e < 8
#pragma unroll
        for (int d = 0; d < 4; d++)
            selmask[d] = (tx.c0 + 2u * (unsigned)d < 8u ? 0x0000ffffu : 0u) | (tx.c0 + 2u * (unsigned)d + 1u < 8u ? 0xffff0000u : 0u);
        __syncthreads();
    }

    // lutopt256_step yields the sample of the state it is GIVEN (and its successor): the planes hold
    // the state before the first sample, so advance once
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
#pragma unroll
    for (int p = 0; p < 256; p++) b[p] = planes[(size_t)p * nlanes + LG];
    lutopt256_advance(b, a);
#define BBB_PARK(p) BBB_ACC_WRITE(pa[p], a[p]);
    LUTOPT256_FOR_PARKED(BBB_PARK)
#undef BBB_PARK

    const unsigned rounds = L / 16;
    const bool wave_full = (wave * 32 + 31) * 64 + 63 < G && ((wave * 32 + 32) * 64) * (unsigned long long)L <= nsamples;
#pragma unroll 1
    for (unsigned r = 0; r < rounds; r++) {
#pragma unroll 1
        for (unsigned tt = 0; tt < 8; tt++) {
            lutopt256_step_parked(a, pa, b, pb, cnt);
            planes8_to_bytes(cnt);
#pragma unroll
            for (int i = 0; i < 8; i++) Z[((2 * tt) * 8 + i) * 64 + lane] = cnt[i];
            lutopt256_step_parked(b, pb, a, pa, cnt);
            planes8_to_bytes(cnt);
#pragma unroll
            for (int i = 0; i < 8; i++) Z[((2 * tt + 1) * 8 + i) * 64 + lane] = cnt[i];
        }
        // TX: the data-bit windows of this lane's generators, ONE load each (32 bits from the byte that holds the window's
        // first bit), 16 of them issued together ahead of the four iterations that use them -- their latency, which the
        // piece mover's and the stores' traffic stretches to microseconds, is exposed twice per round and not eight times
        auto load_windows = [&](const unsigned ihalf, uint32_t (&win)[16]) {
#pragma unroll
            for (unsigned e = 0; e < 16; e++) {
                const unsigned j = 8 * (e >> 2) + 4 * ihalf + (e & 3);                  // generator j = 8 q + i, q = e >> 2
                const unsigned long long g = gen_index(wave, lane, j);
                win[e] = 0;
                if (g < G && tx.use_bits) {
                    const uint32_t off = (uint32_t)g * L + r * 16u;                     // fits 32 bits (host check)
                    const uint32_t rel = (off >> 3) + tx.rel_base;
                    const uint32_t byte = min(rel >> 3, tx.last_word * 4u - 4u);        // (rounds past the end of the request)
                    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
                    win[e] = *reinterpret_cast<const u32_unaligned *>(reinterpret_cast<const char *>(tx.bits) + byte);
                }
            }
        };
        // FULL: every generator of this wave exists and its whole segment lies inside the request (all waves but the
        // last): no per-generator bounds, no divergent branches
        asm volatile("" ::: "memory");      // (keeps hipcc from pulling the round end's first LDS reads into the step loop)
        // the 16 staged words of iteration i (4 steps x 4 groups); read one iteration AHEAD of their use: beside the seeding
        // kernel, which saturates the CU's LDS, a read takes microseconds
        auto load_z = [&](const unsigned i, uint32_t (&zz)[16]) {
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int t = 0; t < 4; t++) zz[4 * w + t] = Z[((4 * w + t) * 8 + i) * 64 + lane];
        };
        auto round_end = [&](auto full_c, const unsigned i, const uint32_t (&w4)[4], const uint32_t (&zz)[16]) {
            constexpr bool FULL = decltype(full_c)::value;
            uint32_t o[4][4];                 // o[w][q] after the transposes
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t z[4] = {zz[4 * w], zz[4 * w + 1], zz[4 * w + 2], zz[4 * w + 3]};
                transpose4x4_bytes(z);        // z[q] = bytes (t = 4w..4w+3) of generator 8q+i
#pragma unroll
                for (int q = 0; q < 4; q++) o[w][q] = z[q];
            }
            // TX: the table rows of all four generators, twelve LDS reads issued together (10-bit window: bit j = data
            // bit M0 - 7 + j; one 16-byte row per window shift)
            u32x4 RA[TX ? 4 : 1], RB[TX ? 4 : 1], RC[TX ? 4 : 1];
            if (TX) {
#pragma unroll
                for (unsigned q = 0; q < 4; q++) {
                    const uint32_t gq = (uint32_t)gen_index(wave, lane, 8 * q + i);
                    const uint32_t rel_q = ((gq * L + r * 16u) >> 3) + tx.rel_base;
                    const uint32_t Q4 = ((w4[q] >> (rel_q & 7u)) & 0x3ffu) << 4;
                    RA[q] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(TT) + (Q4 & 0xff0u));
                    RB[q] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(TT) + ((Q4 >> 1) & 0xff0u));
                    RC[q] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(TT) + ((Q4 >> 2) & 0xff0u));
                }
            }
#pragma unroll
            for (unsigned q = 0; q < 4; q++) {
                const unsigned long long g = gen_index(wave, lane, 8 * q + i);
                const unsigned long long off = g * L + (unsigned long long)r * 16;
                if (!FULL && !(g < G && off < nsamples)) continue;   // (the q loop)
                if (!TX) {
                    int8_t *dst = (int8_t *)dst_;
                    const u32x4 v = {o[0][q], o[1][q], o[2][q], o[3][q]};
                    if (FULL || off + 16 <= nsamples) {
                        *reinterpret_cast<u32x4 *>(dst + off) = v;
                    } else {
                        const unsigned n = (unsigned)(nsamples - off);
                        for (unsigned e = 0; e < n; e++) dst[off + e] = (int8_t)((o[e >> 2][q] >> (8 * (e & 3))) & 0xff);
                    }
                } else {
                    int16_t *dst = (int16_t *)dst_;
                    const u32x4 A = RA[q], B = RB[q], C = RC[q];
                    const u16x2 nv16 = {(uint16_t)(tx.noise_var * 16), (uint16_t)(tx.noise_var * 16)};
                    uint32_t x[8];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const uint32_t u = o[w][q] ^ 0x80808080u;                       // bytes g + 128 of samples 4w .. 4w+3
                        const uint32_t u01 = __builtin_amdgcn_perm(0u, u, 0x0c010c00u);  // [u0, 0, u1, 0]
                        const uint32_t u23 = __builtin_amdgcn_perm(0u, u, 0x0c030c02u);
                        // shaped pairs of samples 4w, 4w+1 and 4w+2, 4w+3 (pair d = (2w) & 3, (2w+1) & 3 of their group of eight)
                        const int d0 = (2 * w) & 3, d1 = (2 * w + 1) & 3;
                        const uint32_t s01 = w < 2 ? bfi_uniform(selmask[d0], A[d0], B[d0]) : bfi_uniform(selmask[d0], B[d0], C[d0]);
                        const uint32_t s23 = w < 2 ? bfi_uniform(selmask[d1], A[d1], B[d1]) : bfi_uniform(selmask[d1], B[d1], C[d1]);
                        const u16x2 m01 = __builtin_bit_cast(u16x2, u01) * nv16 + __builtin_bit_cast(u16x2, s01);
                        const u16x2 m23 = __builtin_bit_cast(u16x2, u23) * nv16 + __builtin_bit_cast(u16x2, s23);
                        x[2 * w] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, m01) >> 4);
                        x[2 * w + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, m23) >> 4);
                    }
                    if (FULL || off + 16 <= nsamples) {
                        const u32x4 lo = {x[0], x[1], x[2], x[3]}, hi = {x[4], x[5], x[6], x[7]};
                        reinterpret_cast<u32x4 *>(dst + off)[0] = lo;
                        reinterpret_cast<u32x4 *>(dst + off)[1] = hi;
                    } else {
                        const unsigned n = (unsigned)(nsamples - off);
                        for (unsigned e = 0; e < n; e++) dst[off + e] = (int16_t)((x[e >> 1] >> (16 * (e & 1))) & 0xffff);
                    }
                }
            }
        };
        auto all_iterations = [&](auto full_c) {
            if (TX) {
#pragma unroll 1
                for (unsigned ihalf = 0; ihalf < 2; ihalf++) {
                    uint32_t win[16];
                    load_windows(ihalf, win);
                    uint32_t zc[16], zn[16];
                    load_z(4 * ihalf, zn);
#pragma unroll
                    for (unsigned ii = 0; ii < 4; ii++) {
#pragma unroll
                        for (int e = 0; e < 16; e++) zc[e] = zn[e];
                        if (ii < 3) load_z(4 * ihalf + ii + 1, zn);
                        const uint32_t w4[4] = {win[ii], win[4 + ii], win[8 + ii], win[12 + ii]};
                        round_end(full_c, 4 * ihalf + ii, w4, zc);
                    }
                }
            } else {
                const uint32_t none[4] = {0, 0, 0, 0};
                uint32_t zc[16], zn[16];
                load_z(0, zn);
#pragma unroll 1
                for (unsigned i = 0; i < 8; i++) {
#pragma unroll
                    for (int e = 0; e < 16; e++) zc[e] = zn[e];
                    load_z((i + 1) & 7, zn);          // (the last one re-reads iteration 0: harmless, keeps the loop uniform)
                    round_end(full_c, i, none, zc);
                }
            }
        };
        if (wave_full) all_iterations(std::true_type{});
        else all_iterations(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// The PLANES form of the staged stream (round 3).  awgn256_kernel<*, true> still spends ~12 % of its issue slots on
// turning count planes into bytes: 47 shift / BFI butterflies per step, 8 LDS stores per step, and per round 128 LDS
// reads, 256 V_PERM, 32 stores and the waits on them -- all of it on the one wave per SIMD that issues a vector
// instruction every 4 cycles.  None of that needs the state: it is a transposition of data that is already final.
// So this kernel stores the 8 count planes of every step as they are -- two 16-byte stores per lane and step, 1 KiB
// per store instruction, no LDS, no round end -- and the transposition moves into the mover (unplane_kernel), a guest
// with two waves per CU that runs in the issue slots the sample kernel cannot use.
// Staging layout: u32x4 stage[wave][step][half][lane], half 0 = planes 0..3, half 1 = planes 4..7 (plane 7 already
// complemented: the int8 two's complement form).  Every wave runs all L steps (generators beyond G are padding).
// (Tried: [wave][lane >> 3][step][half][lane & 7], which makes a mover unit's 32 KiB ONE contiguous block instead of 256 lines
// out of 256 different 1 KiB rows.  Slower: noise 1.085 -> 1.19 ms per 1e9 at one read per kernel, the shaping mover's
// calls 1.44 -> 1.77 ms -- a CU's burst then falls on few memory channels; scattered lines spread over all of them.)
// ---------------------------------------------------------------------------------------------
// Two forms (SMALL).  false: the loaded state is advanced once in front of the loop (`planes` = the state BEFORE the first
// sample, as for every other kernel).  Those few instructions hold the loaded and the advanced state at once -- 512 values, of
// which hipcc parks ~180 in AGPRs -- and make the kernel a 434-register one: what is left of a SIMD's 512 holds ONE guest
// wave.  true: `planes` = the state OF the first sample (the host seeds one clock further), no advance: 344 registers
// (hipcc packs the loop's parked planes into 95 AGPRs), room for the mover's AND the seeding's wave at once.
// Same loop, yet the small form is 1-2 % slower alone and 6 % slower beside the noise stream's guests, whatever its
// register count is made to be (same box: profiles/r03_small_footprint_ab.log) -- so the noise stream, which is bound by
// this kernel, keeps the first form, and the transmitter, which is bound by the kernel's guests (a shaping mover per call,
// then the next kernel's seeding: one after the other beside the first form), takes the second: its guests run at the same
// time (TX 1e9 samples per call: 1.53 -> 1.31 ms at one call per kernel, 1.34 -> 1.25 at two).
template <bool SMALL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
awgn256_planes_kernel(const uint32_t *__restrict planes, u32x4 *__restrict stage, unsigned L, unsigned nlanes
) {
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    __builtin_amdgcn_s_setprio(3);
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
    if constexpr (SMALL) {
#ifdef BBB_SMALL_THROTTLE
        asm volatile("" ::: BBB_SMALL_THROTTLE);      // (experiments: the small form with a chosen register footprint)
#endif
        // (the parked planes straight into their AGPRs; scalar base per plane + ONE 32-bit lane offset)
        const uint32_t voff = (uint32_t)LG * 4u;
#define BBB_PLANE(p) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(planes + (size_t)(p) * nlanes) + (unsigned long long)voff))
#define BBB_PARK(p) { const uint32_t v_ = BBB_PLANE(p); BBB_ACC_WRITE(pa[p], v_); }
        LUTOPT256_FOR_PARKED_HI(BBB_PARK)
#undef BBB_PARK
#pragma unroll
        for (int p = 0; p < 256; p++)
            if (!lutopt256_hi_is_parked(p)) a[p] = BBB_PLANE(p);
#undef BBB_PLANE
    } else {
#pragma unroll
        for (int p = 0; p < 256; p++) b[p] = planes[(size_t)p * nlanes + LG];
        lutopt256_advance(b, a);
#define BBB_PARK(p) BBB_ACC_WRITE(pa[p], a[p]);
        LUTOPT256_FOR_PARKED_HI(BBB_PARK)
#undef BBB_PARK
    }
    u32x4 *out = stage + (wave * L) * 128 + lane;
#pragma unroll 1
    for (unsigned t = 0; t < L; t += 2) {
        lutopt256_step_parked_hi(a, pa, b, pb, cnt);
        {
        __builtin_nontemporal_store((u32x4){cnt[0], cnt[1], cnt[2], cnt[3]}, out);
        __builtin_nontemporal_store((u32x4){cnt[4], cnt[5], cnt[6], cnt[7]}, out + 64);
        }
        lutopt256_step_parked_hi(b, pb, a, pa, cnt);
        {
        __builtin_nontemporal_store((u32x4){cnt[0], cnt[1], cnt[2], cnt[3]}, out + 128);
        __builtin_nontemporal_store((u32x4){cnt[4], cnt[5], cnt[6], cnt[7]}, out + 192);
        }
        out += 256;
    }
}

int awgn256_planes_launch(const uint32_t *d_planes, void *stage, unsigned L, unsigned nlanes, hipStream_t st, bool small_footprint) {
    if (L & 1) return fail(BBB_EINVAL, "segment length must be even");
    if (small_footprint)
        hipLaunchKernelGGL(awgn256_planes_kernel<true>, dim3(nlanes / 64), dim3(64), 0, st, d_planes, (u32x4 *)stage, L, nlanes);
    else
        hipLaunchKernelGGL(awgn256_planes_kernel<false>, dim3(nlanes / 64), dim3(64), 0, st, d_planes, (u32x4 *)stage, L, nlanes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// The mover of the PLANES form: count planes -> the sequential byte stream.
// It runs BESIDE the next fill's sample kernel, whose wave leaves a SIMD 72 registers and every issue slot but one in
// four: a guest there is bound by latency, not by work -- so the loads are LDS-DMA (global_load_lds_dwordx4: no
// destination registers, a whole unit in flight per workgroup while the previous one is processed).
// Unit of work = (source wave w, 8 of its lanes, 128 steps): 32 KiB of planes in, 256 generators x 128 bytes out.
//   DMA      32 wave-instructions of 1 KiB (8 per wave): a lane fetches 16 bytes of one (step, half) row (8 lanes = one full
//            128-byte line); the LDS image is laid out for the readers, raw[c = 2 s + half][quad-step qs][lane8][16 B]
//            (the DMA's destination is lane-linear, its per-lane SOURCE address is free);
//   phase 1  thread (qs, lane8): 8 x ds_read_b128 (a wave reads 1 KiB contiguous), planes8_to_bytes per step, eight 4x4
//            byte transposes -> for each of the lane's 32 generators the 4 sample bytes of steps 4qs..4qs+3, one
//            ds_write_b32 each into the generator's 128-byte row of the tile (row = j * 8 + lane8; 16-byte chunk c of a
//            row sits at chunk position c ^ 2 * ((lane8 >> 1) & 3): a wave's store hits all 64 banks once);
//   phase 2  thread (row, chunk): ds_read_b128 + one 16-byte store; the 8 lanes of a generator write 128 consecutive
//            bytes of its segment, a wave 8 consecutive segments.
// LDS (dynamic, ONE array: 2 x 32 KiB raw + 32 KiB tile; the shaping mover adds its 32 KiB table and 4 KiB of window words):
// the DMA of unit u+1 is in flight while unit u is processed; raw barriers and counted vmcnt (a __syncthreads() would drain the
// DMA: cdna_hip_programming.md, "Pipelining across barriers").  One block of four waves per CU, persistent over its share of
// the units.
// What bounds it (experiments/mover_phases.py, shader-clock stamps per phase and wave): of ~9500 cycles per unit 1900 go to
// phase 1 (its 340 instructions), 5100 to phase 2 -- eight LDS reads and eight stores -- and 1700 to ISSUING the eight DMA
// instructions: the wave waits for the memory pipeline to accept its loads and stores.  64 KiB per unit and CU in 9500 cycles
// is 3.9 TB/s over the chip: the memory system's rate for this pattern (128-byte pieces, read and written 1:1), not the
// mover's instructions -- taking the divisions out of the loop, the DMA addresses down to one add each and the shaping
// mover's per-sample selects out (1290 -> 830 instructions per unit; 810 -> 440 for the plain mover) changed its time by
// nothing, and so did a third raw buffer (two units in flight).  What did help is the ORDER of the units: see Pos below.
constexpr unsigned kUnplaneRaw = 32 * 1024, kUnplaneLds = 3 * 32 * 1024, kUnplaneLdsTx = kUnplaneLds + 32 * 1024 + 2 * 2048;

struct UnplaneGeom {                 // host computed (unplane_launch_with)
    unsigned w_lo, ngroups, nunits;  // first source wave of the window; ceil(L / 128); units = waves x 8 x ngroups
    unsigned per_block;              // a block takes the units [per_block * blockIdx.x, + per_block) of the order (w, q8, rg)
};

// NOWRAP (shaping mover only, round 5): the host has shown that no sample can leave the 12-bit range -- max over the phases of
// sum |coeffs| + 128 noise_var <= 2047 (unplane_tx_nowrap: true of every coefficient set the reference ships up to noise_var 11) -- so
// wrap12 is the identity, the table holds shaped - 128 noise_var itself instead of 16 times it and the V_PK_ASHRREV_I16 behind every
// V_PK_MAD_U16 goes: 64 of the mover's 634 vector instructions per unit, and 4.2 % of the transmitter stream's time, same box
// (profiles/r05_stream_variants.log: the stream is bound by the instructions of everything that runs beside the noise kernel).
template <bool TXM, bool NOWRAP = false>
__global__ void __launch_bounds__(256, 7)      // <= 72 registers: a wave of this kernel must fit beside the sample kernel's (<= 440 of 512)
unplane_kernel(const u32x4 *__restrict stage, char *__restrict dst, unsigned long long win_lo, unsigned long long nbytes_, unsigned L,
               unsigned long long G, UnplaneGeom ge, TxFuse tx) {
    // DYNAMIC shared memory: with a static array of this size hipcc derives "at most one wave per SIMD" from the LDS size
    // and enforces it by declaring 257 registers for this kernel -- which then cannot share a SIMD with the sample
    // kernel's wave (measured: half of the sample waves waited for the mover to leave)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (blockIdx.x * ge.per_block >= ge.nunits) return;
    // the stream window [win_lo, win_lo + nbytes_) of the staged segments goes to dst[0 ..): dst is indexed by stream offset
    char *const dstw = dst - win_lo;
    const unsigned long long nbytes = win_lo + nbytes_;
    // phase 1: quad-step and lane of eight
    const unsigned l8 = tid & 7, qs = tid >> 3;
    const unsigned wr0 = l8 * 32 + ((((qs >> 2) ^ (2 * ((l8 >> 1) & 3))) << 2) | (qs & 3));      // word of (row j*8 + l8, steps 4qs..)
    // phase 2: rows (4 k + wv) * 8 + lq, k = 0..7; chunk c2
    const unsigned lq = (tid >> 3) & 7, c2 = tid & 7;
    const unsigned rd0 = (wv * 8 + lq) * 32 + ((c2 ^ (2 * ((lq >> 1) & 3))) << 2);
    uint32_t *const tile = lds + 2 * (kUnplaneRaw / 4);
    // TXM (the SHAPING mover of bbb_tx_fill_i16 on a staged handle): every 16-byte piece of noise leaves as the 16 int16
    // samples x = wrap12(bit_en shaped + g noise_var) (tx.py:75-81), the arithmetic of the fused kernel's round end (TxFuse and
    // the multiply-add scheme described above awgn256_kernel) done here, by the guest.  The fused kernel's table TT[q][j] (8
    // phases of one 8-bit window) is widened to T10[q10][e]: the 16 shaped samples of a piece for each value of its 10-bit
    // data window -- two ds_read_b128 and no per-sample select.
    uint32_t *const T10 = lds + kUnplaneLds / 4;              // [1024 windows][8 words = 16 samples]
    uint32_t *const winb = lds + kUnplaneLds / 4 + 8192;      // [2][256 generators][2 words]: the units' data-bit windows, by DMA
    if (TXM) {
        uint16_t *const TT = reinterpret_cast<uint16_t *>(tile);      // (the tile is idle until the first unit's phase 1)
        for (int e = (int)tid; e < 256 * 8; e += 256) {
            const int q = e >> 3, j = e & 7, ph = (int)((tx.c0 + (unsigned)j) & 7u);
            int sum = 0;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {
                const int c = tx.coeffs[8 * idx + ph];
                sum += ((q >> (7 - idx)) & 1) ? c : -c;
            }
            const int shaped = tx.bit_en ? ((int)((unsigned)sum << 20) >> 20) : 0;
            TT[e] = (uint16_t)((unsigned)((shaped - 128 * tx.noise_var) * (NOWRAP ? 1 : 16)) & 0xffffu);
        }
        __syncthreads();
        // sample e of a piece has phase (c0 + e) & 7 and sees the window shifted by (c0 + e) >> 3 in {0, 1, 2} data bits
        for (unsigned p = tid; p < 1024 * 8; p += 256) {
            const unsigned q10 = p >> 3, e = 2 * (p & 7);
            const uint32_t lo = TT[(((q10 >> ((tx.c0 + e) >> 3)) & 0xffu) << 3) + (e & 7)];
            const uint32_t hi = TT[(((q10 >> ((tx.c0 + e + 1) >> 3)) & 0xffu) << 3) + ((e + 1) & 7)];
            T10[p] = lo | (hi << 16);
        }
        __syncthreads();      // (TT is overwritten by the first unit's phase 1; nothing is in flight yet, so the full barrier costs nothing)
    }
    // a unit's place: 8 lanes q8 of source wave w (relative to ge.w_lo), steps 128 rg ..; unit index = (w * 8 + q8) * ngroups + rg.
    // A block takes CONSECUTIVE units: the step groups of the same 8 lanes one after the other.  A generator's 128 bytes of a
    // unit are not aligned to the 128-byte lines of the stream unless L is a multiple of 128 (1e9 samples: L = 480); the rest
    // of its first and last line belongs to the neighbouring step groups, and those now pass through the same L2 within
    // microseconds instead of through eight different ones (with a block stride over the units the eight lanes-of-eight of
    // one (wave, step group) ran at the same time on eight XCDs, and every line left its L2 partly written).
    struct Pos { unsigned q8, rg, w; };
    auto advance = [&](Pos &p) {
        if (++p.rg == ge.ngroups) {
            p.rg = 0;
            if (++p.q8 == 8) { p.q8 = 0; p.w++; }
        }
    };
    const unsigned voff_lane = (lane & 7) * 16 + (lane >> 3) * 8192;      // a lane's 16 bytes of its row; rows 4 steps (8 KiB) apart
    // the 8 DMA instructions of this wave for a unit into raw buffer `buf`: 1 KiB block b = wv * 8 + k holds
    // c = b >> 2 = 2 wv + (k >> 2) (step-in-quad s = wv, half = k >> 2) of quad-steps (k & 3) * 8 .. + 8
    auto dma_unit = [&](const Pos &p, unsigned buf) {
        const unsigned step0 = p.rg * 128;
        const unsigned long long wabs = (unsigned long long)ge.w_lo + p.w;
        const char *const sb = reinterpret_cast<const char *>(stage) + ((wabs * L + step0) * 128 + p.q8 * 8) * 16;
        uint32_t *const rawb = lds + buf * (kUnplaneRaw / 4) + wv * 8 * 256;
        if (step0 + 128 <= L) {
            const char *const pl = sb + wv * 2048 + voff_lane;
#pragma unroll
            for (unsigned k = 0; k < 8; k++)
                __builtin_amdgcn_global_load_lds((const void *)(pl + ((k & 3) * 32 * 2048 + (k >> 2) * 1024)),
                                                 (lds_void_ptr)(uintptr_t)(rawb + k * 256), 16, 0, 0);
        } else {
            // a segment's last unit may be short: phase 2 never writes the missing steps, the DMA re-reads the last one
            const unsigned last = L - 1 - step0;
#pragma unroll
            for (unsigned k = 0; k < 8; k++) {
                unsigned st = 4 * ((k & 3) * 8 + (lane >> 3)) + wv;
                st = st < last ? st : last;
                __builtin_amdgcn_global_load_lds((const void *)(sb + (size_t)st * 2048 + (k >> 2) * 1024 + (lane & 7) * 16),
                                                 (lds_void_ptr)(uintptr_t)(rawb + k * 256), 16, 0, 0);
            }
        }
        if (TXM) {
            // the data bits the unit's pieces will need: per generator the two 32-bit words that hold the windows of its 128
            // samples (16 data bits + 10 of window), fetched like the planes: no registers, landed before phase 2 asks.
            // pu = position in this call of the generator's first sample of the unit (negative when the window starts inside
            // the unit; the buffer leads with 128 zero bits, so the word index stays >= 0)
            const long long so = (long long)((wabs * 2048 + p.q8 * 8) * (unsigned long long)L + step0) - (long long)win_lo;
#pragma unroll
            for (unsigned i = 0; i < 2; i++) {
                const unsigned e = (wv * 2 + i) * 64 + lane, row = e >> 1;
                const long long pu = so + (long long)((unsigned long long)((row >> 3) * 64 + (row & 7)) * L);
                uint32_t word = 0;
                if (pu > -128 && pu < (long long)nbytes_) word = (((uint32_t)(pu >> 3) + tx.rel_base) >> 5) + (e & 1);
                word = word < tx.last_word ? word : tx.last_word;
                __builtin_amdgcn_global_load_lds((const void *)(tx.bits + word), (lds_void_ptr)(uintptr_t)(winb + buf * 512 + (wv * 2 + i) * 64), 4, 0, 0);
            }
        }
    };
    // counted wait: all but the wave's n youngest vector-memory operations done (n wave-uniform)
    auto wait_vm = [](unsigned n) {
        switch (n) {
#define BBB_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
            BBB_W(8) BBB_W(10) BBB_W(16) BBB_W(26)
#undef BBB_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    constexpr unsigned NDMA = TXM ? 10 : 8, NST = TXM ? 16 : 8;      // DMA instructions per unit and wave; stores of a straight-line unit
    const unsigned u0 = blockIdx.x * ge.per_block;
    const unsigned n_it = ge.nunits - u0 < ge.per_block ? ge.nunits - u0 : ge.per_block;
    Pos cur;
    cur.rg = u0 % ge.ngroups;
    cur.q8 = (u0 / ge.ngroups) & 7;
    cur.w = (u0 / ge.ngroups) >> 3;
    Pos nxt = cur;
    advance(nxt);
    unsigned buf = 0;
    bool fast1 = false;      // wave-uniform: the whole wave took the straight-line store path in the previous unit
    dma_unit(cur, 0);
    for (unsigned it = 0; it < n_it; it++, buf ^= 1) {
        const unsigned q8 = cur.q8, step0 = cur.rg * 128;
        const unsigned long long w = (unsigned long long)ge.w_lo + cur.w;
        const bool more1 = it + 1 < n_it;
        if (more1) dma_unit(nxt, buf ^ 1);
        // This unit's DMA must have landed.  vmcnt counts loads, DMA and stores together, in issue order.  Younger than this
        // unit's DMA are the previous unit's stores and the DMA just issued: with a KNOWN count (a straight-line unit issues
        // exactly NST store instructions per wave) they stay in flight; otherwise the wait covers the stores too.
        wait_vm((more1 ? NDMA : 0u) + (fast1 ? NST : 0u));
        __builtin_amdgcn_s_barrier();
        // ---- phase 1
        {
            const uint32_t *raw = lds + buf * (kUnplaneRaw / 4) + (qs * 8 + l8) * 4;
            uint32_t Z[4][8];
#pragma unroll
            for (unsigned s = 0; s < 4; s++) {
                const u32x4 lo = *reinterpret_cast<const u32x4 *>(raw + (2 * s) * 1024);
                const u32x4 hi = *reinterpret_cast<const u32x4 *>(raw + (2 * s + 1) * 1024);
                Z[s][0] = lo[0]; Z[s][1] = lo[1]; Z[s][2] = lo[2]; Z[s][3] = lo[3];
                Z[s][4] = hi[0]; Z[s][5] = hi[1]; Z[s][6] = hi[2]; Z[s][7] = TXM ? ~hi[3] : hi[3];      // (TXM: u = g + 128, the unsigned count)
            }
#pragma unroll
            for (unsigned s = 0; s < 4; s++) planes8_to_bytes(Z[s]);
#pragma unroll
            for (unsigned i = 0; i < 8; i++) {
                uint32_t z[4] = {Z[0][i], Z[1][i], Z[2][i], Z[3][i]};
                transpose4x4_bytes(z);                                  // z[q] = bytes (steps 4qs..4qs+3) of generator j = 8q + i
#pragma unroll
                for (unsigned q = 0; q < 4; q++) tile[wr0 + (8 * q + i) * 256] = z[q];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: row (4 k + wv) * 8 + lq = generator (w * 32 + 4 k + wv) * 64 + q8 * 8 + lq, its bytes step0 + 16 c2 ..
        const unsigned inseg = step0 + c2 * 16;
        const unsigned long long g0 = (w * 32 + wv) * 64 + q8 * 8 + lq;
        unsigned long long off = g0 * (unsigned long long)L + inseg;
        const unsigned long long goff = 256ull * L;
        // the straight-line path is taken by the WHOLE wave or not at all (the count of store instructions a wave issues then
        // is exact: see the wait above)
        const bool lane_fast = inseg < L && g0 + 7 * 256 < G && off >= win_lo && off + 7 * goff + 16 <= nbytes;
        const bool wave_fast = __all(lane_fast);
        fast1 = wave_fast;
        if (!TXM && inseg < L) {
            if (wave_fast) {
                // every generator of this thread exists and its chunk lies inside the window
#pragma unroll
                for (unsigned k = 0; k < 8; k++) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[rd0 + k * 1024]);
                    *reinterpret_cast<u32x4 *>(dstw + off) = v;
                    off += goff;
                }
            } else {
                for (unsigned k = 0; k < 8; k++, off += goff) {
                    if (g0 + 256ull * k >= G || off >= nbytes) break;
                    if (off < win_lo) continue;                    // (the window starts on a 16-byte boundary of the stream)
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[rd0 + k * 1024]);
                    if (off + 16 <= nbytes) {
                        *reinterpret_cast<u32x4 *>(dstw + off) = v;
                    } else {
                        const unsigned n = (unsigned)(nbytes - off);
                        for (unsigned e = 0; e < n; e++) dstw[off + e] = (char)((v[e >> 2] >> (8 * (e & 3))) & 0xff);
                    }
                }
            }
        }
        if (TXM && inseg < L) {
            // positions are relative to the window: sample p of this call = stream offset win_lo + p; dst holds int16
            int16_t *const dst16 = reinterpret_cast<int16_t *>(dst);
            const u16x2 nv16 = {(uint16_t)(tx.noise_var * (NOWRAP ? 1 : 16)), (uint16_t)(tx.noise_var * (NOWRAP ? 1 : 16))};
            const uint32_t idx_mask = tx.use_bits ? 0x7fe0u : 0u;            // (no data bits: every window reads as 0)
            // where the thread's pieces sit in their generators' two window words (dma_unit): the same for all eight, their
            // positions differ by multiples of 256 L samples = 32 L data bits
            const uint32_t sh = ((((uint32_t)((long long)(off - win_lo) >> 3)) + tx.rel_base - 2u * c2) & 31u) + 2u * c2;
            const uint32_t *const wb0 = winb + buf * 512 + (wv * 8 + lq) * 2;
            // one 16-byte piece of noise -> 16 int16 samples; FULL: the piece lies wholly inside the window (no tail)
            auto piece = [&](auto full_c, unsigned k, unsigned long long o) {
                constexpr bool FULL = decltype(full_c)::value;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 ww = *reinterpret_cast<const u32x2 *>(wb0 + k * 64);
                const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[rd0 + k * 1024]);
                // the piece's 10-bit data window (bit j = data bit M0 - 7 + j) selects its row of shaped samples
                const uint32_t wk = (uint32_t)((((unsigned long long)ww[1] << 32) | ww[0]) >> sh);
                const char *const row = reinterpret_cast<const char *>(T10) + ((wk << 5) & idx_mask);
                const u32x4 A = *reinterpret_cast<const u32x4 *>(row);
                const u32x4 B = *reinterpret_cast<const u32x4 *>(row + 16);
                uint32_t x[8];
#pragma unroll
                for (int w4 = 0; w4 < 4; w4++) {
                    const uint32_t u01 = __builtin_amdgcn_perm(0u, v[w4], 0x0c010c00u);  // [u0, 0, u1, 0]: bytes g + 128 of samples 4w, 4w+1
                    const uint32_t u23 = __builtin_amdgcn_perm(0u, v[w4], 0x0c030c02u);
                    const uint32_t s01 = w4 < 2 ? A[2 * w4] : B[2 * w4 - 4], s23 = w4 < 2 ? A[2 * w4 + 1] : B[2 * w4 - 3];
                    const u16x2 m01 = __builtin_bit_cast(u16x2, u01) * nv16 + __builtin_bit_cast(u16x2, s01);
                    const u16x2 m23 = __builtin_bit_cast(u16x2, u23) * nv16 + __builtin_bit_cast(u16x2, s23);
                    if constexpr (NOWRAP) {
                        x[2 * w4] = __builtin_bit_cast(uint32_t, m01);
                        x[2 * w4 + 1] = __builtin_bit_cast(uint32_t, m23);
                    } else {
                        x[2 * w4] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, m01) >> 4);
                        x[2 * w4 + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, m23) >> 4);
                    }
                }
                int16_t *out = dst16 + (o - win_lo);
                if (FULL || o + 16 <= nbytes) {
                    const u32x4 lo = {x[0], x[1], x[2], x[3]}, hi = {x[4], x[5], x[6], x[7]};
                    reinterpret_cast<u32x4 *>(out)[0] = lo;
                    reinterpret_cast<u32x4 *>(out)[1] = hi;
                } else {
                    const unsigned n = (unsigned)(nbytes - o);
                    for (unsigned e = 0; e < n; e++) out[e] = (int16_t)((x[e >> 1] >> (16 * (e & 1))) & 0xffff);
                }
            };
            if (wave_fast) {
                // every piece of this wave exists and lies inside the window: straight-line code, so that the LDS reads of
                // one piece are in flight while the previous one is shaped (a guest wave waits ~100 cycles per dependent read)
                // (four at a time: eight in flight need more than the 72 registers a guest wave gets, and a spill to scratch
                // would count in vmcnt, which this kernel counts by hand)
#pragma unroll 1
                for (unsigned k4 = 0; k4 < 8; k4 += 4) {
#pragma unroll
                    for (unsigned k = 0; k < 4; k++) piece(std::true_type{}, k4 + k, off + (k4 + k) * goff);
                }
            } else {
#pragma unroll 1
                for (unsigned k = 0; k < 8; k++) {
                    const unsigned long long o = off + k * goff;
                    if (g0 + 256ull * k >= G || o >= nbytes) break;
                    if (o < win_lo) continue;
                    piece(std::false_type{}, k, o);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // the tile and raw[buf] are free for the next unit
        cur = nxt;
        advance(nxt);
    }
}

// wrap12 cannot act on shaped + g noise_var, g in [-128, 127]: |shaped| <= max over the phases of sum_idx |coeffs[8 idx + ph]| (every sign
// pattern is one of the 256 windows), tx.py:75-81
static bool unplane_tx_nowrap(const TxFuse &tx) {
    int worst = 0;
    if (tx.bit_en)
        for (int ph = 0; ph < 8; ph++) {
            int sum = 0;
            for (int idx = 0; idx < 8; idx++) sum += tx.coeffs[8 * idx + ph] < 0 ? -(int)tx.coeffs[8 * idx + ph] : (int)tx.coeffs[8 * idx + ph];
            worst = sum > worst ? sum : worst;
        }
    return tx.noise_var >= 0 && worst + 128 * tx.noise_var <= 2047;
}

static int unplane_launch_with(const void *stage, void *dst, uint64_t win_lo, uint64_t nbytes, unsigned L, uint64_t G, unsigned nlanes,
                               const TxFuse *tx, hipStream_t st) {
    if (win_lo & 15) return fail(BBB_EINVAL, "window must start on a 16-byte boundary of the staged stream");
    if (nbytes == 0) return BBB_OK;
    int dev = 0, ncu = 256;
    BBB_HIP(hipGetDevice(&dev));
    BBB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    // only the source waves whose generators touch the window: generator g owns [g L, (g + 1) L), wave w generators [2048 w, 2048 (w + 1))
    const uint64_t seg = (uint64_t)L * 2048;
    const unsigned w_lo = (unsigned)(win_lo / seg);
    uint64_t w_hi = (win_lo + nbytes + seg - 1) / seg;
    if (w_hi > nlanes / 64) w_hi = nlanes / 64;
    const unsigned w_n = (unsigned)(w_hi - w_lo);
    const unsigned ngroups = (L + 127) / 128;
    const uint64_t nunits = (uint64_t)w_n * ngroups * 8;
    if (nunits >> 32) return fail(BBB_EINVAL, "staged window too large for one mover launch (2^32 units of 32 KiB)");
    uint64_t blocks = (uint64_t)ncu * (uint64_t)env_knob("BBB_UNPLANE_BLOCKS_PER_CU", 1);
    if (blocks > nunits) blocks = nunits;
    UnplaneGeom ge;
    ge.w_lo = w_lo; ge.ngroups = ngroups; ge.nunits = (unsigned)nunits;
    ge.per_block = (unsigned)((nunits + blocks - 1) / blocks);
    blocks = (nunits + ge.per_block - 1) / ge.per_block;
    {
        static std::mutex mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> g(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            BBB_HIP(hipFuncSetAttribute((const void *)unplane_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUnplaneLds));
            BBB_HIP(hipFuncSetAttribute((const void *)unplane_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUnplaneLdsTx));
            BBB_HIP(hipFuncSetAttribute((const void *)unplane_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUnplaneLdsTx));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    if (tx && unplane_tx_nowrap(*tx)) {
        hipLaunchKernelGGL((unplane_kernel<true, true>), dim3((unsigned)blocks), dim3(256), kUnplaneLdsTx, st, (const u32x4 *)stage, (char *)dst,
                           (unsigned long long)win_lo, (unsigned long long)nbytes, L, (unsigned long long)G, ge, *tx);
    } else if (tx) {
        hipLaunchKernelGGL(unplane_kernel<true>, dim3((unsigned)blocks), dim3(256), kUnplaneLdsTx, st, (const u32x4 *)stage, (char *)dst,
                           (unsigned long long)win_lo, (unsigned long long)nbytes, L, (unsigned long long)G, ge, *tx);
    } else {
        TxFuse none{};
        hipLaunchKernelGGL(unplane_kernel<false>, dim3((unsigned)blocks), dim3(256), kUnplaneLds, st, (const u32x4 *)stage, (char *)dst,
                           (unsigned long long)win_lo, (unsigned long long)nbytes, L, (unsigned long long)G, ge, none);
    }
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int unplane_launch(const void *stage, void *dst, uint64_t win_lo, uint64_t nbytes, unsigned L, uint64_t G, unsigned nlanes, hipStream_t st) {
    return unplane_launch_with(stage, dst, win_lo, nbytes, L, G, nlanes, nullptr, st);
}

int unplane_tx_launch(const void *stage, int16_t *dst, uint64_t win_lo, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                      const int16_t *coeffs, const uint32_t *d_bits, uint32_t nwords32, uint32_t rel_base, uint32_t c0, int noise_var, int bit_en,
                      int use_bits, hipStream_t st) {
    const TxFuse tx = make_txfuse(coeffs, d_bits, nwords32, rel_base, c0, noise_var, bit_en, use_bits);
    return unplane_launch_with(stage, dst, win_lo, nsamples, L, G, nlanes, &tx, st);
}

// int8 -> int16 (sign extension), 16 samples per lane: the int16 form of the k = 256 stream is the fast
// int8 fill followed by this pass
__global__ void __launch_bounds__(256)
widen_i8_i16_kernel(const u32x4 *__restrict src, u32x4 *__restrict dst, unsigned long long n16) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const u32x4 v = src[i];
    u32x4 lo, hi;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = v[w];
        const uint32_t a = (uint32_t)(uint16_t)(int16_t)(int8_t)(x & 0xff) | ((uint32_t)(uint16_t)(int16_t)(int8_t)((x >> 8) & 0xff) << 16);
        const uint32_t b = (uint32_t)(uint16_t)(int16_t)(int8_t)((x >> 16) & 0xff) | ((uint32_t)(uint16_t)(int16_t)(int8_t)(x >> 24) << 16);
        if (w < 2) { lo[2 * w] = a; lo[2 * w + 1] = b; } else { hi[2 * (w - 2)] = a; hi[2 * (w - 2) + 1] = b; }
    }
    dst[2 * i] = lo;
    dst[2 * i + 1] = hi;
}

int widen_i8_i16_launch(const int8_t *src, int16_t *dst, uint64_t n, hipStream_t st) {
    const unsigned long long n16 = (n + 15) / 16;
    if (!n16) return BBB_OK;
    hipLaunchKernelGGL(widen_i8_i16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const u32x4 *>(src), reinterpret_cast<u32x4 *>(dst), n16);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// ---------------------------------------------------------------------------------------------
// Table-driven kernel for any power-of-two k <= 512 and any tap lists (slow path: used for the
// reference's small test matrices and for k != 256).  State planes live in global scratch
// [2][k][nlanes]; every lane touches only its own column, so no synchronisation is needed.
// T = sum_i y_i is kept as a 10-plane ripple counter.
// ---------------------------------------------------------------------------------------------
template <typename OutT>
__global__ void __launch_bounds__(64)
awgn_generic_kernel(int k, int logk, const uint16_t *__restrict taps, const uint32_t *__restrict row_off,
                    uint32_t *__restrict planes2, OutT *__restrict dst, unsigned long long nsamples, unsigned L,
                    unsigned long long G, unsigned nlanes) {
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    uint32_t *cur = planes2, *nxt = planes2 + (size_t)k * nlanes;
    for (unsigned t = 0; t < L; t++) {
        uint32_t c[10];
#pragma unroll
        for (int q = 0; q < 10; q++) c[q] = 0;
        for (int r = 0; r < k; r++) {
            uint32_t acc = 0;
            for (uint32_t e = row_off[r]; e < row_off[r + 1]; e++) acc ^= cur[(size_t)taps[e] * nlanes + LG];
            nxt[(size_t)r * nlanes + LG] = acc;
            uint32_t carry = (__builtin_popcount((unsigned)r) & 1) ? ~acc : acc;   // weight -1 bits enter complemented
#pragma unroll
            for (int q = 0; q < 10; q++) {
                const uint32_t tcar = c[q] & carry;
                c[q] ^= carry;
                carry = tcar;
            }
        }
        for (unsigned j = 0; j < 32; j++) {
            const unsigned long long g = gen_index(wave, lane, j);
            const unsigned long long off = g * L + t;
            if (g < G && off < nsamples) {
                int T = 0;
#pragma unroll
                for (int q = 0; q < 10; q++) T |= (int)((c[q] >> j) & 1u) << q;
                int v = (T - (k >> 1)) & (k - 1);                 // log2(k)-bit two's complement (rng.py:78,108)
                if (v >> (logk - 1)) v -= k;
                dst[off] = (OutT)v;
            }
        }
        uint32_t *sw = cur; cur = nxt; nxt = sw;
    }
}

// ---------------------------------------------------------------------------------------------
// The uniform word stream LUTOPT.x in bulk (rng.py:29-40 "outputs x on each clock"; the reference
// dumps it for dieharder in software/rnghunt/util/verify.py:37-52): state t as k/32 consecutive
// 32-bit words, word j = state bits 32j..32j+31 with bit 32j the LSB, or -- msb_first, verify.py's
// text-to-int conversion -- the MSB.  Table driven like awgn_generic_kernel (any k multiple of 32,
// any taps); per step every 32-plane block is transposed back to one word per generator.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
lutopt_words_kernel(int k, const uint16_t *__restrict taps, const uint32_t *__restrict row_off,
                    uint32_t *__restrict planes2, uint32_t *__restrict dst, unsigned long long nstates, unsigned L,
                    unsigned long long G, unsigned nlanes, int msb_first) {
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    const unsigned wps = (unsigned)k / 32;
    uint32_t *cur = planes2, *nxt = planes2 + (size_t)k * nlanes;
    for (unsigned t = 0; t < L; t++) {
        for (int r = 0; r < k; r++) {
            uint32_t acc = 0;
            for (uint32_t e = row_off[r]; e < row_off[r + 1]; e++) acc ^= cur[(size_t)taps[e] * nlanes + LG];
            nxt[(size_t)r * nlanes + LG] = acc;
        }
        for (unsigned wq = 0; wq < wps; wq++) {
            uint32_t q[32];
#pragma unroll
            for (int p = 0; p < 32; p++) q[p] = nxt[(size_t)(32 * wq + p) * nlanes + LG];
            transpose32(q);                      // q[j] bit p = state bit 32wq+p of generator j
#pragma unroll
            for (unsigned j = 0; j < 32; j++) {
                const unsigned long long g = gen_index(wave, lane, j);
                const unsigned long long off = g * L + t;
                if (g < G && off < nstates) dst[off * wps + wq] = msb_first ? __builtin_bitreverse32(q[j]) : q[j];
            }
        }
        uint32_t *sw = cur; cur = nxt; nxt = sw;
    }
}

// ---------------------------------------------------------------------------------------------
// Adder tree on caller-supplied words (clt-grng-evaluate.py:8-16), closed form: +1 weight where
// popcount(bit index) is even.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
clt_tree_kernel(int nwords, const unsigned long long *__restrict states, unsigned long long nstates,
                int16_t *__restrict out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstates) return;
    int pos = 0, all = 0;
    for (int w = 0; w < nwords; w++) {
        const unsigned long long x = states[i * nwords + w];
        const unsigned long long m = (__builtin_popcount((unsigned)w) & 1) ? ~kThueMorse64 : kThueMorse64;
        pos += __builtin_popcountll(x & m);
        all += __builtin_popcountll(x);
    }
    out[i] = (int16_t)(2 * pos - all);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <int W32>
static int seed_and_slice(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states,
                          uint64_t stride, unsigned nlanes, uint32_t *d_planes, hipStream_t st, int slice_mode, int parts) {
    parts = (int)env_knob("BBB_SEED_PARTS", parts);
    Seed16 s;
    for (int i = 0; i < 16; i++)
        for (int w = 0; w < 16; w++) s.w[i][w] = w < W32 ? s16[i * 16 + w] : 0u;
    int levels = 0;                                   // radix-16 levels needed: 16^levels >= G
    while ((1ull << (4 * levels)) < G) levels++;
    if (parts != 0 && parts != 2 && parts != 4) return fail(BBB_EINVAL, "seeding stages its tables in 2 or 4 pieces, or reads them in place (0)");
    const int nnib_h = (k + 3) / 4, half_h = parts ? seed_part_nibbles(nnib_h, parts) : 0;
    size_t lds = (size_t)half_h * 16 * W32 * sizeof(uint32_t);      // one piece of a table at a time
    if (env_knob("BBB_SEED_LDS_KB", 0) > 0 && lds < (size_t)env_knob("BBB_SEED_LDS_KB", 0) * 1024)
        lds = (size_t)env_knob("BBB_SEED_LDS_KB", 0) * 1024;        // (experiments: fewer seeding blocks per CU beside the sample kernel)
    if (lds > 48 * 1024) {     // per device (hipFuncSetAttribute applies to the current one), guarded
        static std::mutex mu;
        static bool attr_set[64] = {false};
        int dev = 0;
        BBB_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> g(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            BBB_HIP(hipFuncSetAttribute((const void *)seed_level_kernel<W32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    // (tried in round 4: levels 1..3 composed per thread from the tables in global memory, one launch instead of four -- 46 us
    // where the four take 33: uncoalesced 32-byte lookups pull a 128-byte line each through the vector cache)
    hipLaunchKernelGGL((seed_store16_kernel<W32>), dim3(1), dim3(64), 0, st, s, (unsigned long long)G,
                       (unsigned long long)stride, d_states);
    for (int e = 1; e < levels; e++) {               // level 0 (states 1..15) was done on the host
        const uint64_t n = 1ull << (4 * e);
        hipLaunchKernelGGL((seed_level_kernel<W32>), dim3((unsigned)((n + 255) / 256), 15), dim3(256), lds, st, d_tabs, k, e,
                           (unsigned long long)G, (unsigned long long)stride, d_states, parts);
    }
    if (slice_mode == 1) return bitslice512p_launch(d_states, G, stride, nlanes, d_planes, st);
    const uint64_t threads = (uint64_t)nlanes * (uint64_t)((k + 31) / 32);
    hipLaunchKernelGGL((bitslice_kernel<W32>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_states,
                       (unsigned long long)G, (unsigned long long)stride, nlanes, k, d_planes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

static int make_prbs_jump(int k, const uint32_t *s16, const uint32_t *qcol, uint64_t G, PrbsLaneJump *jp, int *levels) {
    if (k < 2 || k > 31) return fail(BBB_EINVAL, "PRBS order must be below 32");
    *jp = PrbsLaneJump{};
    for (int i = 0; i < 16; i++) jp->first[i] = s16[i * 16];
    for (int c = 0; c < 32; c++) jp->qcol[c] = c < k ? qcol[c] : 0u;
    int lv = 0;
    while ((1ull << (4 * lv)) < G) lv++;
    if (lv < 1) lv = 1;
    if (lv > 7) return fail(BBB_EINVAL, "too many generators for the PRBS jump plan");
    *levels = lv;
    return BBB_OK;
}

int awgn_seed_head_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states, hipStream_t st,
                          const PrbsSeedRide *ride) {
    constexpr int W32 = 8;
    if (k != 256 || G > ((uint64_t)kSeedTopTables + 1) * 65536) return fail(BBB_EINVAL, "two-launch seeding: k = 256, at most 2^21 generators");
    Seed16 s;
    for (int i = 0; i < 16; i++)
        for (int w = 0; w < 16; w++) s.w[i][w] = w < W32 ? s16[i * 16 + w] : 0u;
    PrbsRide pr{};
    if (ride) {
        const int rc = make_prbs_jump(ride->k, ride->s16, ride->qcol, G, &pr.jp, &pr.levels);
        if (rc) return rc;
        pr.tabs = ride->d_tabs; pr.planes = ride->d_planes; pr.k = ride->k; pr.nlanes = ride->nlanes; pr.nblocks = (ride->nlanes + 255) / 256;
    }
    const size_t lds = (size_t)2 * 64 * 16 * W32 * sizeof(uint32_t);              // two whole tables: 64 KiB of dynamic LDS
    {   // (per device: hipFuncSetAttribute applies to the current one)
        static std::mutex mu;
        static bool attr_set[64] = {false};
        int dev = 0;
        BBB_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> g(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            BBB_HIP(hipFuncSetAttribute((const void *)seed_head_kernel<W32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    const uint64_t head = G < 65536 ? G : 65536;
    const unsigned nhead = (unsigned)((head + 255) / 256);
    hipLaunchKernelGGL((seed_head_kernel<W32>), dim3(nhead + pr.nblocks), dim3(256), lds, st, d_tabs, s, k,
                       (unsigned long long)G, 65536ull, d_states, nhead, pr);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int awgn_seed_tail_planes_launch(int k, const uint32_t *d_top, uint64_t G, const uint32_t *d_states, unsigned nlanes, uint32_t *d_planes,
                                 hipStream_t st) {
    if (k != 256 || G > ((uint64_t)kSeedTopTables + 1) * 65536 || !d_top || (uint64_t)nlanes * 32 < G || nlanes % 64)
        return fail(BBB_EINVAL, "two-launch seeding: k = 256, at most 2^21 generators");
    const size_t lds = (size_t)64 * 16 * 8 * sizeof(uint32_t);                    // one table: 32 KiB
    hipLaunchKernelGGL(seed_tail_planes_kernel, dim3(nlanes / 64), dim3(256), lds, st, d_top, (unsigned long long)G, 65536ull, d_states,
                       nlanes, d_planes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int prbs_seed_planes_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states, unsigned nlanes,
                            uint32_t *d_planes, hipStream_t st) {
    if (k < 2 || k > 31) return fail(BBB_EINVAL, "PRBS order must be below 32");
    Seed16 s{};
    for (int i = 0; i < 16; i++) s.w[i][0] = s16[i * 16];
    int levels = 0;
    while ((1ull << (4 * levels)) < G) levels++;
    if (levels < 1) levels = 1;
    if (levels > 7) return fail(BBB_EINVAL, "too many generators for the PRBS jump plan");
    const size_t lds = (size_t)(levels - 1) * 15 * ((k + 3) / 4) * 16 * sizeof(uint32_t);        // <= 45 KiB
    hipLaunchKernelGGL(prbs_seed_states_kernel, dim3((unsigned)((G + 2047) / 2048)), dim3(256), lds, st, d_tabs, s, k, levels,
                       (unsigned long long)G, d_states);
    hipLaunchKernelGGL((bitslice_kernel<1>), dim3((nlanes + 255) / 256), dim3(256), 0, st, d_states, (unsigned long long)G,
                       (unsigned long long)G, nlanes, k, d_planes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int prbs_seed_lanes_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, const uint32_t *qcol, uint64_t G, unsigned nlanes,
                           uint32_t *d_planes, hipStream_t st) {
    PrbsLaneJump jp;
    int levels = 0;
    const int rc = make_prbs_jump(k, s16, qcol, G, &jp, &levels);
    if (rc) return rc;
    hipLaunchKernelGGL(prbs_seed_lanes_kernel, dim3((nlanes + 255) / 256), dim3(256), 0, st, d_tabs, jp, k, levels, (unsigned long long)G, nlanes,
                       d_planes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int awgn_seed_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states,
                     uint64_t stride, unsigned nlanes, uint32_t *d_planes, hipStream_t st, int slice_mode, int parts) {
    switch ((k + 31) / 32) {
    case 1: return seed_and_slice<1>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 2: return seed_and_slice<2>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 3: case 4: return seed_and_slice<4>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 5: case 6: return seed_and_slice<6>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 7: case 8: return seed_and_slice<8>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 9: case 10: case 11: case 12: return seed_and_slice<12>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    case 13: case 14: case 15: case 16: return seed_and_slice<16>(k, d_tabs, s16, G, d_states, stride, nlanes, d_planes, st, slice_mode, parts);
    default: return fail(BBB_EINVAL, "k must be in [2, 512]");
    }
}

int awgn256_fill_launch(const uint32_t *d_planes, int8_t *dst, uint64_t nsamples, unsigned L, uint64_t G,
                        unsigned nlanes, hipStream_t st) {
    const unsigned nwaves = nlanes / 64;
    TxFuse none{};
    hipLaunchKernelGGL((awgn256_kernel<false>), dim3(nwaves), dim3(64), 0, st, d_planes, (void *)dst, (unsigned long long)nsamples, L,
                       (unsigned long long)G, nlanes, none);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// data bits of the pulse source (tx.py:20-30: a 1 every 256 bit periods), packed like the PRBS generator's
__global__ void __launch_bounds__(256)
pulse_bits_kernel(unsigned long long *__restrict dst, long long m_first, unsigned long long nwords) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    unsigned long long w = 0;
    for (int j = 0; j < 64; j++) {
        const long long m = m_first + (long long)(64 * i) + j;
        if (m >= 0 && (m & 255) == 0) w |= 1ull << j;
    }
    dst[i] = w;
}

int pulse_bits_launch(uint64_t *dst, int64_t m_first, uint64_t nwords, hipStream_t st) {
    if (!nwords) return BBB_OK;
    hipLaunchKernelGGL(pulse_bits_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, (unsigned long long *)dst,
                       (long long)m_first, (unsigned long long)nwords);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int awgn256_tx_launch(const uint32_t *d_planes, int16_t *dst, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                      const int16_t *coeffs, const uint32_t *d_bits, uint32_t nwords32, uint32_t rel_base, uint32_t c0, int noise_var,
                      int bit_en, int use_bits, hipStream_t st) {
    const TxFuse tx = make_txfuse(coeffs, d_bits, nwords32, rel_base, c0, noise_var, bit_en, use_bits);
    hipLaunchKernelGGL((awgn256_kernel<true>), dim3(nlanes / 64), dim3(64), 0, st, d_planes, (void *)dst, (unsigned long long)nsamples, L,
                       (unsigned long long)G, nlanes, tx);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int awgn_generic_fill_launch(int k, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2,
                             void *dst, int elem_size, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                             hipStream_t st) {
    int logk = 0;
    while ((1 << logk) < k) logk++;
    const unsigned nwaves = nlanes / 64;
    if (elem_size == 1)
        hipLaunchKernelGGL((awgn_generic_kernel<int8_t>), dim3(nwaves), dim3(64), 0, st, k, logk, d_taps, d_row_off,
                           d_planes2, (int8_t *)dst, (unsigned long long)nsamples, L, (unsigned long long)G, nlanes);
    else
        hipLaunchKernelGGL((awgn_generic_kernel<int16_t>), dim3(nwaves), dim3(64), 0, st, k, logk, d_taps, d_row_off,
                           d_planes2, (int16_t *)dst, (unsigned long long)nsamples, L, (unsigned long long)G, nlanes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// The word stream of the shipped n256 matrix on the generated network: the step of the sample kernel (its counter is
// dead code here), then the 256 new planes go back to one word per generator by eight 32 x 32 bit transposes, two at a
// time in the registers the old state has just vacated -> four 8-byte stores per generator and step (one 32-byte piece
// once L2 has merged them).  MSB: bit 32j of the state is the MSB of word j (verify.py:46-52) -- the planes simply enter
// the transposes in reverse order.
template <bool MSB>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
words256_kernel(const uint32_t *__restrict planes, uint32_t *__restrict dst, unsigned long long nstates, unsigned L,
                unsigned long long G, unsigned nlanes) {
    const unsigned lane = threadIdx.x;
    const unsigned long long wave = blockIdx.x;
    const unsigned long long LG = wave * 64 + lane;
    __builtin_amdgcn_s_setprio(3);
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
#pragma unroll
    for (int p = 0; p < 256; p++) a[p] = planes[(size_t)p * nlanes + LG];
#define BBB_PARK(p) BBB_ACC_WRITE(pa[p], a[p]);
    LUTOPT256_FOR_PARKED(BBB_PARK)
#undef BBB_PARK
    // emit state `x` (VGPR part) / `px` (AGPR part) as state number t of every generator of this lane
    auto emit = [&](const uint32_t (&x)[256], const uint32_t (&px)[256], const unsigned t) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            uint32_t q[2][32];
#pragma unroll
            for (int w = 0; w < 2; w++) {
#pragma unroll
                for (int e = 0; e < 32; e++) {
                    const int p = 32 * (2 * h + w) + (MSB ? 31 - e : e);
                    if (lutopt256_is_parked(p)) BBB_ACC_READ(q[w][e], px[p]);
                    else q[w][e] = x[p];
                }
                transpose32(q[w]);                // q[w][j] = word 2h + w of generator j
            }
#pragma unroll
            for (unsigned j = 0; j < 32; j++) {
                const unsigned long long g = gen_index(wave, lane, j);
                const unsigned long long st = g * L + t;
                if (g < G && st < nstates) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = {q[0][j], q[1][j]};
                    *reinterpret_cast<u32x2 *>(dst + st * 8 + 2 * h) = v;
                }
            }
        }
    };
#pragma unroll 1
    for (unsigned t = 0; t < L; t += 2) {
        lutopt256_step_parked(a, pa, b, pb, cnt);
        emit(b, pb, t);
        lutopt256_step_parked(b, pb, a, pa, cnt);
        if (t + 1 < L) emit(a, pa, t + 1);
    }
}

int lutopt_words256_launch(const uint32_t *d_planes, uint32_t *dst, uint64_t nstates, unsigned L, uint64_t G, unsigned nlanes,
                           bool msb_first, hipStream_t st) {
    if (msb_first)
        hipLaunchKernelGGL(words256_kernel<true>, dim3(nlanes / 64), dim3(64), 0, st, d_planes, dst, (unsigned long long)nstates, L,
                           (unsigned long long)G, nlanes);
    else
        hipLaunchKernelGGL(words256_kernel<false>, dim3(nlanes / 64), dim3(64), 0, st, d_planes, dst, (unsigned long long)nstates, L,
                           (unsigned long long)G, nlanes);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int lutopt_words_launch(int k, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2, uint32_t *dst,
                        uint64_t nstates, unsigned L, uint64_t G, unsigned nlanes, bool msb_first, hipStream_t st) {
    hipLaunchKernelGGL(lutopt_words_kernel, dim3(nlanes / 64), dim3(64), 0, st, k, d_taps, d_row_off, d_planes2, dst,
                       (unsigned long long)nstates, L, (unsigned long long)G, nlanes, msb_first ? 1 : 0);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

int clt_tree_launch(int k, const uint64_t *states, uint64_t nstates, int16_t *out, hipStream_t st) {
    if (nstates == 0) return BBB_OK;
    hipLaunchKernelGGL(clt_tree_kernel, dim3((unsigned)((nstates + 255) / 256)), dim3(256), 0, st, (k + 63) / 64,
                       (const unsigned long long *)states, (unsigned long long)nstates, out);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

bool awgn256_matches(int k, const uint16_t *taps, const uint32_t *row_off) {
    if (k != 256) return false;
    uint32_t e = 0;
    for (int r = 0; r < 256; r++) {
        if (row_off[r + 1] - row_off[r] != LUTOPT256_NTAPS[r]) return false;
        for (uint32_t j = row_off[r]; j < row_off[r + 1]; j++)
            if (taps[j] != LUTOPT256_TAPS[e++]) return false;
    }
    return true;
}

}  // namespace bbb
