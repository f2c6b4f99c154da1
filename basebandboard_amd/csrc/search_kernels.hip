// The recurrence search of software/rnghunt (src/bin/rnghunt.rs:20-47) on the GPU: one candidate
// matrix per wavefront at a time --
//   build      random k x k matrix, 3 or 4 ones per row, balanced columns      (search_rng.hpp)
//   recur      2k steps x' = A x from the all-ones state, bit 0 of every state (binary_matrix.rs:68-76)
//   BM         minimal polynomial of the reversed sequence                     (berlekamp_massey.rs:5-31)
//   accept     degree == k and primitive                                       (binary_polynomial.rs:178-216)
//
// Data layout per wave: polynomials and bit vectors are DISTRIBUTED, index i lives in lane i % 64,
// bit i / 64 of that lane's register (R = ceil(k/64) bits per lane).  "Multiply by x" is a one-lane
// shift (DPP wave_shr) plus a carry from lane 63 into the next bit of lane 0; dot products are a
// per-lane AND + popcount and one ballot; squaring modulo p uses, per lane, the columns of the matrix
// of x^(2i) mod p (k/2 <= i < k) kept in registers, so that a squaring is R ballots and R*k/64
// AND+popcount pairs with no memory traffic.  The sparse matrix-vector steps keep the state as one
// byte per bit in LDS (every lane reads the 3-4 taps of its R rows).
//
// Integer / bit work throughout; no MFMA; LDS traffic only in the build and recur phases.
#include "bbb_common.hpp"
#include "gf2poly.hpp"
#include "search_rng.hpp"

#include <vector>

namespace bbb {

typedef unsigned long long u64;

struct SearchOut {
    u64 found;           // smallest accepted candidate (atomicMin), ~0 if none
    u64 tested, full_degree, order_divides, primitive;
    u64 kernel_ns;       // filled by the host from HIP events
};

// index i <- index i-1, index 0 <- ins (uniform 0/1)
__device__ __forceinline__ uint32_t dist_shift1(uint32_t v, uint32_t ins, unsigned lane) {
    const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    if (lane == 0) t = (top << 1) | ins;
    return t;
}

__device__ __forceinline__ uint32_t dist_getbit(uint32_t v, int i) {       // i uniform
    return ((uint32_t)__builtin_amdgcn_readlane((int)v, i & 63) >> (i >> 6)) & 1u;
}

__device__ __forceinline__ uint32_t dist_parity_dot(uint32_t a, uint32_t b) {
    return (uint32_t)__builtin_popcountll(__ballot((__builtin_popcount(a & b) & 1) != 0)) & 1u;
}

template <int K>
__global__ void __launch_bounds__(64)
search_kernel(u64 seed, u64 first, u64 count, const u64 *__restrict exps, int nexp, SearchOut *__restrict out) {
    constexpr int R = (K + 63) / 64;          // indices per lane
    constexpr int H = K / 2;                  // x^(2i), i >= H, needs reduction
    constexpr int HW = (H + 31) / 32;         // 32-bit chunks of the upper half
    constexpr int EW = (K + 63) / 64;         // words per exponent
    __shared__ uint32_t keys[K];
    __shared__ uint16_t perm[4][K];
    __shared__ uint8_t st[2][K + 4];
    __shared__ uint8_t seq[2 * K];
    const unsigned lane = threadIdx.x;
    const u64 nwaves = gridDim.x;

    for (u64 ci = blockIdx.x; ci < count; ci += nwaves) {
        const u64 cand = first + ci;
        if (*(volatile u64 *)&out->found < cand) break;       // a smaller accepted candidate exists: nothing left for this wave
        // ---- build the matrix -------------------------------------------------------------
        int w[R];
        int P[R];
        int before = 0;
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int r = s * 64 + (int)lane;
            w[s] = r < K ? search_row_weight(seed, cand, r) : 4;
            const u64 three = __ballot(r < K && w[s] == 3);
            P[s] = 4 * r - before - __builtin_popcountll(three & ((1ull << lane) - 1ull));
            before += __builtin_popcountll(three);
        }
        const int total = 4 * K - before;
        const int rounds = (total + K - 1) / K;
        for (int j = 0; j < rounds; j++) {
            uint32_t mykey[R];
            int rank[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int c = s * 64 + (int)lane;
                mykey[s] = c < K ? search_perm_key(seed, cand, j, K, c) : 0xffffffffu;
                rank[s] = 0;
                if (c < K) keys[c] = mykey[s];
            }
            __syncthreads();
            for (int c = 0; c < K; c++) {
                const uint32_t kc = keys[c];
#pragma unroll
                for (int s = 0; s < R; s++) rank[s] += kc < mykey[s] ? 1 : 0;
            }
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int c = s * 64 + (int)lane;
                if (c < K) perm[j][rank[s]] = (uint16_t)c;
            }
            __syncthreads();
        }
        // rows that straddle two permutations must not repeat a column
        for (int j = 1; j < rounds; j++) {
            const int edge = j * K;
            u64 hit = 0;
            int slot = -1;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int r = s * 64 + (int)lane;
                const u64 b = __ballot(r < K && P[s] < edge && edge < P[s] + w[s]);
                if (b && slot < 0) { hit = b; slot = s; }
            }
            if (slot >= 0) {
                const int src = __builtin_ctzll(hit);
                int Pr = 0, wr = 0;
#pragma unroll
                for (int s = 0; s < R; s++)
                    if (s == slot) { Pr = __builtin_amdgcn_readlane(P[s], src); wr = __builtin_amdgcn_readlane(w[s], src); }
                const int ntail = edge - Pr, h = Pr + wr - edge;
                int t = 0;
                for (int hp = 0; hp < h; hp++) {
                    for (;;) {
                        const uint16_t v = perm[j][hp];
                        bool clash = false;
                        for (int q = 0; q < ntail; q++) clash |= v == perm[j - 1][K - ntail + q];
                        if (!clash) break;
                        const uint16_t o = perm[j][h + t];
                        __syncthreads();
                        if (lane == 0) { perm[j][hp] = o; perm[j][h + t] = v; }
                        __syncthreads();
                        t++;
                    }
                }
            }
        }
        int tap[R][4];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int r = s * 64 + (int)lane;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int pos = P[s] + q;
                tap[s][q] = (r < K && q < w[s]) ? (int)perm[pos / K][pos % K] : K;     // st[.][K] is a constant 0
            }
        }
        // ---- 2k steps from the all-ones state; seq[t] = bit 0 of state t+1 -----------------
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int r = s * 64 + (int)lane;
            if (r < K) st[0][r] = 1;
        }
        if (lane == 0) { st[0][K] = 0; st[1][K] = 0; }
        __syncthreads();
        for (int t = 0; t < 2 * K; t++) {
            const uint8_t *cur = st[t & 1];
            uint8_t *nxt = st[(t & 1) ^ 1];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int r = s * 64 + (int)lane;
                const uint8_t v = cur[tap[s][0]] ^ cur[tap[s][1]] ^ cur[tap[s][2]] ^ cur[tap[s][3]];
                if (r < K) nxt[r] = v;
                if (r == 0) seq[t] = v;
            }
            __syncthreads();
        }
        // ---- Berlekamp-Massey on the reversed sequence (distributed C, B, window) ------------
        uint32_t C = lane == 0 ? 1u : 0u, B = C, Wn = 0;
        int L = 0;
        for (int i = 0; i < 2 * K; i++) {
            const uint32_t si = seq[2 * K - 1 - i];
            Wn = dist_shift1(Wn, si, lane);               // window index j = s[i - j]
            B = dist_shift1(B, 0, lane);                  // B * x^(i - m)
            if (dist_parity_dot(C, Wn)) {
                const uint32_t T = C;
                C ^= B;
                if (2 * L <= i) { L = i + 1 - L; B = T; }
            }
        }
        if (lane == 0) atomicAdd(&out->tested, 1ull);
        // C is the connection polynomial (C_0 = 1); the reference examines its reciprocal, which is
        // primitive exactly when C is.  Degree k in both senses is required (rnghunt.rs:40-42).
        if (L != K || !dist_getbit(C, K)) continue;
        if (lane == 0) atomicAdd(&out->full_degree, 1ull);
        // odd number of terms (binary_polynomial.rs:191-193)
        if (!(__builtin_popcountll(__ballot(__builtin_popcount(C) & 1)) & 1)) continue;
        const uint32_t p = C;
        // ---- columns of Q: col[s][w] bit b = coefficient (s*64 + lane) of x^(2i) mod p, i = H + 32w + b
        uint32_t col[R][HW];
#pragma unroll
        for (int s = 0; s < R; s++)
#pragma unroll
            for (int q = 0; q < HW; q++) col[s][q] = 0;
        {
            // g = x^(2H) mod p = x^K mod p (K even) = p without its leading term
            uint32_t g = p;
            if (lane == (K & 63)) g &= ~(1u << (K >> 6));
#pragma unroll
            for (int q = 0; q < HW; q++) {
                for (int b = 0; b < 32 && q * 32 + b < H; b++) {
#pragma unroll
                    for (int s = 0; s < R; s++) col[s][q] |= ((g >> s) & 1u) << b;
                    for (int twice = 0; twice < 2; twice++) {
                        g = dist_shift1(g, 0, lane);
                        if (dist_getbit(g, K)) g ^= p;
                    }
                }
            }
        }
        // ---- x^e mod p for the exponents of the test ----------------------------------------
        bool ok = true;
        for (int e = 0; e < nexp && ok; e++) {
            const u64 *ew = exps + (size_t)e * EW;
            int top = -1;
            for (int q = EW - 1; q >= 0 && top < 0; q--)
                if (ew[q]) top = 64 * q + 63 - __builtin_clzll(ew[q]);
            uint32_t f = lane == 1 ? 1u : 0u;                       // x
            if (top < 0) f = lane == 0 ? 1u : 0u;
            for (int b = top - 1; b >= 0; b--) {
                // f <- f^2 mod p
                u64 F[R + 1];
#pragma unroll
                for (int s = 0; s < R; s++) F[s] = __ballot(((f >> s) & 1u) != 0);
                F[R] = 0;
                uint32_t hi[HW];
#pragma unroll
                for (int q = 0; q < HW; q++) {
                    const int off = H + 32 * q;                     // bit offset into F
                    const u64 lo = F[off >> 6] >> (off & 63);
                    const u64 up = (off & 63) ? F[(off >> 6) + 1] << (64 - (off & 63)) : 0ull;
                    uint32_t v = (uint32_t)(lo | up);
                    if (H - 32 * q < 32) v &= (1u << (H - 32 * q)) - 1u;
                    hi[q] = v;
                }
                uint32_t nf = 0;
#pragma unroll
                for (int s = 0; s < R; s++) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int q = 0; q < HW; q++) acc += (uint32_t)__builtin_popcount(hi[q] & col[s][q]);
                    // spread of the lower half: coefficient j = s*64 + lane, j even, takes f_(j/2) (j/2 < H)
                    const int half = s * 32 + (int)(lane >> 1);
                    const uint32_t lowbit = ((lane & 1u) == 0u && half < H) ? (uint32_t)((F[s >> 1] >> (((s & 1) << 5) + (lane >> 1))) & 1ull) : 0u;
                    nf |= ((acc ^ lowbit) & 1u) << s;
                }
                f = nf;
                if ((ew[b >> 6] >> (b & 63)) & 1ull) {
                    f = dist_shift1(f, 0, lane);
                    if (dist_getbit(f, K)) f ^= p;
                }
            }
            const bool is_one = __ballot(f != (lane == 0 ? 1u : 0u)) == 0ull;
            if (e == 0) {
                ok = is_one;                                        // x^(2^k - 1) must be 1
                if (ok && lane == 0) atomicAdd(&out->order_divides, 1ull);
            } else {
                ok = !is_one;                                       // x^((2^k - 1)/q) must not be
            }
        }
        if (!ok) continue;
        if (lane == 0) {
            atomicAdd(&out->primitive, 1ull);
            atomicMin(&out->found, cand);
        }
    }
}

template <int K>
static int search_run(u64 seed, u64 first, u64 count, const MersenneEntry *me, SearchOut *h_out, hipStream_t st) {
    constexpr int EW = (K + 63) / 64;
    u64 *d_exps = nullptr;
    SearchOut *d_out = nullptr;
    BBB_HIP(hipMalloc(&d_exps, (size_t)me->nexp * EW * sizeof(u64)));
    hipError_t e = hipMalloc(&d_out, sizeof(SearchOut));
    if (e != hipSuccess) { (void)hipFree(d_exps); BBB_HIP(e); }
    SearchOut init = {~0ull, 0, 0, 0, 0, 0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    (void)hipEventCreate(&ev0);
    (void)hipEventCreate(&ev1);
    e = hipMemcpyAsync(d_exps, (const void *)(kMersenneWords + me->offset), (size_t)me->nexp * EW * sizeof(u64), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out, &init, sizeof init, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        int dev = 0, ncu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const u64 maxwaves = (u64)ncu * 16;
        const unsigned grid = (unsigned)(count < maxwaves ? count : maxwaves);
        (void)hipEventRecord(ev0, st);
        hipLaunchKernelGGL(search_kernel<K>, dim3(grid), dim3(64), 0, st, seed, first, count, (const u64 *)d_exps, me->nexp, d_out);
        (void)hipEventRecord(ev1, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, sizeof(SearchOut), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0.f;
    if (e == hipSuccess && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) h_out->kernel_ns = (u64)(ms * 1e6);
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    (void)hipFree(d_exps);
    (void)hipFree(d_out);
    BBB_HIP(e);
    return BBB_OK;
}

int lutopt_search_launch(int k, uint64_t seed, uint64_t first, uint64_t count, uint64_t *found, uint16_t *taps_out,
                         uint32_t *row_off_out, bbb_search_stats *stats, hipStream_t st) {
    const MersenneEntry *me = mersenne_entry(k);
    if (!me) return fail(BBB_EUNSUP, "no factorisation of 2^" + std::to_string(k) + " - 1 in the table");
    SearchOut o = {~0ull, 0, 0, 0, 0, 0};
    int rc = BBB_OK;
    if (count) {
        switch (k) {
        case 16: rc = search_run<16>(seed, first, count, me, &o, st); break;
        case 32: rc = search_run<32>(seed, first, count, me, &o, st); break;
        case 64: rc = search_run<64>(seed, first, count, me, &o, st); break;
        case 128: rc = search_run<128>(seed, first, count, me, &o, st); break;
        case 192: rc = search_run<192>(seed, first, count, me, &o, st); break;
        case 256: rc = search_run<256>(seed, first, count, me, &o, st); break;
        case 512: rc = search_run<512>(seed, first, count, me, &o, st); break;
        default: return fail(BBB_EUNSUP, "search supports k = 16, 32, 64, 128, 192, 256, 512");
        }
    }
    if (rc) return rc;
    if (stats) {
        stats->tested = o.tested;
        stats->full_degree = o.full_degree;
        stats->order_divides = o.order_divides;
        stats->primitive = o.primitive;
        stats->kernel_ns = o.kernel_ns;
    }
    if (found) *found = o.found;
    if (o.found != ~0ull) {
        // the accepted matrix is rebuilt and re-examined with the host arithmetic before it is handed out
        std::vector<uint16_t> t4;
        std::vector<uint8_t> w;
        search_candidate_host(k, seed, o.found, t4, w);
        std::vector<uint16_t> taps;
        std::vector<uint32_t> off(k + 1, 0);
        for (int r = 0; r < k; r++) {
            off[r] = (uint32_t)taps.size();
            for (int q = 0; q < w[r]; q++) taps.push_back(t4[(size_t)4 * r + q]);
        }
        off[k] = (uint32_t)taps.size();
        GF2Poly P;
        const int L = lutopt_charpoly(k, taps.data(), off.data(), P);
        if (L != k || gf2_is_primitive(P) != 1)
            return fail(BBB_EHIP, "search: device accepted candidate " + std::to_string(o.found) + " but the host check rejects it");
        if (taps_out) std::copy(taps.begin(), taps.end(), taps_out);
        if (row_off_out) std::copy(off.begin(), off.end(), row_off_out);
    }
    return BBB_OK;
}

}  // namespace bbb
