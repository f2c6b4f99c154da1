// Round 5: does any store flavour leave the memory-side cache clean, so that a read pass right behind a write pass runs as fast as on a
// clean buffer?  1.25 GB, 1024 one-wave blocks (the PRBS kernels' partition), stores by inline asm with the gfx950 cache-policy bits,
// then a non-temporal read pass over the same bytes (ascending and, like the shipped checker, every region from its end).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int FL>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v) {
    if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
}
// RND: every stored dword is pseudo-random (what a PRBS is) instead of three constants and a counter
template <int FL, bool RND = false>
__global__ void __launch_bounds__(64) wr(char *buf, unsigned long long total, unsigned long long chunk) {
    const unsigned lane = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    u32x4 acc = {lane, blockIdx.x, 3u, 4u};
    if (RND) { acc.x = lane * 2654435761u ^ blockIdx.x * 40503u; acc.y = acc.x * 1664525u + 1013904223u; acc.z = acc.y * 1664525u + 1013904223u; acc.w = acc.z * 1664525u + 1013904223u; }
    for (unsigned long long o = lo; o < hi; o += 16 * 1024) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned long long a = o + (unsigned long long)i * 1024;
            if (a >= hi) break;
            acc.x += 0x9e3779b9u;
            if (RND) { acc.y ^= acc.x << 7; acc.z += acc.y ^ (acc.x >> 3); acc.w ^= acc.z * 0x85ebca6bu; acc.x ^= acc.w >> 5; }
            store16<FL>(reinterpret_cast<u32x4 *>(buf + a) + lane, acc);
        }
    }
}
// the PRBS generator's shape: a window of K = 31 rows in registers, per pass K in-place XORs (row[i] ^= row[i - 3]) and K stores of one
// row each; LW = bytes per lane (8: rows of 512 B as shipped, 16: rows of 1 KiB); ASM: stores by inline asm (ordered, as above) or C++
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// ILV: every row is stored right behind its own XOR (asm volatile keeps that order) instead of K XORs, then K stores
template <int LW, bool ASM, bool ILV = false>
__global__ void __launch_bounds__(64, 2) wr_prbs(char *buf, unsigned long long total, unsigned long long chunk) {
    constexpr int K = 31, NW = LW / 4;
    const unsigned lane = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t V[K][NW];
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int w = 0; w < NW; w++) V[i][w] = lane * 2654435761u + i * 40503u + w;
    constexpr unsigned ROW = 64 * LW;
    for (unsigned long long o = lo; o + (unsigned long long)K * ROW <= hi; o += (unsigned long long)K * ROW) {
        if (!ILV) {
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
            for (int w = 0; w < NW; w++) asm("v_xor_b32 %0, %0, %1" : "+v"(V[i][w]) : "v"(V[(i + K - 3) % K][w]));
        }
#pragma unroll
        for (int i = 0; i < K; i++) {
            if (ILV) {
#pragma unroll
                for (int w = 0; w < NW; w++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(V[i][w]) : "v"(V[(i + K - 3) % K][w]));
            }
            char *p = buf + o + (unsigned long long)i * ROW + lane * LW;
            if (LW == 16) {
                const u32x4 v = {V[i][0], V[i][1], V[i][2 % NW], V[i][3 % NW]};
                if (ASM) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory"); else *reinterpret_cast<u32x4 *>(p) = v;
            } else {
                const u32x2 v = {V[i][0], V[i][1 % NW]};
                if (ASM) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); else *reinterpret_cast<u32x2 *>(p) = v;
            }
        }
    }
}
template <int LW, bool ASM, bool ILV = false> void run_prbs(const char *name, char *buf, unsigned long long total) {
    const unsigned long long unit = 31ull * 64 * LW;
    const unsigned long long chunk = ((total + 1023) / 1024 + unit - 1) / unit * unit;
    const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float w = 0;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((wr_prbs<LW, ASM, ILV>), dim3(grid), dim3(64), 0, 0, buf, total, chunk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float a; hipEventElapsedTime(&a, e0, e1);
        if (rep >= 2) w += a / 5;
    }
    printf("%-44s grid %4u: write %.4f ms (%.2f TB/s)\n", name, grid, w, total / w / 1e9);
}
template <bool REV>
__global__ void __launch_bounds__(64) rd(const char *buf, unsigned long long total, unsigned long long chunk, unsigned long long *sink) {
    const unsigned lane = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long n = (hi - lo) / 1024;            // rows of 1 KiB
    for (unsigned long long r0 = 0; r0 < n; r0 += 16) {
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned long long r = r0 + i;
            if (r < n) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(buf + lo + (REV ? n - 1 - r : r) * 1024) + lane);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) if (r0 + i < n) { acc.x ^= v[i].x; acc.y += v[i].y; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}
// the shipped checker's shape: 8-byte lanes, passes of 31 rows in batches of 16 + 15 (the second in flight while the first is compared),
// a popcount of the XOR with a register window per row; LW 16: the same with 16-byte lanes (batches of 8 rows)
template <int LW>
__global__ void __launch_bounds__(64, 2) rd_prbs(const char *buf, unsigned long long total, unsigned long long chunk, unsigned long long *sink) {
    constexpr int K = 31, NW = LW / 4, DB = LW == 8 ? 16 : 8;
    const unsigned lane = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t V[K][NW];
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int w = 0; w < NW; w++) V[i][w] = lane * 2654435761u + i * 40503u + w;
    constexpr unsigned ROW = 64 * LW;
    unsigned errs = 0;
    const long long npass = (long long)((hi - lo) / ((unsigned long long)K * ROW));
    for (long long ps = npass - 1; ps >= 0; ps--) {
        const char *rowp = buf + lo + (unsigned long long)ps * K * ROW + lane * LW;
#pragma unroll
        for (int i = K - 1; i >= 0; i--)
#pragma unroll
            for (int w = 0; w < NW; w++) asm("v_xor_b32 %0, %0, %1" : "+v"(V[i][w]) : "v"(V[(i + K - 3) % K][w]));
        uint32_t D[2][DB][NW];
        auto load = [&](int b, int slot) {
#pragma unroll
            for (int i = 0; i < DB; i++) {
                const int r = b * DB + i;
                if (r < K) {
                    if (LW == 8) { const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(rowp + (unsigned long long)r * ROW)); D[slot][i][0] = v.x; D[slot][i][1 % NW] = v.y; }
                    else { const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rowp + (unsigned long long)r * ROW)); D[slot][i][0] = v.x; D[slot][i][1 % NW] = v.y; D[slot][i][2 % NW] = v.z; D[slot][i][3 % NW] = v.w; }
                }
            }
        };
        constexpr int NB = (K + DB - 1) / DB;
        load(0, 0);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (b + 1 < NB) load(b + 1, (b + 1) & 1);
#pragma unroll
            for (int i = 0; i < DB; i++) {
                const int r = b * DB + i;
                if (r < K)
#pragma unroll
                    for (int w = 0; w < NW; w++) errs += __builtin_popcount(D[b & 1][i][w] ^ V[r][w]);
            }
        }
    }
    if (errs == 0x12345678u) sink[0] = 1;
}
template <int LW> void run_rd_prbs(const char *name, char *buf, unsigned long long total, unsigned long long *sink) {
    const unsigned long long unit = 31ull * 64 * LW;
    const unsigned long long chunk = ((total + 1023) / 1024 + unit - 1) / unit * unit;
    const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
    const unsigned long long wchunk = ((total + 1023) / 1024 + 16383) / 16384 * 16384;
    const unsigned wgrid = (unsigned)((total + wchunk - 1) / wchunk);
    hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
    float w = 0, rb = 0, rc = 0;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(e[0]);
        hipLaunchKernelGGL((wr<0, false>), dim3(wgrid), dim3(64), 0, 0, buf, total, wchunk);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL(rd_prbs<LW>, dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
        hipEventRecord(e[2]);
        hipLaunchKernelGGL(rd_prbs<LW>, dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
        hipEventRecord(e[3]); hipEventSynchronize(e[3]);
        float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]);
        if (rep >= 2) { w += a / 5; rb += b / 5; rc += c / 5; }
    }
    printf("%-44s write %.4f ms | checker-shaped read behind it %.4f ms (%.2f TB/s) | again (clean) %.4f ms (%.2f)\n", name, w, rb, total / rb / 1e9, rc, total / rc / 1e9);
}
template <int FL, bool RND = false> void run(const char *name, char *buf, unsigned long long total, unsigned long long *sink) {
    const unsigned long long chunk = ((total + 1023) / 1024 + 16383) / 16384 * 16384;
    const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
    hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
    float w = 0, ra = 0, rr = 0, rc = 0;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(e[0]);
        hipLaunchKernelGGL((wr<FL, RND>), dim3(grid), dim3(64), 0, 0, buf, total, chunk);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
        hipEventRecord(e[2]);
        hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
        hipEventRecord(e[3]); hipEventSynchronize(e[3]);
        float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]);
        hipEventRecord(e[0]);
        hipLaunchKernelGGL((wr<FL, RND>), dim3(grid), dim3(64), 0, 0, buf, total, chunk);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
        hipEventRecord(e[2]); hipEventSynchronize(e[2]);
        float d; hipEventElapsedTime(&d, e[1], e[2]);
        if (rep >= 2) { w += a / 5; rr += b / 5; rc += c / 5; ra += d / 5; }
    }
    printf("%-28s write %.4f ms (%.2f TB/s) | read behind it, regions from their ends %.4f ms (%.2f), ascending %.4f ms (%.2f) | read again (clean) %.4f ms (%.2f) | write + read %.4f ms = %.3f of 8 TB/s\n",
           name, w, total / w / 1e9, rr, total / rr / 1e9, ra, total / ra / 1e9, rc, total / rc / 1e9, w + rr, 2 * total / (w + rr) / 1e9 / 8);
}
int main() {
    const unsigned long long total = 1250000000ull / 16384 * 16384;
    char *buf; unsigned long long *sink;
    hipMalloc(&buf, total + 65536); hipMalloc(&sink, 8);
    hipMemset(buf, 1, total);
    for (int pass = 0; pass < 2; pass++) {
        run<0>("store", buf, total, sink);
        run_rd_prbs<8>("checker-shaped reader, 8 B lanes", buf, total, sink);
        run_rd_prbs<16>("checker-shaped reader, 16 B lanes", buf, total, sink);
        run<1>("store nt", buf, total, sink);
    }
    return 0;
}
