// Round 5: what a streaming WRITER / READER of 1.25 GB loses against a classic fill when its waves are long-lived (the PRBS
// generator's and checker's structure) -- and which knob gets it back.  One wave per block, rows of 64 lanes x LW bytes.
//   chunked: grid = total / chunk blocks, wave b streams chunk b from start to end and exits   (chunk = 16 KiB ... the whole region of a
//            1024-wave partition); the hardware's dispatcher refills a SIMD when a wave ends
//   throttle: s_waitcnt vmcnt(T) behind every store / in front of every load batch: at most T + 1 in flight per wave
//   waves per SIMD: __launch_bounds__ + LDS padding is not used; occupancy is whatever 64-thread blocks reach (up to 8 per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LW> struct LT;
template <> struct LT<8> { typedef u32x2 t; };
template <> struct LT<16> { typedef u32x4 t; };
template <int T> __device__ __forceinline__ void throttle() {
    if (T == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (T == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (T == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (T == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (T == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}
// MODE 0 store, 1 store nt, 2 load nt, 3 load
template <int LW, int MODE, int T>
__global__ void __launch_bounds__(64) stream_k(char *buf, unsigned long long total, unsigned long long chunk, unsigned long long *sink) {
    typedef typename LT<LW>::t v_t;
    const unsigned lane = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    v_t acc;
    acc.x = lane; acc.y = blockIdx.x;
    constexpr unsigned ROW = 64 * LW;
    for (unsigned long long o = lo; o < hi; o += 8 * ROW) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned long long a = o + (unsigned long long)i * ROW;
            if (a >= hi) break;
            v_t *p = reinterpret_cast<v_t *>(buf + a) + lane;
            if (MODE == 0) { acc.x += 0x9e3779b9u; *p = acc; }
            else if (MODE == 1) { acc.x += 0x9e3779b9u; __builtin_nontemporal_store(acc, p); }
            else if (MODE == 2) { const v_t v = __builtin_nontemporal_load(p); acc.x ^= v.x; acc.y += v.y; }
            else { const v_t v = *p; acc.x ^= v.x; acc.y += v.y; }
            if (T >= 0 && MODE < 2) throttle<T>();
        }
    }
    if (MODE >= 2 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}
template <int LW, int MODE, int T> float run(char *buf, unsigned long long total, unsigned long long chunk, unsigned long long *sink) {
    const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((stream_k<LW, MODE, T>), dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
    hipEventRecord(e0);
    const int N = 8;
    for (int rep = 0; rep < N; rep++) hipLaunchKernelGGL((stream_k<LW, MODE, T>), dim3(grid), dim3(64), 0, 0, buf, total, chunk, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / N;
}
template <int LW, int MODE> void sweep(const char *name, char *buf, unsigned long long total, unsigned long long *sink) {
    const unsigned long long region = ((total + 1023) / 1024 + 4095) / 4096 * 4096;        // a 1024-wave partition
    const unsigned long long chunks[] = {16ull << 10, 64ull << 10, 256ull << 10, 512ull << 10, region, 2 * region, 4 * region};
    for (unsigned long long c : chunks) {
        const float ms = run<LW, MODE, -1>(buf, total, c, sink);
        printf("%-18s %2d B/lane chunk %8llu B (grid %6llu): %.4f ms  %.2f TB/s\n", name, LW, c, (total + c - 1) / c, ms, total / ms / 1e9);
    }
}
template <int LW, int MODE> void sweep_throttle(const char *name, char *buf, unsigned long long total, unsigned long long *sink) {
    const unsigned long long region = ((total + 1023) / 1024 + 4095) / 4096 * 4096;
    for (unsigned long long c : {region, region / 2, region / 4}) {
        printf("%-18s %2d B/lane chunk %8llu B, in flight <= 1/2/4/8/16: %.4f %.4f %.4f %.4f %.4f ms\n", name, LW, c,
               run<LW, MODE, 0>(buf, total, c, sink), run<LW, MODE, 1>(buf, total, c, sink), run<LW, MODE, 3>(buf, total, c, sink),
               run<LW, MODE, 7>(buf, total, c, sink), run<LW, MODE, 15>(buf, total, c, sink));
    }
}
int main() {
    const unsigned long long total = 1250000000ull / 4096 * 4096;
    char *buf; unsigned long long *sink;
    hipMalloc(&buf, total + 4096); hipMalloc(&sink, 8);
    hipMemset(buf, 1, total);
    sweep<8, 0>("store", buf, total, sink);
    sweep<16, 0>("store", buf, total, sink);
    sweep<16, 1>("store nt", buf, total, sink);
    sweep<8, 2>("load nt", buf, total, sink);
    sweep<16, 2>("load nt", buf, total, sink);
    sweep<16, 3>("load", buf, total, sink);
    sweep_throttle<8, 0>("store", buf, total, sink);
    sweep_throttle<16, 0>("store", buf, total, sink);
    // loopback order: a store pass, then a load pass over the same bytes
    {
        const unsigned long long region = ((total + 1023) / 1024 + 4095) / 4096 * 4096;
        for (unsigned long long c : {region, 256ull << 10, 64ull << 10}) {
            hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
            float f = 0, l = 0;
            const unsigned grid = (unsigned)((total + c - 1) / c);
            for (int rep = 0; rep < 6; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((stream_k<16, 0, -1>), dim3(grid), dim3(64), 0, 0, buf, total, c, sink);
                hipEventRecord(e1);
                hipLaunchKernelGGL((stream_k<16, 2, -1>), dim3(grid), dim3(64), 0, 0, buf, total, c, sink);
                hipEventRecord(e2); hipEventSynchronize(e2);
                float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
                if (rep >= 2) { f += a / 4; l += b / 4; }
            }
            printf("loopback 16 B/lane chunk %8llu B: store %.4f ms + load nt behind it %.4f ms = %.4f ms (%.3f of 8 TB/s)\n", c, f, l, f + l, 2 * total / (f + l) / 1e9 / 8);
        }
    }
    return 0;
}
