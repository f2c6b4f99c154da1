// Streaming pattern microbenchmark for the PRBS kernels: W waves (4 per CU), 8-byte lanes, rows of 512 bytes.
//   pattern 0: wave w owns the contiguous region [w*R, (w+1)*R) rows (what prbs_stream_kernel does)
//   pattern 1: wave w owns rows w, w+W, w+2W, ... (all waves write inside one sliding window of W rows)
// store and load (with / without the non-temporal hint), 1.25 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int PAT, int MODE>   // MODE 0 store, 1 load, 2 load nt, 3 store nt
__global__ void __launch_bounds__(64, 2) k(u32x2 *buf, unsigned long long rows, unsigned long long rpw, unsigned long long *sink) {
    const unsigned lane = threadIdx.x;
    const unsigned long long w = blockIdx.x, W = gridDim.x;
    u32x2 acc = {lane, (unsigned)w};
    for (unsigned long long j = 0; j < rpw; j += 16) {
        u32x2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned long long jj = j + i;
            const unsigned long long row = PAT == 0 ? w * rpw + jj : jj * W + w;
            if (row >= rows) continue;
            u32x2 *p = buf + row * 64 + lane;
            if (MODE == 0) { acc.x += 0x9e3779b9u; *p = acc; }
            else if (MODE == 3) { acc.x += 0x9e3779b9u; __builtin_nontemporal_store(acc, p); }
            else if (MODE == 1) v[i] = *p;
            else v[i] = __builtin_nontemporal_load(p);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) { const unsigned long long jj = j + i; const unsigned long long row = PAT == 0 ? w * rpw + jj : jj * W + w; if (row < rows) { acc.x ^= v[i].x; acc.y += v[i].y; } }
        }
    }
    if ((MODE == 1 || MODE == 2) && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}
template <int PAT, int MODE> void run(const char *name, u32x2 *buf, unsigned long long rows, unsigned long long *sink, int waves) {
    const unsigned long long rpw = (rows + waves - 1) / waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<PAT, MODE>), dim3(waves), dim3(64), 0, 0, buf, rows, rpw, sink);
    hipEventRecord(e0);
    const int N = 10;
    for (int rep = 0; rep < N; rep++) hipLaunchKernelGGL((k<PAT, MODE>), dim3(waves), dim3(64), 0, 0, buf, rows, rpw, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("%-34s waves %5d: %.4f ms  %.2f TB/s\n", name, waves, ms, rows * 512.0 / ms / 1e9);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// classic fill: blocks of 256 threads, 16 bytes per thread, UNR chunks per thread, chunk c of block b at (b*UNR + c)*256 + tid
template <int UNR, bool NT>
__global__ void __launch_bounds__(256) fillk(u32x4 *buf, unsigned long long n16) {
    const unsigned long long base = (unsigned long long)blockIdx.x * UNR * 256 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
    for (int c = 0; c < UNR; c++) {
        const unsigned long long i = base + (unsigned long long)c * 256;
        if (i < n16) { if (NT) __builtin_nontemporal_store(v, buf + i); else buf[i] = v; }
    }
}
// the same with DELAY dependent integer ops between a thread's stores, and a variant whose chunks lie a whole grid apart
template <int UNR, int DELAY, bool FAR>
__global__ void __launch_bounds__(256) fillk2(u32x4 *buf, unsigned long long n16) {
    const unsigned long long per = FAR ? (unsigned long long)gridDim.x * 256 : 256;
    const unsigned long long base = FAR ? (unsigned long long)blockIdx.x * 256 + threadIdx.x : (unsigned long long)blockIdx.x * UNR * 256 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
    for (int c = 0; c < UNR; c++) {
        const unsigned long long i = base + (unsigned long long)c * per;
#pragma unroll
        for (int d = 0; d < DELAY; d++) v.x = v.x * 1664525u + 1013904223u;
        if (i < n16) buf[i] = v;
    }
}
template <int UNR, int DELAY, bool FAR> void runfill2(const char *name, u32x4 *buf, unsigned long long n16) {
    const unsigned grid = (unsigned)((n16 + 256ull * UNR - 1) / (256ull * UNR));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((fillk2<UNR, DELAY, FAR>), dim3(grid), dim3(256), 0, 0, buf, n16);
    hipEventRecord(e0);
    const int N = 10;
    for (int rep = 0; rep < N; rep++) hipLaunchKernelGGL((fillk2<UNR, DELAY, FAR>), dim3(grid), dim3(256), 0, 0, buf, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("%-34s grid %7u: %.4f ms  %.2f TB/s\n", name, grid, ms, n16 * 16.0 / ms / 1e9);
}
template <int UNR, bool NT> void runfill(const char *name, u32x4 *buf, unsigned long long n16) {
    const unsigned grid = (unsigned)((n16 + 256ull * UNR - 1) / (256ull * UNR));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((fillk<UNR, NT>), dim3(grid), dim3(256), 0, 0, buf, n16);
    hipEventRecord(e0);
    const int N = 10;
    for (int rep = 0; rep < N; rep++) hipLaunchKernelGGL((fillk<UNR, NT>), dim3(grid), dim3(256), 0, 0, buf, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("%-34s grid %7u: %.4f ms  %.2f TB/s\n", name, grid, ms, n16 * 16.0 / ms / 1e9);
}
int main() {
    const unsigned long long rows = 1250000000ull / 512;
    u32x2 *buf; unsigned long long *sink;
    hipMalloc(&buf, rows * 512 + 4096); hipMalloc(&sink, 8);
    hipMemset(buf, 1, rows * 512);
    runfill<1, false>("classic fill 16 B/thread", (u32x4 *)buf, rows * 32);
    runfill<4, false>("classic fill 4 x 16 B/thread", (u32x4 *)buf, rows * 32);
    runfill<16, false>("classic fill 16 x 16 B/thread", (u32x4 *)buf, rows * 32);
    runfill<4, true>("classic fill nt 4 x 16 B/thread", (u32x4 *)buf, rows * 32);
    runfill2<4, 40, false>("4 stores, 40 ops between", (u32x4 *)buf, rows * 32);
    runfill2<16, 40, false>("16 stores, 40 ops between", (u32x4 *)buf, rows * 32);
    runfill2<4, 0, true>("4 stores, a grid apart", (u32x4 *)buf, rows * 32);
    runfill2<16, 0, true>("16 stores, a grid apart", (u32x4 *)buf, rows * 32);
    runfill2<64, 0, true>("64 stores, a grid apart", (u32x4 *)buf, rows * 32);
    runfill2<64, 20, true>("64 stores apart, 20 ops between", (u32x4 *)buf, rows * 32);
    for (int waves : {1024}) {
        run<0, 0>("store contiguous regions", buf, rows, sink, waves);
        run<1, 0>("store interleaved rows", buf, rows, sink, waves);
        run<0, 3>("store nt contiguous regions", buf, rows, sink, waves);
        run<1, 3>("store nt interleaved rows", buf, rows, sink, waves);
        run<0, 1>("load contiguous regions", buf, rows, sink, waves);
        run<1, 1>("load interleaved rows", buf, rows, sink, waves);
        run<0, 2>("load nt contiguous regions", buf, rows, sink, waves);
        run<1, 2>("load nt interleaved rows", buf, rows, sink, waves);
    }
    return 0;
}
