// Round 5: can the hardware's own dispatcher do the balancing?  A static partition into MANY more regions than there are resident
// waves -- each 64-thread block streams its region and exits, the dispatcher starts the next block on the freed slot -- with the
// number of RESIDENT blocks per CU capped by a dynamic LDS allocation (160 KiB / lds per block).  A `boot` of ~6 us of ALU work in
// front of every region stands for the PRBS generator's window bootstrap.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k(char *buf, unsigned long long total, unsigned long long chunk, unsigned boot_iters, unsigned *sink) {
    extern __shared__ uint32_t pad[];
    const unsigned lane = threadIdx.x;
    u32x2 acc; acc.x = lane; acc.y = blockIdx.x;
    // the bootstrap: dependent integer work, no memory traffic
    for (unsigned i = 0; i < boot_iters; i++) acc.x = acc.x * 1664525u + 1013904223u + (acc.y ^= acc.x >> 7);
    if (acc.x == 0x12345u) pad[lane] = acc.y;        // (keeps the allocation and the loop alive)
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    constexpr unsigned ROW = 64 * 8;
    for (unsigned long long o = lo; o < hi; o += 31 * ROW) {
#pragma unroll
        for (int i = 0; i < 31; i++) {
            const unsigned long long a = o + (unsigned long long)i * ROW;
            if (a >= hi) break;
            acc.x += 0x9e3779b9u;
            *(reinterpret_cast<u32x2 *>(buf + a) + lane) = acc;
        }
    }
    if (acc.x == 0x54321u) sink[0] = pad[lane];
}
int main() {
    const unsigned long long total = 1250000000ull / 4096 * 4096;
    char *buf; unsigned *sink;
    hipMalloc(&buf, total + 65536); hipMalloc(&sink, 8);
    hipMemset(buf, 1, total);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned boot : {0u, 3000u}) {                       // 3000 dependent iterations of ~4 instructions ~ 6 us for a lone wave
        for (unsigned per_cu : {1u, 2u, 3u, 4u, 6u}) {
            const unsigned lds = (160u * 1024u / per_cu) & ~1023u;
            for (unsigned regions : {256u * per_cu, 1024u, 2048u, 4096u, 8192u, 16384u}) {
                if (regions < 256 * per_cu) continue;
                const unsigned long long chunk = ((total + regions - 1) / regions + 15871) / 15872 * 15872;      // whole passes of 31 rows
                const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
                float sum = 0;
                for (int rep = 0; rep < 7; rep++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds > 65536 ? lds - 1024 : lds, 0, buf, total, chunk, boot, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 2) sum += ms / 5;
                }
                printf("boot %4u iters, %u resident per CU (%3u KiB LDS each), %5u regions of %8llu B: %.4f ms  %.2f TB/s\n", boot, per_cu, lds / 1024, grid, chunk, sum,
                       total / sum / 1e9);
            }
        }
    }
    return 0;
}
