// Round 2 microbenchmark: moving 16 registers between the AGPR and VGPR files with ONE integer MFMA
// (D = 0 * 0 + C, exact) instead of 16 v_accvgpr moves -- what does it cost in a VALU-bound stream at one wave
// per SIMD?   hipcc -O3 --offload-arch=gfx950 ubench_mfma_move.hip -o ubench_mfma_move
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define VALU8(x)                                                                         \
    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %2, %0 bitop3:0x96\n\t" \
                 "v_bitop3_b32 %2, %2, %0, %1 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\t" \
                 "v_bitop3_b32 %1, %1, %2, %0 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %0, %1 bitop3:0x96\n\t" \
                 "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %2, %0 bitop3:0x96"     \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]))

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(uint32_t *out, int iters, long long *cycles) {
    uint32_t x[3] = {threadIdx.x * 2654435761u, 12345u + blockIdx.x, 999u};
    v16i blkA, blkV;
    for (int i = 0; i < 16; i++) blkV[i] = (int)(threadIdx.x * 16 + i) * 0x01010101 - 77;
    const v4i zero = {0, 0, 0, 0};
    // park once
    asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %1, %2\n\ts_nop 15\n\ts_nop 3" : "=a"(blkA) : "v"(zero), "v"(blkV));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        for (int r = 0; r < 8; r++) VALU8(x);               // 64 VALU
        if (MODE == 1) {                                    // restore by MFMA, park again by MFMA
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %1, %2\n\ts_nop 11" : "=v"(blkV) : "v"(zero), "a"(blkA));
            blkV[0] ^= x[0] & 1;                            // (a use, so that nothing is dead)
            asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %1, %2" : "=a"(blkA) : "v"(zero), "v"(blkV));
        } else if (MODE == 2) {                             // the same by 16 + 16 accvgpr moves
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(blkV[i]) : "a"(blkA[i]));
            blkV[0] ^= x[0] & 1;
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(blkA[i]) : "v"(blkV[i]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_mfma_i32_32x32x32_i8 %0, %1, %1, %2\n\ts_nop 15\n\ts_nop 3" : "=v"(blkV) : "v"(zero), "a"(blkA));
    uint32_t acc = x[0] ^ x[1] ^ x[2];
    for (int i = 0; i < 16; i++) out[(blockIdx.x * 64 + threadIdx.x) * 16 + i] = (uint32_t)blkV[i];
    out[(size_t)gridDim.x * 64 * 16 + blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    const int blocks = 1024, iters = 20000;
    uint32_t *d; long long *dc;
    hipMalloc(&d, (size_t)blocks * 64 * 17 * 4); hipMalloc(&dc, 8);
    std::vector<uint32_t> h((size_t)blocks * 64 * 16);
    for (int mode = 0; mode < 3; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, dc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, dc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters, dc);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        // expected: value i of lane l, with bit 0 of element 0 toggled some number of times (modes 1, 2)
        int bad = 0;
        for (int b = 0; b < blocks; b += 97)
            for (int l = 0; l < 64; l++)
                for (int i = 1; i < 16; i++)
                    if (h[((size_t)b * 64 + l) * 16 + i] != (uint32_t)((l * 16 + i) * 0x01010101 - 77)) bad++;
        printf("mode %d (%s): %.3f ms, %.1f cycles per iteration (s_memtime units), per-iteration extra vs 64 VALU; mismatches %d\n", mode,
               mode == 0 ? "64 VALU only" : mode == 1 ? "+ MFMA restore (s_nop 11) + MFMA park" : "+ 16 accvgpr_read + 16 accvgpr_write",
               ms, (double)cyc / iters, bad);
    }
    return 0;
}
