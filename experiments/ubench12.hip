// Round 5: what one vector instruction costs a lone wave per SIMD, by opcode -- the shaping mover's candidates (V_PK_MAD_U16, V_PK_ASHRREV_I16,
// V_PERM_B32, V_MAD_U32_U24, V_MOV_B32), independent and as a dependent chain.  hipcc --offload-arch=gfx950 -O3 ubench12.hip -o ubench12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP, bool DEP>
__global__ void __launch_bounds__(64) k(unsigned long long *out, unsigned seed) {
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 * 3u, a3 = a0 + 77u, b = seed | 5u, c = seed * 7u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
        if (OP == 0) { if (DEP) { REP64(asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
                       else { REP64(asm volatile("v_pk_mad_u16 %0, %4, %5, %0\n\tv_pk_mad_u16 %1, %4, %5, %1\n\tv_pk_mad_u16 %2, %4, %5, %2\n\tv_pk_mad_u16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) } }
        if (OP == 1) { if (DEP) { REP64(asm volatile("v_pk_ashrrev_i16 %0, 4, %0" : "+v"(a0));) }
                       else { REP64(asm volatile("v_pk_ashrrev_i16 %0, 4, %0\n\tv_pk_ashrrev_i16 %1, 4, %1\n\tv_pk_ashrrev_i16 %2, 4, %2\n\tv_pk_ashrrev_i16 %3, 4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) } }
        if (OP == 2) { if (DEP) { REP64(asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a0) : "v"(b), "v"(c));) }
                       else { REP64(asm volatile("v_perm_b32 %0, %4, %0, %5\n\tv_perm_b32 %1, %4, %1, %5\n\tv_perm_b32 %2, %4, %2, %5\n\tv_perm_b32 %3, %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) } }
        if (OP == 3) { if (DEP) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
                       else { REP64(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n\tv_mad_u32_u24 %1, %4, %5, %1\n\tv_mad_u32_u24 %2, %4, %5, %2\n\tv_mad_u32_u24 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) } }
        if (OP == 4) { if (DEP) { REP64(asm volatile("v_mov_b32 %0, %0" : "+v"(a0));) }
                       else { REP64(asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) } }
        if (OP == 5) { if (DEP) { REP64(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a0) : "v"(b));) }
                       else { REP64(asm volatile("v_pk_add_u16 %0, %0, %4\n\tv_pk_add_u16 %1, %1, %4\n\tv_pk_add_u16 %2, %2, %4\n\tv_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) } }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) out[0] = 0;
}
template <int OP, bool DEP> void run(const char *name, unsigned long long *d, int blocks) {
    hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    const double n = 64.0 * 64 * (DEP ? 1 : 4);
    printf("%-18s %s, %4d waves: %.2f cycles per instruction\n", name, DEP ? "dependent  " : "independent", blocks, s / blocks / n);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 8 * 2048);
    for (int blocks : {1, 1024}) {
        run<0, false>("v_pk_mad_u16", d, blocks); run<0, true>("v_pk_mad_u16", d, blocks);
        run<1, false>("v_pk_ashrrev_i16", d, blocks); run<1, true>("v_pk_ashrrev_i16", d, blocks);
        run<5, false>("v_pk_add_u16", d, blocks); run<5, true>("v_pk_add_u16", d, blocks);
        run<2, false>("v_perm_b32", d, blocks); run<2, true>("v_perm_b32", d, blocks);
        run<3, false>("v_mad_u32_u24", d, blocks); run<3, true>("v_mad_u32_u24", d, blocks);
        run<4, false>("v_mov_b32", d, blocks); run<4, true>("v_mov_b32", d, blocks);
    }
    return 0;
}
