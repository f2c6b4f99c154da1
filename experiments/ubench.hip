// Round-1 design microbenchmarks (not product code): VALU issue rate for 3-input
// bit ops at 1/2/4 waves per SIMD, and HBM write efficiency of partial-line
// "segment front" store patterns.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

extern __shared__ unsigned dyn_lds[];

template<int NREG>
__global__ void __launch_bounds__(512) valu_rate(unsigned* out, int iters, unsigned seed){
  unsigned r[NREG];
#pragma unroll
  for(int i=0;i<NREG;i++) r[i]=seed*(i+1)+threadIdx.x*2654435761u;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NREG;i++){
      r[i]=__builtin_amdgcn_bitop3_b32(r[i], r[(i+7)%NREG], r[(i+13)%NREG], 0x96);
    }
#pragma unroll
    for(int i=0;i<NREG;i++){
      r[i]=__builtin_amdgcn_bitop3_b32(r[i], r[(i+5)%NREG], r[(i+11)%NREG], 0xE8);
    }
  }
  unsigned a=0;
#pragma unroll
  for(int i=0;i<NREG;i++) a^=r[i];
  if(a==0x12345678u) out[0]=a + dyn_lds[0];
}

__global__ void valu_rate_xor2(unsigned* out, int iters, unsigned seed){
  constexpr int NREG=32;
  unsigned r[NREG];
#pragma unroll
  for(int i=0;i<NREG;i++) r[i]=seed*(i+1)+threadIdx.x*2654435761u;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int k=0;k<2;k++)
#pragma unroll
    for(int i=0;i<NREG;i++){
      r[i]=r[i]^(r[(i+7)%NREG]);
    }
  }
  unsigned a=0;
#pragma unroll
  for(int i=0;i<NREG;i++) a+=r[i];
  if(a==0x12345678u) out[0]=a + dyn_lds[0];
}

__global__ void bitop_check(unsigned* out){
  unsigned a=0xF0F0F0F0u ^ (threadIdx.x>>7), b=0xCCCCCCCCu^ (threadIdx.x>>7), c=0xAAAAAAAAu^ (threadIdx.x>>7);
  out[0]=__builtin_amdgcn_bitop3_b32(a,b,c,0xCA); // expect bfi-like: a?b:c if tt=f(0xF0,0xCC,0xAA)
  out[1]=(a&b)|(~a&c);
  out[2]=__builtin_amdgcn_bitop3_b32(a,b,c,0x96);
  out[3]=a^b^c;
  out[4]=__builtin_amdgcn_bitop3_b32(a,b,c,0xE8);
  out[5]=(a&b)|(a&c)|(b&c);
}

// ---------------- write patterns ----------------
// plain streaming fill, 16 B per lane, grid-stride
__global__ void fill_stream(uint4* dst, size_t n16){
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x*blockDim.x;
  uint4 v = make_uint4(i,1,2,3);
  for(; i<n16; i+=stride) dst[i]=v;
}
// segment fronts: every lane owns 32 segments of L bytes; per round writes W bytes
// (W/16 consecutive 16-B stores) to each of its segments. ORDER 0: g=(lane_global*32+j); ORDER 1: g=(wave*32+j)*64+lane
template<int W, int ORDER, bool NT>
__global__ void __launch_bounds__(256) fill_fronts(char* dst, int L, size_t G, int delay){
  size_t lane_global = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t wave = lane_global>>6; int lane = lane_global&63;
  int rounds = L / W;
  unsigned x = lane_global;
  for(int r=0;r<rounds;r++){
    // synthetic compute delay
    for(int d=0; d<delay; d++){ x = x*1664525u + 1013904223u; }
#pragma unroll 8
    for(int j=0;j<32;j++){
      size_t g = ORDER==0 ? (lane_global*32 + j) : ((wave*32 + j)*64 + lane);
      if(g<G){
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* p = (v4u*)(dst + g*(size_t)L + (size_t)r*W);
#pragma unroll
        for(int q=0;q<W/16;q++){
          v4u v = {x,(unsigned)j,(unsigned)r,(unsigned)q};
          if(NT) __builtin_nontemporal_store(v, p+q); else p[q]=v;
        }
      }
    }
  }
}

static double time_kernel(std::function<void()> f, int reps=5){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  double best=1e30;
  for(int i=0;i<reps;i++){
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best) best=ms;
  }
  return best;
}

int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  unsigned* d; CK(hipMalloc(&d, 1<<20));
  // bitop check
  bitop_check<<<1,64>>>(d); unsigned h[6]; CK(hipMemcpy(h,d,24,hipMemcpyDeviceToHost));
  printf("bitop3 0xCA=%08x (a?b:c=%08x)  0x96=%08x (xor3=%08x)  0xE8=%08x (maj=%08x)\n",h[0],h[1],h[2],h[3],h[4],h[5]);

  int nCU = prop.multiProcessorCount;
  // VALU rate: blocks of 256 threads (one wave per SIMD). LDS sizing forces blocks/CU.
  {
    int iters=20000;
    struct Cfg{int threads; int blocks_per_cu; size_t lds;};
    Cfg cfgs[]={{256,1,120*1024},{256,2,70*1024},{256,4,36*1024},{512,1,120*1024},{512,2,70*1024},{256,8,16*1024}};
    CK(hipFuncSetAttribute((const void*)valu_rate<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    CK(hipFuncSetAttribute((const void*)valu_rate_xor2, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    for(auto c: cfgs){
      int grid = nCU*c.blocks_per_cu;
      double ms = time_kernel([&]{ hipLaunchKernelGGL(valu_rate<32>, dim3(grid), dim3(c.threads), c.lds, 0, d, iters, 12345u); });
      double ops = (double)grid*c.threads*iters*64.0; // lane-ops
      printf("valu bitop3: threads %d blocks/CU %d -> %.3f ms  %.2f Tlane-op/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", c.threads, c.blocks_per_cu, ms, ops/ms/1e9,
             2.4e9*(ms/1e3) / ((double)iters*64.0*c.blocks_per_cu*c.threads/256.0));
      ms = time_kernel([&]{ hipLaunchKernelGGL(valu_rate_xor2, dim3(grid), dim3(c.threads), c.lds, 0, d, iters, 12345u); });
      printf("valu xor2  : threads %d blocks/CU %d -> %.3f ms  %.2f Tlane-op/s\n", c.threads, c.blocks_per_cu, ms, ops/ms/1e9);
    }
  }
  // write patterns
  {
    size_t N = (size_t)1<<30; // 1 GiB
    char* buf; CK(hipMalloc(&buf, N + (1<<20)));
    CK(hipMemset(buf,0,N));
    double ms = time_kernel([&]{ hipLaunchKernelGGL(fill_stream, dim3(nCU*8), dim3(256), 0,0,(uint4*)buf, N/16); });
    printf("fill_stream 1GiB: %.3f ms %.1f GB/s\n", ms, N/ms/1e6);
    int Ls[]={512, 1024, 4096};
    for(int L: Ls){
      size_t G = N / L;
      size_t lanes = (G+31)/32; int grid = (int)((lanes+255)/256);
      for(int delay: {0, 2000}){
#define RUN(W,ORDER,NT) { double ms = time_kernel([&]{ hipLaunchKernelGGL((fill_fronts<W,ORDER,NT>), dim3(grid), dim3(256),0,0,buf,L,G,delay); },3); \
        printf("fronts L=%d W=%d order=%d nt=%d delay=%d grid=%d: %.3f ms %.1f GB/s\n", L,W,ORDER,(int)NT,delay,grid,ms,N/ms/1e6); }
        RUN(16,0,false) RUN(16,1,false) RUN(16,0,true) RUN(16,1,true)
        RUN(32,0,false) RUN(32,1,false) RUN(32,1,true)
        RUN(64,0,false) RUN(64,1,false) RUN(64,1,true)
        RUN(128,0,false) RUN(128,1,false)
      }
    }
  }
  return 0;
}
