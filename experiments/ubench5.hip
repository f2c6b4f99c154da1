// Round 3 microbenchmark (not product code): what does a GUEST wave's instruction cost the HOST wave of its SIMD?
// Host: one wave per SIMD (40 KiB of LDS per block: four blocks per CU), a long stream of independent V_BITOP3 at
// s_setprio 3, timed per wave with s_memtime.  Guest: one block of GW waves per CU (a second stream), which repeats
// { K instructions of one kind ; s_sleep S } -- kind: 0 V_BITOP3, 1 V_PERM_B32, 2 v_mov, 3 ds_write_b32, 4 global load 16 B,
// 5 s_nop (scalar only), 6 v_mad_u64_u32.  Printed: host cycles per instruction alone and beside the guest, the guest's
// instruction count per wave, and the host cycles lost per guest instruction on the SIMDs that hosted a guest.
// Build: hipcc --offload-arch=gfx950 -O3 ubench5.hip -o ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void __launch_bounds__(64) host_kernel(unsigned long long* stamps, unsigned* sink, int iters){
  __shared__ unsigned pad[10*1024];
  unsigned r[64];
#pragma unroll
  for(int i=0;i<64;i++) r[i]=threadIdx.x*2654435761u+i;
  __builtin_amdgcn_s_setprio(3);
  const unsigned long long r0=__builtin_amdgcn_s_memrealtime();
  const unsigned long long t0=__builtin_amdgcn_s_memtime();
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int k=0;k<4;k++)
#pragma unroll
    for(int i=0;i<64;i++) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(r[i]) : "v"(r[i]), "v"(r[(i+7)&63]), "v"(r[(i+13)&63]));
  }
  const unsigned long long t1=__builtin_amdgcn_s_memtime();
  unsigned a=0;
#pragma unroll
  for(int i=0;i<64;i++) a^=r[i];
  if(a==0x12345678u) sink[0]=a+pad[threadIdx.x];
  if(threadIdx.x==0){
    unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[4*blockIdx.x]=t1-t0; stamps[4*blockIdx.x+1]=((hwid>>4)&0x3ff) | ((unsigned long long)(xcc&15)<<10); stamps[4*blockIdx.x+2]=r0; stamps[4*blockIdx.x+3]=__builtin_amdgcn_s_memrealtime();
  }
}

template<int KIND>
__global__ void __launch_bounds__(256) guest_kernel(unsigned long long* gstamps, unsigned* sink, const uint4* src, int reps, int K, int S){
  __shared__ unsigned lds[4096];
  unsigned r[16];
#pragma unroll
  for(int i=0;i<16;i++) r[i]=threadIdx.x*40503u+i;
  uint4 acc=make_uint4(0,0,0,0);
  unsigned long long m=0;
  const unsigned long long gr0=__builtin_amdgcn_s_memrealtime();
  const uint4* p=src+(blockIdx.x*256+threadIdx.x);
  for(int rep=0; rep<reps; ++rep){
    for(int k=0;k<K;k+=16){
#pragma unroll
      for(int i=0;i<16;i++){
        if(KIND==0) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(r[i]) : "v"(r[i]), "v"(r[(i+5)&15]), "v"(r[(i+3)&15]));
        if(KIND==1) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(r[(i+5)&15]), "v"(r[(i+3)&15]));
        if(KIND==2) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i+1)&15]));
        if(KIND==3) lds[(threadIdx.x + 64*i) & 4095]=r[i];
        if(KIND==4) { uint4 v=p[(size_t)(rep*K+k+i)*65536 % (1u<<24)]; acc.x^=v.x; }
        if(KIND==5) asm volatile("s_nop 0");
        if(KIND==6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m) : "v"(r[i]), "v"(r[(i+1)&15]) : "vcc");
      }
    }
    if(S>=64) __builtin_amdgcn_s_sleep(64); else if(S>=8) __builtin_amdgcn_s_sleep(8);
  }
  unsigned a=acc.x^(unsigned)m^(unsigned)(m>>32);
#pragma unroll
  for(int i=0;i<16;i++) a^=r[i];
  if(a==0x12345678u) sink[1]=a+lds[threadIdx.x];
  if((threadIdx.x&63)==0){
    unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* g=gstamps+4*(blockIdx.x*4+(threadIdx.x>>6));
    g[0]=1; g[1]=((hwid>>4)&0x3ff) | ((unsigned long long)(xcc&15)<<10); g[2]=gr0; g[3]=__builtin_amdgcn_s_memrealtime();
  }
}

int main(int argc,char**argv){
  int iters=argc>1?atoi(argv[1]):3000;   // host: iters*256 instructions per wave
  unsigned long long* stamps; unsigned* sink; unsigned long long* gst; uint4* src;
  CK(hipMalloc(&stamps, 4*1024*8)); CK(hipMalloc(&sink,64)); CK(hipMalloc(&gst,4*1024*8)); CK(hipMalloc(&src,(size_t)(1u<<24)*16+65536*16*64));
  hipStream_t s1,s2; CK(hipStreamCreateWithFlags(&s1,hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2,hipStreamNonBlocking));
  auto run=[&](int kind,int gw,int reps,int K,int S,const char*name){
    CK(hipMemset(gst,0,4*1024*8));
    CK(hipDeviceSynchronize());
    if(kind>=0){
      dim3 g(256), b(64*gw);
      switch(kind){
        case 0: hipLaunchKernelGGL(guest_kernel<0>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 1: hipLaunchKernelGGL(guest_kernel<1>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 2: hipLaunchKernelGGL(guest_kernel<2>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 3: hipLaunchKernelGGL(guest_kernel<3>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 4: hipLaunchKernelGGL(guest_kernel<4>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 5: hipLaunchKernelGGL(guest_kernel<5>,g,b,0,s2,gst,sink,src,reps,K,S); break;
        case 6: hipLaunchKernelGGL(guest_kernel<6>,g,b,0,s2,gst,sink,src,reps,K,S); break;
      }
    }
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0,s1));
    hipLaunchKernelGGL(host_kernel,dim3(1024),dim3(64),0,s1,stamps,sink,iters);
    CK(hipEventRecord(e1,s1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    std::vector<unsigned long long> h(4096), g(4096);
    CK(hipMemcpy(h.data(),stamps,4096*8,hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(),gst,4096*8,hipMemcpyDeviceToHost));
    std::vector<unsigned long long> gk; unsigned long long g0=~0ull,g1=0;
    for(int i=0;i<1024;i++) if(g[4*i]){ gk.push_back(g[4*i+1]); g0=std::min(g0,g[4*i+2]); g1=std::max(g1,g[4*i+3]); }
    std::sort(gk.begin(),gk.end());
    unsigned long long h0=~0ull,h1=0; for(int w=0;w<1024;w++){ h0=std::min(h0,h[4*w+2]); h1=std::max(h1,h[4*w+3]); }
    double sum_with=0,sum_wo=0; int n_with=0,n_wo=0;
    const double ninst=(double)iters*256;
    for(int w=0;w<1024;w++){ const bool shared=std::binary_search(gk.begin(),gk.end(),h[4*w+1]); const double c=(double)h[4*w]/ninst; if(shared){sum_with+=c;n_with++;} else {sum_wo+=c;n_wo++;} }
    const double ginst=(double)reps*K;
    const double a=n_with?sum_with/n_with:0, b=n_wo?sum_wo/n_wo:0;
    printf("%-42s host %6.1f us [t=0..%.0f] guest [t=%.0f..%.0f us] | host cyc/instr: free SIMDs %6.3f (%4d)  guest SIMDs %6.3f (%4d) | guest %7.0f instr/wave -> host cycles lost per guest instr %6.2f\n",
           name,(h1-h0)/100.0,(h1-h0)/100.0, kind>=0?((double)g0-(double)h0)/100.0:0.0, kind>=0?((double)g1-(double)h0)/100.0:0.0, b,n_wo,a,n_with,kind>=0?ginst:0.0, (kind>=0&&n_with&&n_wo)?(a-b)*ninst/ginst:0.0);
  };
  run(-1,0,0,0,0,"host alone");
  run(-1,0,0,0,0,"host alone");
  const char* kn[7]={"V_BITOP3","V_PERM","v_mov","ds_write_b32","global_load 16B","s_nop","v_mad_u64_u32"};
  for(int kind=0;kind<7;kind++){
    char nm[96];
    for(int S: {0, 8, 64}){
      int K=64; int reps= kind==4? 600 : (S==0? 6000 : (S==8? 3000: 1000));
      snprintf(nm,96,"guest 2 waves/CU %s K=64 sleep %d",kn[kind],S);
      run(kind,2,reps,K,S,nm);
    }
  }
  run(0,4,6000,64,0,"guest 4 waves/CU V_BITOP3 K=64 sleep 0");
  run(0,1,6000,64,0,"guest 1 wave/CU V_BITOP3 K=64 sleep 0");
  return 0;
}
