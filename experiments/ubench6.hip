// Round 4 microbenchmark (not product code): what does ONE wave per SIMD pay for an instruction that is not a VALU
// instruction?  The sample / BER kernels are one wave per SIMD issuing a V_BITOP3 every ~4 cycles; this measures what an
// s_add, s_load, ds_write, ds_read, global_store, v_accvgpr_write ... between them costs that wave.
// Loop body: 64 independent V_BITOP3 + N extra instructions of one kind, spread evenly.  Printed: cycles per iteration and
// the cycles each extra instruction added.
// Build: hipcc --offload-arch=gfx950 -O3 ubench6.hip -o ubench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

enum { K_NONE=0, K_SALU, K_SMEM8, K_DSW, K_DSR, K_DSW128, K_DSR128, K_GST, K_ACCW, K_ACCR, K_VXOR, K_SNOP, K_DSWADDTID, K_DSADD, K_BCNT, K_SMOVM0, K_VOR_S };

template<int KIND,int N>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1,1)))
k(unsigned long long* stamps, unsigned* sink, const unsigned* cmem, u32x4* gdst, int iters){
  __shared__ unsigned lds[8*1024];
  unsigned r[64];
#pragma unroll
  for(int i=0;i<64;i++) r[i]=threadIdx.x*2654435761u+i;
  unsigned sv=blockIdx.x, acc=0, x=threadIdx.x*7u;
  unsigned av=0;
  u32x8 sm={0,0,0,0,0,0,0,0};
  u32x4 q={1,2,3,4};
  unsigned laddr=threadIdx.x*4;
  u32x4* gp=gdst+(size_t)blockIdx.x*64*64+threadIdx.x;
  __builtin_amdgcn_s_setprio(3);
  const unsigned long long t0=__builtin_amdgcn_s_memtime();
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<64;i++){
      asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r[i]) : "v"(r[i]), "v"(r[(i+7)&63]), "v"(r[(i+13)&63]));
      if(N>0 && (i % (64/(N>0?N:1)))==0){
        if(KIND==K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv) :: "scc");
        if(KIND==K_SMEM8) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(sm) : "s"(cmem));
        if(KIND==K_DSW) asm volatile("ds_write_b32 %0, %1" :: "v"(laddr), "v"(r[i]));
        if(KIND==K_DSR) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(laddr));
        if(KIND==K_DSW128) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(q));
        if(KIND==K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(laddr));
        if(KIND==K_GST) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(gp), "v"(q));
        if(KIND==K_ACCW) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(av) : "v"(r[i]));
        if(KIND==K_ACCR) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(av));
        if(KIND==K_VXOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(r[i]), "v"(r[(i+1)&63]));
        if(KIND==K_SNOP) asm volatile("s_nop 0");
        if(KIND==K_DSWADDTID) asm volatile("ds_write_addtid_b32 %0" :: "v"(r[i]) );
        if(KIND==K_DSADD) asm volatile("ds_add_u32 %0, %1" :: "v"(laddr), "v"(r[i]));
        if(KIND==K_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(r[i]));
        if(KIND==K_SMOVM0) asm volatile("s_add_u32 m0, %0, 0x100" :: "s"(sv) : "scc");
        if(KIND==K_VOR_S) asm volatile("v_or_b32 %0, %1, %2" : "=v"(x) : "s"(sv), "v"(r[i]));
      }
    }
    if(KIND==K_SMEM8||KIND==K_DSR||KIND==K_DSR128) asm volatile("s_waitcnt lgkmcnt(0)");
    if(KIND==K_GST) gp+=64;     // (one VALU pair per iteration)
  }
  const unsigned long long t1=__builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned a=acc^x^sv^sm[0]^sm[7]^q[0]^q[3];
#pragma unroll
  for(int i=0;i<64;i++) a^=r[i];
  if(KIND==K_ACCW){ unsigned y; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(av)); a^=y; }
  if(a==0x12345678u) sink[0]=a+lds[threadIdx.x];
  if(threadIdx.x==0) stamps[blockIdx.x]=t1-t0;
}

static unsigned long long* stamps; static unsigned* sink; static unsigned* cmem; static u32x4* gdst;
static double base_cyc=0;

template<int KIND,int N>
void run(const char* name,int iters){
  CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int rep=0;rep<2;rep++){
    CK(hipEventRecord(e0,0));
    hipLaunchKernelGGL((k<KIND,N>),dim3(1024),dim3(64),0,0,stamps,sink,cmem,gdst,iters);
    CK(hipEventRecord(e1,0));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  std::vector<unsigned long long> h(1024);
  CK(hipMemcpy(h.data(),stamps,1024*8,hipMemcpyDeviceToHost));
  std::sort(h.begin(),h.end());
  const double med=(double)h[512]/iters;
  if(KIND==K_NONE) base_cyc=med;
  printf("%-34s N=%2d  %8.1f cycles/iteration (median wave; min %.1f max %.1f)  %6.3f cyc per V_BITOP3-slot  extra per instr %6.2f cycles   [%.3f ms]\n",
         name,N,med,(double)h[0]/iters,(double)h[1023]/iters,med/64.0, N? (med-base_cyc)/N : 0.0, ms);
}

int main(int argc,char**argv){
  int iters=argc>1?atoi(argv[1]):4000;
  CK(hipMalloc(&stamps,1024*8)); CK(hipMalloc(&sink,64)); CK(hipMalloc(&cmem,4096)); CK(hipMemset(cmem,0,4096));
  CK(hipMalloc(&gdst,(size_t)1024*64*64*16*2));
  run<K_NONE,0>("baseline 64 V_BITOP3",iters);
  run<K_NONE,0>("baseline 64 V_BITOP3",iters);
  run<K_VXOR,16>("v_xor_b32",iters);
  run<K_BCNT,16>("v_bcnt_u32_b32 (accumulate)",iters);
  run<K_VOR_S,16>("v_or_b32 with SGPR operand",iters);
  run<K_ACCW,16>("v_accvgpr_write",iters);
  run<K_ACCR,16>("v_accvgpr_read",iters);
  run<K_SALU,16>("s_add_u32",iters);
  run<K_SALU,64>("s_add_u32",iters);
  run<K_SMOVM0,16>("s_add_u32 m0",iters);
  run<K_SNOP,16>("s_nop 0",iters);
  run<K_SMEM8,8>("s_load_dwordx8",iters);
  run<K_SMEM8,16>("s_load_dwordx8",iters);
  run<K_DSW,8>("ds_write_b32",iters);
  run<K_DSW,16>("ds_write_b32",iters);
  run<K_DSWADDTID,16>("ds_write_addtid_b32",iters);
  run<K_DSADD,16>("ds_add_u32",iters);
  run<K_DSR,8>("ds_read_b32",iters);
  run<K_DSR,16>("ds_read_b32",iters);
  run<K_DSW128,8>("ds_write_b128",iters);
  run<K_DSR128,8>("ds_read_b128",iters);
  run<K_GST,2>("global_store_dwordx4 nt",iters/4);
  run<K_GST,4>("global_store_dwordx4 nt",iters/4);
  return 0;
}
