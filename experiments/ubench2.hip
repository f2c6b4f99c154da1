// Read-bandwidth patterns for the PRBS checker (design experiment, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// (1) grid-stride: block-contiguous chunks, UNR loads in flight per lane
template<int UNR>
__global__ void __launch_bounds__(256) read_gridstride(const u64x2* __restrict src, size_t n16, u64* out){
  size_t tid = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x*blockDim.x;
  u64 acc=0;
  for(size_t i=tid; i+ (UNR-1)*stride < n16; i+=UNR*stride){
    u64x2 v[UNR];
#pragma unroll
    for(int u=0;u<UNR;u++) v[u]=src[i+u*stride];
#pragma unroll
    for(int u=0;u<UNR;u++) acc += __builtin_popcountll(v[u].x)+__builtin_popcountll(v[u].y);
  }
  if(acc==0x123456789ull) out[0]=acc;
}
// (2) per-wave contiguous regions: wave w reads rows of 1 KiB (16 B per lane), K rows in flight
template<int K, int WPB>
__global__ void __launch_bounds__(64*WPB) read_regions(const u64x2* __restrict src, size_t rows_per_wave, size_t total_rows, u64* out){
  size_t wave = blockIdx.x*(size_t)WPB + (threadIdx.x>>6);
  int lane = threadIdx.x&63;
  size_t r0 = wave*rows_per_wave; 
  size_t r1 = r0+rows_per_wave; if(r1>total_rows) r1=total_rows;
  u64 acc=0;
  for(size_t r=r0; r+K<=r1; r+=K){
    u64x2 v[K];
#pragma unroll
    for(int u=0;u<K;u++) v[u]=src[(r+u)*64+lane];
#pragma unroll
    for(int u=0;u<K;u++) acc += __builtin_popcountll(v[u].x)+__builtin_popcountll(v[u].y);
  }
  if(acc==0x123456789ull) out[0]=acc;
}
static double time_kernel(std::function<void()> f, int reps=5){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  double best=1e30;
  for(int i=0;i<reps;i++){ CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best) best=ms; }
  return best;
}
int main(){
  size_t N = 1250000000ull/16*16; // 1.25 GB
  char* buf; CK(hipMalloc(&buf, N+4096)); CK(hipMemset(buf,1,N));
  u64* out; CK(hipMalloc(&out,8));
  size_t n16=N/16;
  for(int grid: {256*4, 256*8, 256*16}){
    double ms=time_kernel([&]{ hipLaunchKernelGGL(read_gridstride<4>, dim3(grid), dim3(256),0,0,(const u64x2*)buf,n16,out); });
    printf("gridstride UNR4 grid %d: %.3f ms %.1f GB/s\n",grid,ms,N/ms/1e6);
    ms=time_kernel([&]{ hipLaunchKernelGGL(read_gridstride<8>, dim3(grid), dim3(256),0,0,(const u64x2*)buf,n16,out); });
    printf("gridstride UNR8 grid %d: %.3f ms %.1f GB/s\n",grid,ms,N/ms/1e6);
  }
  size_t total_rows = N/1024;
  for(int waves: {1024, 1280, 2048, 4096, 8192, 16384}){
    size_t rpw=(total_rows+waves-1)/waves;
#define RUNR(K,WPB) { double ms=time_kernel([&]{ hipLaunchKernelGGL((read_regions<K,WPB>), dim3(waves/WPB), dim3(64*WPB),0,0,(const u64x2*)buf,rpw,total_rows,out); }); \
    printf("regions K=%d WPB=%d waves %d rpw %zu: %.3f ms %.1f GB/s\n",K,WPB,waves,rpw,ms,N/ms/1e6);}
    RUNR(8,1) RUNR(16,1) RUNR(31,1) RUNR(8,4) RUNR(16,4)
  }
  return 0;
}
