// Round 5: is the persistent streaming writer's deficit against a classic fill a BALANCE problem?  1024 one-wave blocks, each
// (a) owning one contiguous 1.2 MB region (the PRBS generator's partition) or (b) taking 64 / 256 KiB chunks from an atomic
// counter until none is left.  Per wave: s_memrealtime at its first and after its last store -> the spread of finishing times.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ void stream_chunk(char *buf, unsigned long long lo, unsigned long long hi, unsigned lane, V &acc) {
    constexpr unsigned ROW = 64 * sizeof(V);
    for (unsigned long long o = lo; o < hi; o += 8 * ROW) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned long long a = o + (unsigned long long)i * ROW;
            if (a >= hi) break;
            acc.x += 0x9e3779b9u;
            *(reinterpret_cast<V *>(buf + a) + lane) = acc;
        }
    }
}
// DYN 0: static regions; 1: chunks from a counter, chunk c at c * chunk (all waves inside one moving window); 2: chunks from a
// counter, chunk c in stretch c % S at position c / S (S far-apart sequential streams, dynamically balanced: what a generator whose
// waves chain through their own stretch and steal from slow ones would look like to the memory system)
template <typename V, int DYN>
__global__ void __launch_bounds__(64) k(char *buf, unsigned long long total, unsigned long long chunk, unsigned long long *counter,
                                        unsigned long long *stamps) {
    const unsigned lane = threadIdx.x;
    V acc; acc.x = lane; acc.y = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (DYN == 0) {
        const unsigned long long lo = (unsigned long long)blockIdx.x * chunk;
        stream_chunk<V>(buf, lo, lo + chunk < total ? lo + chunk : total, lane, acc);
    } else {
        for (;;) {
            unsigned long long c = 0;
            if (lane == 0) c = atomicAdd(counter, 1ull);
            c = __shfl(c, 0, 64);
            unsigned long long lo = c * chunk;
            if (lo >= total) break;
            if (DYN == 2) {
                const unsigned long long S = gridDim.x, nchunks = (total + chunk - 1) / chunk, per = (nchunks + S - 1) / S;
                const unsigned long long cc = (c % S) * per + c / S;
                if (c / S >= per || cc >= nchunks) continue;
                lo = cc * chunk;
            }
            stream_chunk<V>(buf, lo, lo + chunk < total ? lo + chunk : total, lane, acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}
template <typename V, int DYN> void run(const char *name, char *buf, unsigned long long total, unsigned long long chunk, unsigned grid,
                                         unsigned long long *counter, unsigned long long *stamps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    std::vector<unsigned long long> h(2 * grid);
    for (int rep = 0; rep < 8; rep++) {
        hipMemsetAsync(counter, 0, 8, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, DYN>), dim3(grid), dim3(64), 0, 0, buf, total, chunk, counter, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { best = std::min(best, ms); sum += ms / 6; }
    }
    hipMemcpy(h.data(), stamps, 16ull * grid, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (unsigned i = 0; i < grid; i++) t0 = std::min(t0, h[2 * i]);
    std::vector<double> ends;
    for (unsigned i = 0; i < grid; i++) ends.push_back((h[2 * i + 1] - t0) / 100.0);        // us (100 MHz)
    std::sort(ends.begin(), ends.end());
    printf("%-44s grid %5u chunk %8llu: mean %.4f ms (%.2f TB/s), best %.4f; waves finish at us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f\n", name, grid, chunk,
           sum, total / sum / 1e9, best, ends.front(), ends[grid / 10], ends[grid / 2], ends[grid * 9 / 10], ends.back());
}
int main() {
    const unsigned long long total = 1250000000ull / 4096 * 4096;
    char *buf; unsigned long long *counter, *stamps;
    hipMalloc(&buf, total + 4096); hipMalloc(&counter, 8); hipMalloc(&stamps, 16 * 8192);
    hipMemset(buf, 1, total);
    const unsigned long long region = ((total + 1023) / 1024 + 4095) / 4096 * 4096;
    run<u32x2, 0>("static regions, 8 B/lane", buf, total, region, (unsigned)((total + region - 1) / region), counter, stamps);
    run<u32x4, 0>("static regions, 16 B/lane", buf, total, region, (unsigned)((total + region - 1) / region), counter, stamps);
    for (unsigned waves : {256u, 384u, 512u, 1024u})
        for (unsigned long long c : {128ull << 10, 256ull << 10}) {
            run<u32x2, 1>("dynamic, one moving window, 8 B/lane", buf, total, c, waves, counter, stamps);
            run<u32x2, 2>("dynamic, far-apart stretches, 8 B/lane", buf, total, c, waves, counter, stamps);
            run<u32x4, 1>("dynamic, one moving window, 16 B/lane", buf, total, c, waves, counter, stamps);
            run<u32x4, 2>("dynamic, far-apart stretches, 16 B/lane", buf, total, c, waves, counter, stamps);
        }
    return 0;
}
