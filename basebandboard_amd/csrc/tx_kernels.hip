// tx_kernels.hip -- pulse shaper and transmitter output (first "next" row).
//
// Reference semantics (paths relative to the reference checkout):
//   PRBSShaper  gateware/bbb/bitshaper.py:12-86   8 samples per bit; 8-deep shift register of data
//               bits; 8 ROMs x 8 phases of +-coefficients (address LSB = the data bit, :52-58,:74);
//               3-level adder tree to a 12-bit signed sample (:76-86)
//   TX          gateware/bbb/tx.py:60-81          x = wrap12(bit_en*shaped + noise_en*wrap12(g*noise_var))
//   test model  bitshaper.py:143-155              +-1 impulses at the middle of each bit period through
//               the 64-tap pulse, 13 samples of pipeline delay
//
// GPU formulation: sample n depends on 8 consecutive data bits and a phase,
//     shaped[n] = T[ph][q],  ph = (n-17) & 7,  q = bits M-7..M (oldest in bit 0),  M = (n-17) >> 3,
// with T (8 x 256 int16, 4 KiB) built per block in LDS from the 64 coefficients.  One thread
// produces 16 consecutive samples (two 16-byte stores), reading 16 noise bytes and one 10-bit data
// window.
// Roofline: HBM (2 B written + 1 B noise read per sample).
#include "bbb_common.hpp"
#include "awgn_launch.hpp"

namespace bbb {

struct Coeffs64 { int16_t c[64]; };

__device__ __forceinline__ int wrap12_dev(int v) { return (int)((unsigned)v << 20) >> 20; }

// Q bit j = data bit M0-7+j, j = 0..9 (bits before the first one are 0: the reset shift register).
// `bits` holds data bits m0 .. m0+navail-1 and nothing else may be read: a window reaching past them
// (only bits that no sample of the request needs) takes the bit-by-bit path, which reads 0 there
__device__ __forceinline__ unsigned data_window10(const unsigned long long *__restrict bits, long long m0, unsigned long long navail,
                                                  int source, long long M0) {
    unsigned Q = 0;
    if (source == 0 && M0 - 7 >= m0 && (unsigned long long)(M0 - 7 - m0) + 10 <= navail) {
        const unsigned long long rel = (unsigned long long)(M0 - 7 - m0);
        const unsigned sh = (unsigned)(rel & 63);
        unsigned long long w = bits[rel >> 6] >> sh;
        if (sh > 54) w |= bits[(rel >> 6) + 1] << (64 - sh);
        Q = (unsigned)w & 0x3ffu;
    } else {
#pragma unroll 1
        for (int j = 0; j < 10; j++) {
            const long long m = M0 - 7 + j;
            unsigned b = 0;
            if (m >= 0) {
                if (source == 0) {
                    const unsigned long long rel = (unsigned long long)(m - m0);
                    if (m >= m0 && rel < navail) b = (unsigned)((bits[rel >> 6] >> (rel & 63)) & 1ull);
                } else {
                    b = (m & 255) == 0;                                  // Pulser: counter == 0 (tx.py:28-30)
                }
            }
            Q |= b << j;
        }
    }
    return Q;
}

// The shaper alone (noise off, bits on: PRBSShaper.x, bitshaper.py:25-86): nothing but table rows.  A thread's 8 samples
// start at n = first_sample + 8 g, so c0 = (n - 17) & 7 is the same for every thread: sample e has phase (c0 + e) & 7 and
// window shift (c0 + e) >> 3 in {0, 1}.  TT[q][j] = T[(c0 + j) & 7][q] gives the 8 phases of one window by ONE 16-byte
// read; the rows of the two shifts and a per-dword select with uniform masks give the 8 samples as 4 packed pairs (the
// scheme of the fused kernel, awgn_kernels.hip).  The 4 KiB table is built once per call by shaper_table_kernel and read
// copied into LDS by every block (one 16-byte load per thread; gathering the rows straight from global memory was slower:
// 3.0 against 3.7 TB/s), so the grid can be what streams best -- many short blocks, four 16-byte stores per thread, a
// wave's store covering 1 KiB (profiles/r02_ubench4_stream_patterns.log: every further store of a thread costs rate).
typedef uint32_t shu32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
shaper_table_kernel(Coeffs64 cf, unsigned c0, uint16_t *__restrict TT) {
    for (int e = threadIdx.x; e < 256 * 8; e += blockDim.x) {
        const int q = e >> 3, j = e & 7, ph = (int)((c0 + (unsigned)j) & 7u);
        int sum = 0;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const int c = cf.c[8 * idx + ph];
            sum += ((q >> (7 - idx)) & 1) ? c : -c;
        }
        TT[e] = (uint16_t)(wrap12_dev(sum) & 0xffff);
    }
}

constexpr int kShaperIters = 4;       // 16-byte stores per thread: few (every further store of a thread costs rate), but
                                      // enough to pay for the block's copy of the table
__global__ void __launch_bounds__(256)
shaper_only_kernel(const uint16_t *__restrict TTg, const unsigned long long *__restrict bits, long long m0, unsigned long long navail,
                   int source, unsigned c0, unsigned long long first_sample, unsigned long long nsamples, int16_t *__restrict out) {
    __shared__ __attribute__((aligned(16))) uint16_t TT[256 * 8];
    reinterpret_cast<shu32x4 *>(TT)[threadIdx.x] = reinterpret_cast<const shu32x4 *>(TTg)[threadIdx.x];      // 256 x 16 B = the table
    uint32_t sel[4];
#pragma unroll
    for (unsigned d = 0; d < 4; d++)
        sel[d] = (c0 + 2u * d < 8u ? 0x0000ffffu : 0u) | (c0 + 2u * d + 1u < 8u ? 0xffff0000u : 0u);
    __syncthreads();
    const char *tt = reinterpret_cast<const char *>(TT);
#pragma unroll
    for (int it = 0; it < kShaperIters; it++) {
        const unsigned long long g = ((unsigned long long)blockIdx.x * kShaperIters + it) * 256 + threadIdx.x;
        const unsigned long long base = g * 8;
        if (base >= nsamples) break;
        const long long M0 = ((long long)(first_sample + base) - 17) >> 3;                 // floor
        const unsigned Q = data_window10(bits, m0, navail, source, M0);
        const shu32x4 A = *reinterpret_cast<const shu32x4 *>(tt + ((Q & 255u) << 4));
        const shu32x4 B = *reinterpret_cast<const shu32x4 *>(tt + (((Q >> 1) & 255u) << 4));
        shu32x4 v;
#pragma unroll
        for (unsigned d = 0; d < 4; d++) v[d] = (A[d] & sel[d]) | (B[d] & ~sel[d]);       // samples 2d, 2d+1
        if (base + 8 <= nsamples) {
            *reinterpret_cast<shu32x4 *>(out + base) = v;
        } else {
            for (unsigned e = 0; base + e < nsamples; e++) out[base + e] = (int16_t)((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
        }
    }
}

__global__ void __launch_bounds__(256)
tx_waveform_kernel(Coeffs64 cf, const unsigned long long *__restrict bits, long long m0, unsigned long long navail, int source,
                   const int8_t *__restrict noise, int noise_var, int bit_en, int noise_en,
                   unsigned long long first_sample, unsigned long long nsamples, int16_t *__restrict out) {
    __shared__ int16_t T[8 * 256];
    // T[ph][q]: ROM idx contributes +c[8 idx + ph] when data bit M-idx is 1 (q bit 7-idx), else -c
    for (int e = threadIdx.x; e < 8 * 256; e += blockDim.x) {
        const int ph = e >> 8, q = e & 255;
        int s = 0;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const int c = cf.c[8 * idx + ph];
            s += ((q >> (7 - idx)) & 1) ? c : -c;
        }
        T[e] = (int16_t)wrap12_dev(s);
    }
    __syncthreads();
    // 16 samples per thread: one 16-byte noise load, two 16-byte stores, one 10-bit data window
    const unsigned long long ngroups = (nsamples + 15) / 16;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long base = g * 16;
        const long long np0 = (long long)(first_sample + base) - 17;        // n - 17 of the first sample
        const long long M0 = np0 >> 3;                                      // floor
        const unsigned Q = data_window10(bits, m0, navail, source, M0);
        const bool full = base + 16 <= nsamples;
        typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
        u2 nz = {0, 0};
        if (noise_en) {
            if (full) nz = *reinterpret_cast<const u2 *>(noise + base);
            else for (unsigned e = 0; base + e < nsamples; e++) nz[e >> 3] |= (unsigned long long)(uint8_t)noise[base + e] << (8 * (e & 7));
        }
        int16_t v[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const long long np = np0 + e;
            const int ph = (int)(np & 7);
            const unsigned q = (Q >> (unsigned)((np >> 3) - M0)) & 255u;
            const int shaped = bit_en ? (int)T[ph * 256 + q] : 0;                       // tx.py:65-66
            const int gsample = (int)(int8_t)(nz[e >> 3] >> (8 * (e & 7)));
            const int nmux = noise_en ? wrap12_dev(gsample * noise_var) : 0;             // tx.py:75-77
            v[e] = (int16_t)wrap12_dev(shaped + nmux);                                   // tx.py:80-81
        }
        if (full) {
            typedef short s8 __attribute__((ext_vector_type(8)));
            const s8 lo = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const s8 hi = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            reinterpret_cast<s8 *>(out + base)[0] = lo;
            reinterpret_cast<s8 *>(out + base)[1] = hi;
        } else {
            for (unsigned e = 0; base + e < nsamples; e++) out[base + e] = v[e];
        }
    }
}

int tx_waveform_launch(const int16_t *coeffs, const uint64_t *d_bits, int64_t m0, uint64_t navail, int source, const int8_t *d_noise,
                       int noise_var, int bit_en, int noise_en, uint64_t first_sample, uint64_t nsamples,
                       int16_t *d_out, hipStream_t st) {
    if (nsamples == 0) return BBB_OK;
    Coeffs64 cf;
    for (int i = 0; i < 64; i++) cf.c[i] = coeffs[i];
    const uint64_t groups = (nsamples + 15) / 16;
    uint64_t blocks = (groups + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (!noise_en && bit_en) {
        const uint64_t groups8 = (nsamples + 7) / 8;
        const uint64_t nblk = (groups8 + 256 * kShaperIters - 1) / (256 * kShaperIters);
        if (nblk > 0x7fffffffull) return fail(BBB_EINVAL, "nsamples too large for one call: split it");
        uint16_t *d_tt = nullptr;
        BBB_HIP(hipMallocAsync((void **)&d_tt, 256 * 8 * sizeof(uint16_t), st));
        const unsigned c0 = (unsigned)(((int64_t)first_sample - 17) & 7);
        hipLaunchKernelGGL(shaper_table_kernel, dim3(1), dim3(256), 0, st, cf, c0, d_tt);
        hipLaunchKernelGGL(shaper_only_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const uint16_t *)d_tt,
                           (const unsigned long long *)d_bits, (long long)m0, (unsigned long long)(d_bits ? navail : 0), source, c0,
                           (unsigned long long)first_sample, (unsigned long long)nsamples, d_out);
        const hipError_t e = hipGetLastError();
        (void)hipFreeAsync(d_tt, st);
        if (e != hipSuccess) return fail(BBB_EHIP, hipGetErrorString(e));
        return BBB_OK;
    }
    hipLaunchKernelGGL(tx_waveform_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cf,
                       (const unsigned long long *)d_bits, (long long)m0, (unsigned long long)(d_bits ? navail : 0), source, d_noise, noise_var, bit_en, noise_en,
                       (unsigned long long)first_sample, (unsigned long long)nsamples, d_out);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

// ---------------------------------------------------------------------------------------------
// Receiver front end: sign slicer + sampling phase + clock division.
//   RX.sliced = ~sample[-1]  (sample >= 0 -> 1)        gateware/bbb/rx.py:29
//   BitDelayLine picks the sampling phase               gateware/bbb/delayline.py:45-66, rx.py:32-33
//   one decision per `stride` samples                   rx.py:35-43 (clock division), and
//   software/memdump/decode.py:15-16: (dat > 0)[::4]    (strict threshold, stride 4)
// bit j = decide(sample[phase + j*stride]); lane l of a wave takes bit 64w + l and one wave-wide
// ballot packs the word (LSB first) -- the layout bbb_prbs_check reads.
// ---------------------------------------------------------------------------------------------
// Two forms.  Strides 1, 2, 4: a lane reads 8 consecutive samples with ONE 16-byte load (non-temporal: every sample is read
// once; a wave's load covers 1 KiB), decides its 8 / stride of them, and the `stride` lanes of an output byte OR their
// pieces together (one or two cross-lane moves); little-endian u64 words are those bytes in order.  Any other stride: lane l of a wave takes bit 64w + l (one
// 2-byte load each; with stride 8 a wave touches 8 lines per load) and a wave-wide ballot packs the word.
__global__ void __launch_bounds__(256)
rx_slice_kernel(const int16_t *__restrict samples, unsigned long long nbits, unsigned long long stride,
                unsigned long long phase, int strict, unsigned long long *__restrict out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long nwords = (nbits + 63) / 64;
    const unsigned long long wave0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    unsigned long long w = wave0;
    // four words per trip while they are all whole: four independent loads in flight per lane
    for (; w + 3 * nwaves < nwords && (w + 3 * nwaves) * 64 + 64 <= nbits; w += 4 * nwaves) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(samples + phase + ((w + u * nwaves) * 64 + lane) * stride);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long word = __ballot(strict ? v[u] > 0 : v[u] >= 0);
            if (lane == 0) out[w + u * nwaves] = word;
        }
    }
    for (; w < nwords; w += nwaves) {
        const unsigned long long j = w * 64 + lane;
        int decision = 0;
        if (j < nbits) {
            const int v = samples[phase + j * stride];
            decision = strict ? v > 0 : v >= 0;
        }
        const unsigned long long word = __ballot(decision);
        if (lane == 0) out[w] = word;
    }
}

typedef uint32_t rxu32x4 __attribute__((ext_vector_type(4)));
template <int S>
__global__ void __launch_bounds__(256)
rx_slice_bytes_kernel(const int16_t *__restrict samples, unsigned long long nbits, unsigned long long phase, int strict,
                      uint8_t *__restrict out, unsigned long long nbytes) {
    constexpr int PB = 8 / S;                                       // decisions per lane: 8, 4 or 2
    const unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long nchunks = nbytes * S;                  // a multiple of S: the S lanes of a byte stay together
    for (unsigned long long ch = t0; ch < nchunks; ch += nthreads) {
        const unsigned long long bit0 = ch * PB;
        unsigned piece = 0;
        if (bit0 + PB < nbits) {
            // the lane's 8 samples all exist: the sample of bit bit0 + PB lies behind them
            typedef rxu32x4 __attribute__((aligned(2))) chunk_t;
            const rxu32x4 c = __builtin_nontemporal_load(reinterpret_cast<const chunk_t *>(samples + phase + ch * 8));
#pragma unroll
            for (int e = 0; e < PB; e++) {
                const int idx = e * S;
                const uint32_t dw = c[idx / 2];
                const int v = (int)(int16_t)((idx & 1) ? (dw >> 16) : (dw & 0xffffu));
                piece |= (unsigned)(strict ? v > 0 : v >= 0) << e;
            }
        } else {
            for (int e = 0; e < PB; e++) {
                if (bit0 + e >= nbits) break;
                const int v = samples[phase + (bit0 + e) * S];
                piece |= (unsigned)(strict ? v > 0 : v >= 0) << e;
            }
        }
        unsigned byte = piece << ((unsigned)(ch % S) * PB);
        if (S >= 2) byte |= (unsigned)__shfl_xor((int)byte, 1, 64);
        if (S >= 4) byte |= (unsigned)__shfl_xor((int)byte, 2, 64);
        if (ch % S == 0) out[ch / S] = (uint8_t)byte;
    }
}

int rx_slice_launch(const int16_t *d_samples, uint64_t nbits, uint64_t stride, uint64_t phase, int strict,
                    uint64_t *d_out, hipStream_t st) {
    if (nbits == 0) return BBB_OK;
    const uint64_t nwords = (nbits + 63) / 64;
    if (stride == 1 || stride == 2 || stride == 4) {
        const uint64_t nbytes = nwords * 8;                 // whole words: the bytes behind the last bit are written as 0
        uint64_t blocks = (nbytes * stride + 255) / 256;    // a thread per 8 samples
        if (blocks > 256 * 64) blocks = 256 * 64;
        uint8_t *ob = reinterpret_cast<uint8_t *>(d_out);
        const dim3 grid((unsigned)blocks), block(256);
        if (stride == 1)
            hipLaunchKernelGGL(rx_slice_bytes_kernel<1>, grid, block, 0, st, d_samples, (unsigned long long)nbits, (unsigned long long)phase, strict, ob, (unsigned long long)nbytes);
        else if (stride == 2)
            hipLaunchKernelGGL(rx_slice_bytes_kernel<2>, grid, block, 0, st, d_samples, (unsigned long long)nbits, (unsigned long long)phase, strict, ob, (unsigned long long)nbytes);
        else
            hipLaunchKernelGGL(rx_slice_bytes_kernel<4>, grid, block, 0, st, d_samples, (unsigned long long)nbits, (unsigned long long)phase, strict, ob, (unsigned long long)nbytes);
        BBB_HIP(hipGetLastError());
        return BBB_OK;
    }
    uint64_t blocks = (nwords + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(rx_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_samples, (unsigned long long)nbits,
                       (unsigned long long)stride, (unsigned long long)phase, strict, (unsigned long long *)d_out);
    BBB_HIP(hipGetLastError());
    return BBB_OK;
}

}  // namespace bbb
