// awgn_launch.hpp -- host-side launch entry points of the generator kernels.
#pragma once
#include "bbb_common.hpp"

namespace bbb {

// start states S[g] = B^g s0 (per radix-16 level e and digit j = 1..15 a nibble-combination table of
// B^(j*16^e): [e][j-1][k/4 * 16 * W32]) stored word-major with `stride` words per state word, and
// their bit planes
// s16: the first 16 start states, [16][16] words (host computed)
// slice_mode 0: planes [k][nlanes], 32 generators per lane; 1: the packed n512 layout of awgn512.hip, 16 per lane
// parts: the level kernels stage their jump tables in LDS in 2 or 4 pieces (16 / 8 KiB for k = 256): 4 when the
// seeding is to run beside the transmitter variant of the sample kernel, which leaves less LDS free
int awgn_seed_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states,
                     uint64_t stride, unsigned nlanes, uint32_t *d_planes, hipStream_t st, int slice_mode = 0, int parts = 2);
// round 5, the BER trial's own seeding: the same start states AND their bit planes in TWO launches (seed_head_kernel: the
// first 65536 states; seed_tail_planes_kernel: one block per wave of the consumer -- 2048 generators -- does the top-level
// mat-vecs and the bit transposition in LDS and writes planes[k][nlanes] directly; the packed states above the first
// 65536 never reach memory).  d_top: the plan's merged top level, tables of B^(d 65536) for d = 1 .. kSeedTopTables; k = 256
// (W32 = 8), G <= 2^21; d_states holds 65536 x 8 words.  Two calls, so that a caller can queue other work between them (the
// trial's PRBS seeding goes to its side stream while the head kernel runs)
constexpr int kSeedTopTables = 31;
// `ride`: the trial's PRBS start states (prbs_seed_lanes_launch's arguments) seeded by extra blocks of the head launch, or null
struct PrbsSeedRide { int k; const uint32_t *d_tabs, *s16, *qcol; unsigned nlanes; uint32_t *d_planes; };
int awgn_seed_head_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states, hipStream_t st,
                          const PrbsSeedRide *ride = nullptr);
int awgn_seed_tail_planes_launch(int k, const uint32_t *d_top, uint64_t G, const uint32_t *d_states, unsigned nlanes, uint32_t *d_planes,
                                 hipStream_t st);
// PRBS start states (k <= 31) of the BER kernels' generators, d_states[G] -> bit planes [k][nlanes], two launches; d_tabs = the radix-16
// plan of the LFSR's jump matrix (W32 = 1), s16 as above
int prbs_seed_planes_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, uint64_t G, uint32_t *d_states, unsigned nlanes,
                            uint32_t *d_planes, hipStream_t st);
// the same planes in ONE launch without LDS (round 5): one table-composed state per consumer lane, its other 31 generators by steps
// with Q = B^64 (qcol[c] = column c of Q: JumpPlan::qcol64)
int prbs_seed_lanes_launch(int k, const uint32_t *d_tabs, const uint32_t *s16, const uint32_t *qcol, uint64_t G, unsigned nlanes,
                           uint32_t *d_planes, hipStream_t st);
// awgn512.hip: generated kernel for the shipped n512 matrix (packed state, 16 generators per lane, int16 out)
bool awgn512p_matches(int k, const uint16_t *taps, const uint32_t *row_off);
int bitslice512p_launch(const uint32_t *d_states, uint64_t G, uint64_t stride, unsigned nlanes, uint32_t *d_planes, hipStream_t st);
int awgn512p_fill_launch(const uint32_t *d_planes, int16_t *dst, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                         hipStream_t st);
int awgn256_fill_launch(const uint32_t *d_planes, int8_t *dst, uint64_t nsamples, unsigned L, uint64_t G,
                        unsigned nlanes, hipStream_t st);
// PLANES form of the staged stream: the sample kernel leaves the 8 count planes of every step, u32x4 stage[wave][step][half][lane]
// (nlanes / 64 waves x L steps x 2 KiB); the mover turns them into bytes: bytes [win_lo, win_lo + nbytes) of the staged stream
// (generator g owns [g L, (g + 1) L)) go to dst[0 .. nbytes)
// small_footprint: the form without the advance in front of its loop -- d_planes then is the state OF the first sample (seeded one
// clock further), and the kernel leaves room for two guest waves per SIMD (slower by itself: awgn_kernels.hip)
int awgn256_planes_launch(const uint32_t *d_planes, void *stage, unsigned L, unsigned nlanes, hipStream_t st, bool small_footprint);
int unplane_launch(const void *stage, void *dst, uint64_t win_lo, uint64_t nbytes, unsigned L, uint64_t G, unsigned nlanes, hipStream_t st);
// the SHAPING mover: the same window as the transmitter's int16 output x = wrap12(bit_en * shaped + g * noise_var) at dst[0 .. nsamples)
// d_bits: packed data bits (32-bit words) of THIS window; rel_base = window bit offset of its sample 0; c0 = (first_sample - 17) & 7
int unplane_tx_launch(const void *stage, int16_t *dst, uint64_t win_lo, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                      const int16_t *coeffs, const uint32_t *d_bits, uint32_t nwords32, uint32_t rel_base, uint32_t c0, int noise_var, int bit_en,
                      int use_bits, hipStream_t st);
// the transmitter's output fused into the one-kernel form of the sample kernel (int16 straight to its place)
int awgn256_tx_launch(const uint32_t *d_planes, int16_t *dst, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                      const int16_t *coeffs, const uint32_t *d_bits, uint32_t nwords32, uint32_t rel_base, uint32_t c0, int noise_var,
                      int bit_en, int use_bits, hipStream_t st);
int pulse_bits_launch(uint64_t *dst, int64_t m_first, uint64_t nwords, hipStream_t st);
int widen_i8_i16_launch(const int8_t *src, int16_t *dst, uint64_t n, hipStream_t st);   // n rounded up to 16 by the caller's buffers
int awgn_generic_fill_launch(int k, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2,
                             void *dst, int elem_size, uint64_t nsamples, unsigned L, uint64_t G, unsigned nlanes,
                             hipStream_t st);
// LUTOPT.x as 32-bit words, k/32 per state (k a multiple of 32), table driven
int lutopt_words_launch(int k, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2, uint32_t *dst,
                        uint64_t nstates, unsigned L, uint64_t G, unsigned nlanes, bool msb_first, hipStream_t st);
// the same for the shipped n256 matrix on the generated network (dst 16-byte aligned)
int lutopt_words256_launch(const uint32_t *d_planes, uint32_t *dst, uint64_t nstates, unsigned L, uint64_t G, unsigned nlanes,
                           bool msb_first, hipStream_t st);
int clt_tree_launch(int k, const uint64_t *states, uint64_t nstates, int16_t *out, hipStream_t st);
bool awgn256_matches(int k, const uint16_t *taps, const uint32_t *row_off);
// awgn_small.hip: generated kernels for the shipped n16 / n32 / n64 / n128 matrices
int awgn_small_matches(int k, const uint16_t *taps, const uint32_t *row_off);
int awgn_small_fill_launch(int k, const uint32_t *d_planes, int8_t *dst, uint64_t nsamples, unsigned L, uint64_t G,
                           unsigned nlanes, hipStream_t st);

// fused BER trial kernels (ber_kernels.hip)
struct TrialDev {            // one trial as the kernel sees it
    int32_t prbs_k, prbs_tap;
    int32_t nthr[2];         // number of decision-flip thresholds for bit = 0 / bit = 1
    int32_t thr[2][4];       // error indicator for bit b = XOR_i [ T >= thr[b][i] ],  T = sample + 128 (mod 256)
    uint32_t L;              // bits per generator IN THIS LAUNCH (even)
    uint64_t G, nbits;       // generators; bits this launch counts
    uint32_t flags, last_len;   // kTrialLastLen: the last generator runs `last_len` (<= L, may be 0) of the L steps instead of
                                // nbits - (G - 1) L; kTrialSaveState: the kernel leaves the generator's and the PRBS state in
                                // the plane buffers (the next launch of a continued trial starts from them)
};
enum { kTrialLastLen = 1, kTrialSaveState = 2 };
#define BBB_BER_MAX_GROUP 12
// one launch for `ncfg` channel settings that share one noise / PRBS stream (same geometry in t[0..ncfg))
// d_planes: the bit-sliced states OF the first sample (one clock past the stream position), d_prbs_planes: the LFSR states
int ber256_launch(uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialDev *t, int ncfg, unsigned nlanes,
                  unsigned long long *d_counters, hipStream_t st);
int prbs_state_at_host(int k, uint64_t init_state, uint64_t nbits, uint64_t *state);

// pulse shaper / transmitter output (tx_kernels.hip); d_bits holds data bits m0 .. m0+navail-1 packed LSB first
int tx_waveform_launch(const int16_t *coeffs, const uint64_t *d_bits, int64_t m0, uint64_t navail, int source, const int8_t *d_noise,
                       int noise_var, int bit_en, int noise_en, uint64_t first_sample, uint64_t nsamples,
                       int16_t *d_out, hipStream_t st);

int rx_slice_launch(const int16_t *d_samples, uint64_t nbits, uint64_t stride, uint64_t phase, int strict,
                    uint64_t *d_out, hipStream_t st);

}  // namespace bbb
