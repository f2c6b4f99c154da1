/*
 * bbb_oracle.h -- CPU restatement (plain C) of the basebandboard AWGN / PRBS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/, the smoke
 * check in __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.
 * The product (basebandboard_amd/, libbbb_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  LUTOPT states (every shipped matrix n16..n512, all state bits), the uniform
 * word stream and the CLT tree are held to outputs of the reference's own Python RUN in the build
 * container (tests/golden/ref_*, made by tools/make_golden_ref.py from
 * software/rnghunt/util/binarymatrix.py recur(), util/verify.py, util/pack.py and
 * software/clt-grng/clt-grng-evaluate.py executed unchanged).  In addition
 * tests/golden/{lutopt_clt,prbs,gf2}.json, which tools/make_golden.py derives from the models
 * embedded in the reference's HDL tests (rng.py:134-135, rng.py:173-181, prbs.py:112-113; those
 * files need migen and cannot be imported) and from literal known-answer strings in the
 * reference's Rust tests, pin PRBS and the same LUTOPT/CLT values a second way.  The PRBSErrorDetector FSM has no literal
 * trace anywhere in the reference (its only test uses unseeded random errors and
 * needs migen): it is pinned by re-running the reference's test *protocol*
 * (prbs.py:124-163) as a property.  BER counters / Eb-N0 mapping do not exist in the
 * reference: build-defined, "parity unpinned" (sanity: Q-function).
 *
 * All file:line citations are relative to the reference checkout root.
 */
#ifndef BBB_ORACLE_H
#define BBB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBO_MAX_K 512
#define BBO_WORDS (BBO_MAX_K / 64)

/* A LUTOPT recurrence: row r of A lists the old-state bit indices XORed into new bit r
 * (gateware/bbb/rng.py:38-40; packed form rng.py:42-55). */
typedef struct {
    int k;
    int ntaps[BBO_MAX_K];
    uint16_t taps[BBO_MAX_K][8];
} bbo_lutopt;

/* Parse the 0/1 text matrix format of software/rnghunt/matrices/N (line r, char c = A[r][c];
 * written by rnghunt.rs:51-53, read by util/pack.py:6-18).  Returns 0, or -1 on error. */
int bbo_lutopt_load(bbo_lutopt *m, const char *path);
/* Build from packed tap lists (rng.py:42-55): taps_flat holds row 0's taps, then row 1's ...;
 * row_off[r]..row_off[r+1] delimit row r. */
int bbo_lutopt_from_packed(bbo_lutopt *m, int k, const uint16_t *taps_flat, const uint32_t *row_off);

/* One synchronous step x' = A x over GF(2) (rng.py:38-40).  State bit i lives at
 * x[i/64] >> (i%64) & 1, i.e. bit i of the HDL integer (rng.py:135). */
void bbo_lutopt_step(const bbo_lutopt *m, const uint64_t *x, uint64_t *xnew);

/* CLTGRNG adder tree, literally: log2(n) levels of y[j] = x[2j] - x[2j+1]
 * (rng.py:96-105; clt-grng-evaluate.py:10-15).  Returns the UN-truncated tree value. */
int bbo_clt_tree(const uint64_t *x, int n);
/* Same value by the closed form sum_i (-1)^popcount(i) x[i]. */
int bbo_clt_popcount(const uint64_t *x, int n);
/* Truncate to log2(n) bits, signed (CLTGRNG.x is Signal((logn, True)), rng.py:78,108). */
int bbo_clt_wrap(int tree_value, int n);

/* Sequential AWGN stream: out[i] = wrap(tree(A^(first_step+i+1) init)), i < nsamples.
 * For k = 256 the output is int8 (tx.py:68-71).  `init` is ceil(k/64) words. */
void bbo_awgn_stream_i8(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step,
                        uint64_t nsamples, int8_t *out);
/* Faster k=256-only evaluation of the same stream (byte-indexed column tables +
 * popcount closed form); used as the timed CPU baseline.  Must equal bbo_awgn_stream_i8. */
void bbo_awgn_stream_i8_fast256(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step,
                                uint64_t nsamples, int8_t *out);
/* Bulk forms: states A^(first_step+1+i) init as ceil(k/64) words each; the same as 32-bit words
 * (msb_first = 1: the dieharder dump format of software/rnghunt/util/verify.py:46-52); the tree over
 * caller-supplied words. */
void bbo_lutopt_states(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step, uint64_t nstates,
                       uint64_t *out);
int bbo_lutopt_words_u32(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step, uint64_t nstates,
                         int msb_first, uint32_t *out);
void bbo_clt_tree_bulk(const uint64_t *x, int n, uint64_t nstates, int16_t *out);
/* Final state after `nsteps` steps from init. */
void bbo_lutopt_run(const bbo_lutopt *m, const uint64_t *init, uint64_t nsteps, uint64_t *xout);

/* PRBS-k Fibonacci LFSR (gateware/bbb/prbs.py:14,32-35; model prbs.py:112-113).
 * Returns the tap for k or 0 when k is not one of 7,9,11,15,20,23,31 (prbs.py:29-30). */
int bbo_prbs_tap(int k);
/* Emit nbits bits (one per byte, 0/1) starting from LFSR state *state; updates *state. */
int bbo_prbs_bits(int k, uint64_t *state, uint64_t nbits, uint8_t *bits);
/* Same stream packed LSB-first into 64-bit words: bit t -> word t/64, bit t%64. */
int bbo_prbs_packed(int k, uint64_t *state, uint64_t nbits, uint64_t *words);
/* Word-parallel variant used for the timed CPU baseline (same output as bbo_prbs_packed). */
int bbo_prbs_packed_fast(int k, uint64_t *state, uint64_t nbits, uint64_t *words);
/* Count mismatches between `words` (packed as above) and the PRBS from *state. */
int bbo_prbs_check_packed(int k, uint64_t *state, uint64_t nbits, const uint64_t *words, uint64_t *nerr);

/* PRBSErrorDetector, cycle exact (prbs.py:61-99).  For clock i the input wire holds
 * bits[i]; err[i], reload[i] are the values of `err` and `reload` sampled after that
 * clock edge -- exactly what the reference's testbench reads (prbs.py:146-150). */
int bbo_prbs_detector_run(int k, const uint8_t *bits, uint64_t n, uint8_t *err, uint8_t *reload);

/* Same machine on a packed stream (bit t at word t/64, LSB first); err_words / reload_words may be
 * NULL; stats = {err while reload==0, err, reload clocks, err_count > k/2 events}. */
int bbo_prbs_detector_packed(int k, const uint64_t *words, uint64_t nbits, uint64_t *err_words,
                             uint64_t *reload_words, uint64_t stats[4]);

/* TX noise path + RX slicer (tx.py:75-81, rx.py:29), one sample per bit:
 *   noise = wrap12(g * noise_var), x = wrap12(level(bit) + noise), decision = (x >= 0)
 * level(bit) = bit ? +amp : -amp.  Returns decision (0/1). */
int bbo_txrx_decide(int g_i8, int bit, int amp, int noise_var);

typedef struct {
    int prbs_k;            /* PRBS order */
    uint64_t prbs_state;   /* initial LFSR state (reference reset value: 1) */
    int amp;               /* BPSK level, +-amp added to the noise (12-bit signed domain) */
    int noise_var;         /* 4-bit unsigned multiplier of the CLT sample (tx.py:52,76) */
    uint64_t warmup;       /* LUTOPT steps discarded before the first sample (rng.py:161-162 uses 2*logn) */
    uint64_t first_bit;    /* offset into both streams (for sharded trials) */
    uint64_t nbits;        /* bits in this trial */
} bbo_trial;
/* BUILD-DEFINED (not in the reference): count decision errors of BPSK over PRBS-k
 * through the CLT AWGN stream.  bit t uses PRBS bit (first_bit+t) and the CLT sample of
 * state A^(warmup+first_bit+t+1) init. */
int bbo_ber_trial(const bbo_lutopt *m, const uint64_t *init, const bbo_trial *t,
                  uint64_t *bits_out, uint64_t *errors_out);

/* PRBSShaper (gateware/bbb/bitshaper.py:25-86), 8 samples per bit, for a stream of data bits:
 * an 8-deep shift register of the most recent bits (sr[0] newest, reset 0), 8 coefficient ROMs of
 * 8 entries each addressed by the sample phase, ROM idx contributing +c or -c by sr[idx]
 * (:52-58,:74), summed by a 3-level adder tree into a 12-bit signed output; 13 samples of
 * pipeline / alignment delay as compensated in the reference's test (:155).  Sample n shows
 *   sum_idx  (bit[M-idx] ? +1 : -1) * c[8*idx + ph],  M = floor((n-17)/8), ph = (n-17) mod 8,
 * bits before the first one count as 0 (the reset value of sr).  source: 0 = PRBS-k from
 * prbs_state, 1 = Pulser (tx.py:20-30: one 1 every 256 bit periods, first at bit 0). */
int bbo_shaper_i16(const int16_t coeffs[64], int source, int k, uint64_t prbs_state,
                   uint64_t first_sample, uint64_t nsamples, int16_t *out);
/* TX.x (gateware/bbb/tx.py:60-81): wrap12( bit_en * shaped + noise_en * wrap12(g * noise_var) ),
 * g = CLT sample of LUTOPT state A^(warmup + n + 1) init.  The alignment of the noise stream
 * against the bit stream is build-defined (no vector for it exists in the reference). */
int bbo_tx_i16(const bbo_lutopt *m, const uint64_t *init, const int16_t coeffs[64], int source, int k,
               uint64_t prbs_state, int bit_en, int noise_en, int noise_var, uint64_t warmup,
               uint64_t first_sample, uint64_t nsamples, int16_t *out);

/* RX front end: bit j = decide(samples[phase + j*stride]); decide = (v >= 0) (rx.py:29) or, strict,
 * (v > 0) (software/memdump/decode.py:15).  One bit per byte; returns the number of bits. */
uint64_t bbo_rx_slice(const int16_t *samples, uint64_t nsamples, uint64_t stride, uint64_t phase, int strict,
                      uint8_t *bits);

/* rnghunt BinaryMatrix::recur restated (binary_matrix.rs:53-76): column-major u64 words,
 * MSbit = row 0; x given as one bit per byte; emits bit 0 of each successive A x. */
int bbo_rnghunt_recur(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits,
                      int n, uint8_t *out_bits);

#ifdef __cplusplus
}
#endif
#endif
