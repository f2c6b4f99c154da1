/*
 * bbb.h -- C ABI of libbbb_hip.so: the MI355X (gfx950) implementation of basebandboard's
 * AWGN / PRBS Monte-Carlo path.
 *
 * The reference has no FFI for this path (it is migen gateware plus one numpy script),
 * so each entry point below names the reference interface whose semantics it carries;
 * paths are relative to the reference checkout.  A host in any language binds these
 * directly (INTEGRATION.md shows the ctypes and Rust `extern "C"` stubs).
 *
 * Conventions
 *  - plain C types only; every function returns BBB_OK (0) or a negative BBB_E* code;
 *    nothing throws across the boundary.  bbb_strerror() names a code,
 *    bbb_last_error_detail() gives the failing HIP call of the calling thread.
 *  - pointers named *_dev are device memory on the handle's / call's device (hipMalloc or
 *    a torch tensor's data_ptr()); all other pointers are host memory.
 *  - work is enqueued on the given hipStream_t (passed as void*; NULL = default stream)
 *    and is asynchronous unless the function returns a result to host memory.
 *  - there is NO CPU execution path: a call without a usable gfx950 device fails with
 *    BBB_ENODEV.
 *  - a handle is not thread-safe; different handles may be used from different threads.
 *  - bit order: state bit i of a k-bit state is word[i/64] >> (i%64) & 1, the bit order of
 *    the HDL integer (gateware/bbb/rng.py:135); packed PRBS bit t is word[t/64] >> (t%64) & 1.
 */
#ifndef BBB_H
#define BBB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBB_OK        0
#define BBB_EINVAL   -1   /* bad argument; includes "k invalid for PRBS" (prbs.py:29-30, 55-56) */
#define BBB_ENOMEM   -2
#define BBB_EHIP     -3   /* a HIP runtime call failed */
#define BBB_EIO      -4   /* matrix file unreadable / malformed */
#define BBB_ENODEV   -5   /* no gfx950 device available */
#define BBB_EUNSUP   -6   /* valid request this build has no kernel for */

#define BBB_ABI_VERSION 1
#define BBB_MAX_K 512

int bbb_abi_version(void);
const char *bbb_strerror(int code);
const char *bbb_last_error_detail(void);
int bbb_device_count(int *count);
/* free() for buffers returned by bbb_lutopt_load_matrix_file */
void bbb_free(void *p);

/* ---- uniform + Gaussian generator: LUTOPT -> CLTGRNG ------------------------------------ */

typedef struct bbb_lutopt bbb_lutopt;

/* Read a recurrence matrix in the text format of software/rnghunt/matrices/N (N lines of N
 * chars '0'/'1', line r char c = A[r][c]; written by software/rnghunt/src/bin/rnghunt.rs:51-53,
 * read by software/rnghunt/util/pack.py:6-18) into packed tap lists: row r's taps are
 * taps[row_off[r] .. row_off[r+1]) -- the `packed` argument of LUTOPT.from_packed
 * (gateware/bbb/rng.py:42-55).  Caller frees *taps and *row_off with bbb_free. */
int bbb_lutopt_load_matrix_file(const char *path, int *k, uint16_t **taps, uint32_t **row_off);

/* LUTOPT(a, init) / LUTOPT.from_packed(packed, init): gateware/bbb/rng.py:21-55.
 * Any k in [2, 512] (LUTOPT puts no constraint on k; gateware/bbb/rng_recurrences.py:105 ships n192,
 * rnghunt searches n = 192: rnghunt.rs:14); every row 1..8 taps.  Only the CLTGRNG entry points
 * (bbb_awgn_fill_*, bbb_tx_fill_i16 with noise) need k to be a power of two (rng.py:72-76) and
 * return BBB_EUNSUP otherwise.
 * init_words = ceil(k/64) words; the reference default is 1 (bit 0 set, rng.py:21).
 * device = -1 makes a host-only handle: bbb_lutopt_state_at works (pure GF(2) algebra), every
 * compute call on it returns BBB_ENODEV. */
int bbb_lutopt_create(bbb_lutopt **h, int k, const uint16_t *taps, const uint32_t *row_off,
                      const uint64_t *init_words, int device);
int bbb_lutopt_destroy(bbb_lutopt *h);
/* Bind the handle to a HIP stream (NULL = the default stream): every later call is ordered on it like a kernel launch -- behind what the
 * caller had queued there, in front of what the caller queues next.  The library orders ITS OWN work across a re-bind (kernels, staging
 * slots, start-state buffers of earlier calls, whichever stream they were made on).  The CALLER's work is the caller's to order, as with
 * any stream-ordered library: a buffer that was filled, or is still being read, under one stream must not be handed to a call under
 * another stream without an event between the two (tests/test_gpu_staged.py's random mix does exactly that). */
int bbb_lutopt_set_stream(bbb_lutopt *h, void *hip_stream);
/* 1 when the handle runs the generated straight-line kernel (the n256 matrix of
 * gateware/bbb/rng_recurrences.py:172-259 used by tx.py:70), 0 for the table-driven one. */
int bbb_lutopt_is_specialised(const bbb_lutopt *h);

/* Device-side timing of the generator kernels with hipEvents recorded on the handle's stream
 * around every bbb_awgn_fill_i8 of a specialised handle: [start, seeding kernels, sample kernel].
 * bbb_lutopt_profile_read waits for the recorded events and returns the accumulated
 * milliseconds (seeding / sample kernel) and the number of calls; reset != 0 clears the sums.
 * A sample kernel that is dispatched while the previous one still holds the machine (the staged form with announced
 * start states) is counted from that kernel's completion, not from its own dispatch: time on the machine. */
int bbb_lutopt_profile(bbb_lutopt *h, int enable);
int bbb_lutopt_profile_read(bbb_lutopt *h, double *seed_ms, double *kernel_ms, uint64_t *calls, int reset);
/* The same for the second kernel of the two-kernel form (the mover): accumulated milliseconds
 * and number of movers since the last reset. */
int bbb_lutopt_profile_read_mover(bbb_lutopt *h, double *mover_ms, uint64_t *calls, int reset);

/* LUTOPT.x after `nsteps` clocks from reset (rng.py:38-40), by GF(2) jump-ahead
 * (the x' = A x algebra of software/rnghunt/src/binary_matrix.rs:53-76). Host result. */
int bbb_lutopt_state_at(bbb_lutopt *h, uint64_t nsteps, uint64_t *state_words);

/* LUTOPT.x in bulk -- the uniform word stream itself (rng.py:29-40: "outputs x ... on each clock"; the
 * reference dumps it for dieharder in software/rnghunt/util/verify.py:37-52, 200 000 states as 32-bit
 * words).  State A^(first_step + i + 1) init, 0 <= i < nstates, as k/32 consecutive words:
 *   dst_dev[i * (k/32) + j] = state bits 32j .. 32j+31,
 *   msb_first = 0: bit 32j is the LSB (the HDL integer of rng.py:135, little-endian words);
 *   msb_first = 1: bit 32j is the MSB (verify.py:46-52 prints x[0] x[1] ... and reads 32 characters as a
 *                  binary number) -- the words of its `outnums` file in order.
 * k must be a multiple of 32.  Asynchronous on the handle's stream. */
int bbb_lutopt_fill_words(bbb_lutopt *h, uint32_t *dst_dev, uint64_t nstates, uint64_t first_step, int msb_first);

/* The CLTGRNG sample stream (gateware/bbb/rng.py:70-108; tx.py:70-71):
 *   dst_dev[i] = trunc_signed_log2k( tree( A^(first_step + i + 1) * init ) ),  0 <= i < nsamples
 * i.e. sample i is the adder-tree value of the LUTOPT state after first_step+i+1 clocks,
 * exactly the sequential stream a single CLTGRNG emits (pipeline delay removed).
 * Output is int8 (for k = 256: -128..127, +128 wraps to -128 as the 8-bit Signal does).
 * dst_dev must be 16-byte aligned.  Asynchronous on the handle's stream. */
int bbb_awgn_fill_i8(bbb_lutopt *h, int8_t *dst_dev, uint64_t nsamples, uint64_t first_step);
/* Optional hint: the NEXT bbb_awgn_fill_i8 on this handle will ask for exactly (nsamples, first_step).
 * The start states of that fill are then derived right away on an internal side stream -- the small
 * seeding kernels fit beside the running sample kernel (which leaves ~110 registers per SIMD and
 * 32 KiB of LDS per CU unused) -- and the matching fill only waits for them.  A fill with other
 * arguments ignores the hint.  Results are identical with or without it. */
int bbb_awgn_prefetch(bbb_lutopt *h, uint64_t nsamples, uint64_t first_step);
/* Choose the two-kernel ("staged") form of the k = 256 sample stream for this handle's large fills
 * (bbb_awgn_fill_i8, bbb_tx_fill_i16 of 2^24 samples and more).  Results are identical; what changes is how the
 * bytes reach HBM.  The one-kernel form writes every generator's 16 new bytes straight to their place: 62.5 M
 * scattered pieces per 1e9 samples, one DRAM row activation each.  The staged form has the sample kernel leave its
 * pieces in an internal buffer as full lines and a second kernel move them with full-line reads and writes.  That
 * second kernel runs on the caller's stream, which first waits for the sample kernel (on an internal stream): anything
 * queued after the call sees the output complete, as before; but the NEXT fill's arithmetic does not wait for it -- the
 * mover (memory bound) and the next
 * arithmetic (integer-issue bound) share the machine.  Worth it for back-to-back fills; a single isolated fill
 * finishes later than in the one-kernel form.  Costs two staging buffers of the fill's size.
 * enable = m in 2..8 adds LOOK-AHEAD for bbb_awgn_fill_i8 (and for bbb_tx_fill_i16 with noise on this handle, while the
 * configuration stays the same and m n < 2^34): a fill of n samples at `first` lets its sample kernel produce
 * the m n samples from `first` (one seeding, one launch for m fills; the rest waits in the staging buffer); while the
 * following fills ask for exactly (n, first + n), (n, first + 2 n) ... -- a consumer reading the one sequential stream
 * the reference's generator emits -- each only costs its piece mover.  A fill elsewhere discards what still waits (that
 * work was wasted).  Staging buffers are m times as large.  Results are identical in every mode; calling this function
 * (with any value) drops whatever waits. */
int bbb_lutopt_set_staged(bbb_lutopt *h, int enable);
/* Same stream as int16 (needed for k = 512, whose CLTGRNG output is 9 bits: rng.py:78). */
int bbb_awgn_fill_i16(bbb_lutopt *h, int16_t *dst_dev, uint64_t nsamples, uint64_t first_step);

/* The sample stream as an OBJECT that is drained sequentially -- what a CLTGRNG is: one value per clock, in order
 * (gateware/bbb/rng.py:70-108; tx.py:70-71 is its only consumer).  The stream owns everything a fast sequential reader
 * would otherwise have to choreograph with bbb_lutopt_set_staged / bbb_awgn_prefetch: it turns the two-kernel form on
 * for the handle (with two reads per sample kernel, level 2 of bbb_lutopt_set_staged, unless the caller had chosen a
 * level: staging then takes 4 bytes per sample of a read), announces every next read itself (so the start states are
 * derived beside the running kernel) and restores the handle's mode when it is closed.  A host simply calls
 * bbb_awgn_stream_next in a loop.
 *   open   nsamples_per_call = the length bbb_awgn_stream_next delivers (and the length the stream prepares for);
 *          first_step = LUTOPT clocks before the first sample, as in bbb_awgn_fill_i8; elem_bytes 1 (int8, k <= 256)
 *          or 2 (int16).  One stream per handle at a time (BBB_EINVAL otherwise); the handle must outlive it.
 *   next   the next nsamples_per_call samples to dst_dev (16-byte aligned, elem_bytes * nsamples_per_call bytes),
 *          asynchronous on the handle's stream like bbb_awgn_fill_i8.
 *   read   the next `nsamples` samples, any length (a ragged tail, a short probe): the stream continues behind them.
 *   seek   continue at another position (what was produced ahead at the old one is dropped, the next read announced there).
 *   tell   *next_step = clocks before the sample the next read starts with.
 * Other calls on the handle between two reads are allowed (they cost the pending announcement at most).  Every byte is
 * the one bbb_awgn_fill_i8 / _i16 would deliver for the same position. */
typedef struct bbb_awgn_stream bbb_awgn_stream;
int bbb_awgn_stream_open(bbb_lutopt *h, uint64_t nsamples_per_call, uint64_t first_step, int elem_bytes, bbb_awgn_stream **s);
int bbb_awgn_stream_next(bbb_awgn_stream *s, void *dst_dev);
int bbb_awgn_stream_read(bbb_awgn_stream *s, void *dst_dev, uint64_t nsamples);
int bbb_awgn_stream_seek(bbb_awgn_stream *s, uint64_t first_step);
int bbb_awgn_stream_tell(const bbb_awgn_stream *s, uint64_t *next_step);
int bbb_awgn_stream_close(bbb_awgn_stream *s);

/* CLTGRNG adder tree on caller-supplied uniform words (the loop body of
 * software/clt-grng/clt-grng-evaluate.py:8-16): states_dev holds nstates states of
 * ceil(k/64) u64 words each; out_dev[i] = un-truncated tree value (int16). */
int bbb_clt_tree_i16(int k, const uint64_t *states_dev, uint64_t nstates, int16_t *out_dev,
                     int device, void *hip_stream);

/* ---- PRBS generator / checker ---------------------------------------------------------- */

/* PRBS(k).x (gateware/bbb/prbs.py:23-35; TAPS prbs.py:14): bits first_bit .. first_bit+nbits-1
 * of the sequence started from LFSR state init_state (reference reset value 1), packed
 * LSB-first into u64 words.  Writes ceil(nbits/64) words; unused high bits of the last
 * word are zero.  dst_dev must be 16-byte aligned. */
int bbb_prbs_fill(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                  uint64_t *dst_packed_dev, int device, void *hip_stream);
/* The same fill with a hint about what the caller does next.  BBB_PRBS_WILL_READ_BACK: the buffer is about to be read
 * back (a loopback: bbb_prbs_check / bbb_prbs_detector_stream right behind the fill).  The generator then writes with
 * non-temporal stores: the fill itself takes ~15 % longer, but it leaves no dirty lines in the memory-side cache for the
 * reader to write back, and fill + check together finish sooner.  Same bits either way. */
#define BBB_PRBS_WILL_READ_BACK 1u
int bbb_prbs_fill_hint(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                       uint64_t *dst_packed_dev, unsigned flags, int device, void *hip_stream);
/* Phase-known checker: number of positions where src differs from the same PRBS.  This is
 * the steady-state (`reload == 0`) behaviour of PRBSErrorDetector.err (prbs.py:79) summed
 * over the stream.  *nerr is a host result (the call synchronises the stream). */
int bbb_prbs_check(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                   const uint64_t *src_packed_dev, uint64_t *nerr, int device, void *hip_stream);
/* Same, accumulating into a device counter (no synchronisation): *nerr_dev += mismatches. */
int bbb_prbs_check_dev(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                       const uint64_t *src_packed_dev, uint64_t *nerr_dev, int device,
                       void *hip_stream);
/* LFSR state after nbits clocks (prbs.py:35 iterated), by jump-ahead.  Host result. */
int bbb_prbs_state_at(int k, uint64_t init_state, uint64_t nbits, uint64_t *state);

/* PRBSErrorDetector (gateware/bbb/prbs.py:43-99), cycle exact, for `nstreams` independent
 * detectors run in parallel (one per GPU lane).  bits_dev[s*n + i] is the input wire of
 * stream s during clock i (0/1); err_dev / reload_dev [s*n + i] are `err` and `reload`
 * sampled after that clock edge -- what the reference testbench reads (prbs.py:146-150).
 * Either output may be NULL. */
int bbb_prbs_detector_run(int k, const uint8_t *bits_dev, uint64_t nstreams, uint64_t n,
                          uint8_t *err_dev, uint8_t *reload_dev, int device, void *hip_stream);

/* The same detector over ONE long stream (chunked execution with state
 * hand-off).  bits_packed_dev: input wire during clock t at word t/64, bit t%64 (the layout
 * bbb_prbs_fill / bbb_rx_slice write).  err_packed_dev / reload_packed_dev (either may be NULL,
 * ceil(nbits/64) words): `err` / `reload` sampled after clock t, same packing.  The stream is cut
 * into chunks of chunk_bits (0: 4096 ... 32768 by the length; a multiple of 64), each run by one GPU lane from a
 * speculative state obtained by running the reset detector over the warm_bits (0: 1024) before the
 * chunk; every chunk whose speculative start differs from its predecessor's true end state is run
 * again from that state until the chain is consistent, so the outputs equal the serial machine's
 * bit for bit (prbs.py:61-99).  (A re-run stops where its state meets the speculative run's: from there on the chunk
 * is what the first pass made of it.)  *stats is a host result (the call synchronises the stream).  The call's device
 * workspace (72 bytes per chunk: 22 MB at 1e10 bits; kept up to 256 MiB) and a pinned read-back buffer stay allocated
 * between calls, one set per device and concurrent call. */
typedef struct {
    uint64_t bits;            /* clocks processed */
    uint64_t errors;          /* clocks with err == 1 and reload == 0 (what the reference test compares, prbs.py:152-163) */
    uint64_t errors_raw;      /* clocks with err == 1 */
    uint64_t reload_clocks;   /* clocks with reload == 1 */
    uint64_t resyncs;         /* times err_count exceeded k/2 (prbs.py:92), the reload out of reset included */
    uint64_t chunks;          /* execution detail: chunks, chunk re-runs needed, 1 if the serial guard ran */
    uint64_t chunks_rerun;
    uint64_t serial_fallback;
} bbb_detector_stats;
int bbb_prbs_detector_stream(int k, const uint64_t *bits_packed_dev, uint64_t nbits, uint64_t *err_packed_dev,
                             uint64_t *reload_packed_dev, bbb_detector_stats *stats, uint64_t chunk_bits,
                             uint64_t warm_bits, int device, void *hip_stream);

/* A sample kernel built for the handle's own matrix (one that is not among the shipped ones, e.g. a result of
 * bbb_lutopt_search): `fn` launches it on hip_stream for the bit-plane start states the library prepares
 * (planes_dev[p * nlanes + lane_global] = state bit p of 32 generators, the state BEFORE the first sample; generator
 * numbering and segment geometry as csrc/custom_fill_template.hip uses them) and returns 0 or a hipError_t.
 * bbb_awgn_fill_i8 then calls it instead of the table-driven kernel.  basebandboard_amd.LUTOPT.specialise() generates,
 * compiles (hipcc) and attaches such a kernel; fn == NULL detaches.  k must be a power of two <= 256. */
typedef int (*bbb_custom_fill_fn)(const uint32_t *planes_dev, int8_t *dst_dev, uint64_t nsamples, uint32_t L, uint64_t G,
                                  uint32_t nlanes, void *hip_stream);
int bbb_lutopt_set_custom_fill(bbb_lutopt *h, bbb_custom_fill_fn fn);
/* The same for the fused BER trial (k = 256 only): `trials` points at the library's internal per-trial records
 * (csrc/awgn_launch.hpp, TrialDev) and is passed through unchanged; returns 0 or a BBB_E* code.  Here planes_dev holds
 * the state OF the first sample (one clock further than for the sample kernel), and a continued trial (bbb_ber_run_*)
 * has the kernel write the states it ends in back to both buffers. */
typedef int (*bbb_custom_ber_fn)(uint32_t *planes_dev, uint32_t *prbs_planes_dev, const void *trials, int ncfg,
                                 uint32_t nlanes, uint64_t *counters_dev, void *hip_stream);
int bbb_lutopt_set_custom_ber(bbb_lutopt *h, bbb_custom_ber_fn fn);
/* Load a library built from csrc/custom_fill_template.hip for THIS handle's matrix (basebandboard_amd/
 * gen_lutopt_kernel.py <taps> custom_gen.inc; hipcc --offload-arch=gfx950 -shared -DBBB_N=k -DBBB_LOG=log2 k ...)
 * and attach what it exports: the sample kernel (bbb_custom_fill) and, for k = 256, the BER kernels
 * (bbb_custom_ber).  The C-ABI form of LUTOPT.specialise(): a host without Python generates and compiles the
 * library at build time and attaches it here.  The library stays loaded for the life of the process. */
int bbb_lutopt_attach_custom_library(bbb_lutopt *h, const char *path);

/* ---- fused Monte-Carlo trial: PRBS -> BPSK + scaled CLT noise -> slicer -> error count --- */

/* One trial.  Bit t (0 <= t < nbits) uses PRBS bit first_bit+t and the CLT sample of LUTOPT
 * state A^(warmup+first_bit+t+1) init; the channel is the TX noise path and RX slicer:
 *   noise = wrap12(sample * noise_var)            gateware/bbb/tx.py:75-77
 *   x     = wrap12((bit ? +amp : -amp) + noise)   gateware/bbb/tx.py:80-81
 *   bit^  = (x >= 0)                              gateware/bbb/rx.py:29
 * and an error is bit^ != bit.  The counters and the (amp, noise_var) <-> Eb/N0 mapping are
 * build-defined: the reference has neither. */
typedef struct {
    int32_t  prbs_k;       /* 7, 9, 11, 15, 20, 23 or 31 */
    int32_t  amp;          /* 0..2047 */
    int32_t  noise_var;    /* 0..15 (4-bit unsigned, tx.py:52) */
    int32_t  reserved;
    uint64_t prbs_state;   /* initial LFSR state, non-zero, < 2^k */
    uint64_t warmup;       /* LUTOPT clocks discarded first (rng.py:161-162 uses 2*log2 k) */
    uint64_t first_bit;
    uint64_t nbits;
} bbb_trial_cfg;

typedef struct { uint64_t bits, errors; } bbb_ber;

/* Run ncfg trials on the handle's generator; out[i] (host) receives trial i's counters. */
int bbb_ber_trials(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, bbb_ber *out);
/* Same, adding into device counters counters_dev[2*i] (bits), [2*i+1] (errors) without
 * synchronising -- the buffer a multi-GPU host hands to one RCCL all-reduce (ncclUint64, sum).
 * Calls queued back to back overlap: the start states of a trial (generators and PRBS) are derived on internal streams into
 * a second set of buffers while the kernel of the trial before runs on the handle's stream, and its kernel follows directly. */
int bbb_ber_trials_dev(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, uint64_t *counters_dev);

/* A trial group CONTINUED over several calls (round 4).  One bbb_ber_trials call spends 0.16 ms in front of its kernel on
 * the start states of 2 M generators, whatever its length; a Monte-Carlo run that keeps adding bits until it has seen
 * enough errors pays that per call.  Here a BLOCK of calls_per_block (m) calls shares ONE seeding: the block's m * nbits
 * bits are cut into the generators' segments once, call c runs the c-th m-th of every segment and the kernel leaves the
 * generators' and the PRBS states where the next call finds them (in buffers the run owns: other calls on the handle do
 * not disturb it).  cfgs: up to BBB_BER_MAX_GROUP settings that share prbs_k, prbs_state, warmup, first_bit and nbits
 * (= bits per call; single-threshold channels unless ncfg = 1) -- one pass over the noise stream serves all of them.
 * Block b covers bits [first_bit + b m nbits, first_bit + (b + 1) m nbits).  The (bit, sample) pairs of a block are
 * exactly those of one bbb_ber_trials call over it, visited in another order: after every m-th call the totals equal
 * that call's counters bit for bit (between block ends a call may count up to one segment fewer or more bits than
 * nbits; `bits` always says how many were counted).  The reference's counterpart is the free-running generator itself:
 * one LFSR, one LUTOPT, clocked for as long as the test lasts (gateware/bbb/tx.py:56-81, prbs.py:32-35).
 * _next: runs the next call on the handle's stream and returns the running totals of the whole run (host, synchronises;
 * totals may be NULL: no read-back, no synchronisation).  _next_dev: ADDS the call's counters to counters_dev
 * ([ncfg][2] uint64, device) without synchronising.  _tell: calls made so far / first bit of the next block to start.
 * LIFETIME: a run borrows its handle -- every call on the run, _close included, uses the handle's streams and plans: close
 * the run BEFORE bbb_lutopt_destroy of its handle (a run that outlives its handle is a use after free). */
typedef struct bbb_ber_run bbb_ber_run;
int bbb_ber_run_open(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, uint32_t calls_per_block, bbb_ber_run **out);
int bbb_ber_run_next(bbb_ber_run *r, bbb_ber *totals);
int bbb_ber_run_next_dev(bbb_ber_run *r, uint64_t *counters_dev);
int bbb_ber_run_tell(const bbb_ber_run *r, uint64_t *calls_done, uint64_t *next_block_first_bit);
int bbb_ber_run_close(bbb_ber_run *r);

/* The sweep sharded over the GPUs of ONE process (BASELINE.json configs[4]).  The
 * reference's only multi-worker program has this shape -- workers plus one channel back,
 * software/rnghunt/src/bin/rnghunt.rs:16-18,54-65 -- and no collective of its own; here the channel back is the
 * path's single collective.  handles[r] is a generator created on device r's GPU (all on distinct devices, same
 * matrix).  One host thread per device runs that device's share of the trials, then ONE
 * ncclAllReduce(ncclUint64, ncclSum) of the uint64[2 * ncfg] {bits, errors} counters over xGMI (RCCL,
 * ncclCommInitAll over the handles' devices; the communicators are cached per device list) leaves the totals on
 * every device; out[i] (host) receives them.  Integer sums: the result does not depend on ndev or on the order
 * of the reduction.  mode selects the share of rank r (bbb_sweep_shard computes it, host only):
 *   BBB_SHARD_TRIALS  trial i runs on rank i % ndev (independent trials, round robin);
 *   BBB_SHARD_SEEDS   every rank runs EVERY trial on its own handle -- give the handles different reset states
 *                     (seeds): the points x seeds form, N times the bits per point in the time of one sweep;
 *   BBB_SHARD_BITS    every rank runs every trial over ITS slice of the trial's bit range
 *                     [first_bit + r nbits / ndev, first_bit + (r+1) nbits / ndev) -- same reset state on all
 *                     handles; the totals equal the single-device counters of the same trials exactly;
 *   BBB_SHARD_GROUPS  consecutive trials that read the same noise and PRBS streams (same prbs_k, prbs_state, warmup,
 *                     first_bit, nbits: an Eb/N0 sweep on one seed) stay together -- group q runs on rank q % ndev, in one
 *                     pass over its streams as on one device.  The form of BASELINE configs[4]: 11 points x 8 seeds given
 *                     as 8 groups of 11 trials (the seeds as stretches of the one cycle: warmup = 16 + (s << 48)) is one
 *                     11-point sweep per device on eight devices and eight sweeps back to back on one; the totals do not
 *                     depend on ndev. */
#define BBB_SHARD_TRIALS 0
#define BBB_SHARD_SEEDS 1
#define BBB_SHARD_BITS 2
#define BBB_SHARD_GROUPS 3
int bbb_ber_sweep_multi(bbb_lutopt *const *handles, int ndev, const bbb_trial_cfg *cfgs, int ncfg, int mode,
                        bbb_ber *out);
/* The share of `rank` among `ndev`: mine[i] is trial i as that rank runs it (nbits = 0: not at all). */
int bbb_sweep_shard(const bbb_trial_cfg *cfgs, int ncfg, int ndev, int rank, int mode, bbb_trial_cfg *mine);
/* What the last bbb_ber_sweep_multi of this process ran on, for a caller (or a first multi-GPU run) that wants to check
 * itself: the number of handles it was given, the rank count the RCCL communicator reports (ncclCommCount), the file the
 * RCCL entry points were resolved from and whether that library was already mapped in the process (a PyTorch process holds
 * its own copy: it is reused, not loaded a second time). */
typedef struct {
    int32_t n_devices;      /* handles of the call */
    int32_t n_ranks_seen;   /* ncclCommCount of its communicator (0: no call yet) */
    int32_t rccl_reused;    /* 1: found with RTLD_NOLOAD */
    int32_t reserved;
    char    rccl_path[256];
} bbb_multi_info;
int bbb_multi_last_info(bbb_multi_info *out);
/* Destroy the cached RCCL communicators and the library's pooled internal streams (optional; before unloading the
 * library or resetting devices: destroy every handle first -- handles created afterwards get fresh streams). */
int bbb_multi_release(void);

/* ---- pulse shaper and transmitter output (8 samples per data bit) -------------------------- */

/* PRBSShaper (gateware/bbb/bitshaper.py:12-86) and TX (gateware/bbb/tx.py:33-81).
 * Sample n of the shaper is  sum_{idx<8} (bit[M-idx] ? +1 : -1) * coeffs[8*idx + ph]  (12-bit
 * signed), M = floor((n-17)/8), ph = (n-17) mod 8: the 64-tap pulse applied to +-1 impulses at the
 * middle of each 8-sample bit period with the 13-sample pipeline delay the reference's own test
 * compensates (bitshaper.py:143-155); data bits before the first one count as 0 (reset shift
 * register).  From sample 73 on this equals scipy.signal.lfilter(coeffs, [1], impulses)[n-13].
 * TX.x = wrap12(bit_en * shaped + noise_en * wrap12(g * noise_var)), g = CLT sample of LUTOPT
 * state A^(warmup + n + 1) init (tx.py:70-81); the relative alignment of noise and bits is
 * build-defined. */
typedef struct {
    int16_t  coeffs[64];   /* one coefficient set, each in (-256, 255] (bitshaper.py:19-21) */
    int32_t  source;       /* 0 = PRBS (tx.py src_sel 0), 1 = Pulser: one 1 every 256 bits (tx.py:20-30) */
    int32_t  prbs_k;
    uint64_t prbs_state;
    int32_t  bit_en, noise_en, noise_var, reserved;   /* tx.py:39-52 */
    uint64_t warmup;       /* LUTOPT clocks before the first noise sample */
} bbb_tx_cfg;

/* PRBSShaper.x: samples first_sample .. first_sample+nsamples-1 as int16 (only cfg->coeffs,
 * source, prbs_k, prbs_state are used).  out_dev must be 16-byte aligned. */
int bbb_shaper_fill_i16(const bbb_tx_cfg *cfg, int16_t *out_dev, uint64_t nsamples, uint64_t first_sample,
                        int device, void *hip_stream);
/* TX.x on the handle's generator and stream. */
int bbb_tx_fill_i16(bbb_lutopt *h, const bbb_tx_cfg *cfg, int16_t *out_dev, uint64_t nsamples,
                    uint64_t first_sample);
/* TX.x as a sequential stream (tx.py:39-81: one sample per clock), the counterpart of bbb_awgn_stream_* for the waveform:
 * open keeps a copy of *cfg, turns the two-kernel form on for the handle (level 2 of bbb_lutopt_set_staged: one noise
 * kernel per two calls, a shaping mover per call -- unless the caller had chosen a level) and announces the first call;
 * next delivers nsamples_per_call samples at the stream position and announces the call after it; read is next with
 * another length (a ragged tail); seek moves the position (what waits ahead is dropped); close restores the handle's
 * staging level.  Every call is bbb_tx_fill_i16 on the handle's stream: same samples, same errors.  One open stream per
 * handle (noise or transmitter). */
typedef struct bbb_tx_stream bbb_tx_stream;
int bbb_tx_stream_open(bbb_lutopt *h, const bbb_tx_cfg *cfg, uint64_t nsamples_per_call, uint64_t first_sample, bbb_tx_stream **s);
int bbb_tx_stream_next(bbb_tx_stream *s, int16_t *out_dev);
int bbb_tx_stream_read(bbb_tx_stream *s, int16_t *out_dev, uint64_t nsamples);
int bbb_tx_stream_seek(bbb_tx_stream *s, uint64_t first_sample);
int bbb_tx_stream_tell(const bbb_tx_stream *s, uint64_t *next_sample);
int bbb_tx_stream_close(bbb_tx_stream *s);

/* Receiver front end: sign slicer (gateware/bbb/rx.py:29: bit = sample >= 0), sampling phase
 * (the BitDelayLine of rx.py:32-33, delayline.py:45-66) and clock division (rx.py:35-43); with
 * strict != 0 the threshold of software/memdump/decode.py:15-16 (sample > 0, there with stride 4).
 * Bit j = decide(samples_dev[phase + j*stride]) for every j with phase + j*stride < nsamples,
 * packed LSB first into u64 words (the layout bbb_prbs_check reads); *nbits_out (host, may be
 * NULL) receives the number of bits.  bits_packed_dev needs ceil(nbits/64) words. */
int bbb_rx_slice(const int16_t *samples_dev, uint64_t nsamples, uint64_t stride, uint64_t phase, int strict,
                 uint64_t *bits_packed_dev, uint64_t *nbits_out, int device, void *hip_stream);

/* The reference's sampling-phase knob (`sample_delay`, rx.py:19,32-33: 0..samples_per_bit) tried at
 * every setting: for phase p in [0, nphases) slice bit j = decide(samples_dev[p + j*stride]) and run
 * the exact detector over the resulting stream; stats_out[p] (host) receives the totals.  The best
 * phase is the one with the fewest `errors`; nothing here chooses for the caller. */
int bbb_rx_phase_search(const int16_t *samples_dev, uint64_t nsamples, uint64_t stride, uint64_t nphases, int strict,
                        int k, bbb_detector_stats *stats_out, int device, void *hip_stream);

/* ---- GF(2) helpers (host only; the pieces of software/rnghunt this path leans on) ------------ */

/* Berlekamp-Massey (software/rnghunt/src/berlekamp_massey.rs:5-31): minimal polynomial of the bit
 * sequence bits[0..n) (one bit per byte).  coeffs_out[i] (needs n+1 bytes) = coefficient of x^i,
 * *degree = linear complexity.  E.g. the first 19 bits of PRBS9 give x^9 + x^5 + 1 (its test,
 * berlekamp_massey.rs:40-42). */
int bbb_gf2_berlekamp_massey(const uint8_t *bits, uint64_t n, uint8_t *coeffs_out, int64_t *degree);
/* BinaryMatrix::recur (software/rnghunt/src/binary_matrix.rs:68-76) on rnghunt's matrix storage
 * (column-major u64 words, first row in the MSbit, binary_matrix.rs:15-19): out_bits[s] = bit 0 of
 * A^(s+1) x. */
int bbb_gf2_recur(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits, int nsteps,
                  uint8_t *out_bits);

/* BinaryMatrix::dot (binary_matrix.rs:52-63), same storage: out_bits[r] = (A x)[r], nrows bytes. */
int bbb_gf2_dot(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits, uint8_t *out_bits);
/* BinaryPolynomial::is_primitive (software/rnghunt/src/binary_polynomial.rs:178-216).  coeffs[i] is
 * the coefficient of x^(ncoeffs-1-i), the order of BinaryPolynomial::from_coefficients (:48-53).
 * *result = 1 / 0.  The test needs the prime factors of 2^deg - 1: the degrees in
 * basebandboard_amd/data/mersenne_factors.txt are served, others give BBB_EUNSUP. */
int bbb_gf2_poly_is_primitive(const uint8_t *coeffs, int ncoeffs, int *result);
/* BinaryPolynomial::modexp (:135-163): x^e mod p, e = little-endian 64-bit words; out_coeffs has
 * ncoeffs bytes in the same order as coeffs. */
int bbb_gf2_poly_modexp(const uint8_t *coeffs, int ncoeffs, const uint64_t *exponent_words, int nwords,
                        uint8_t *out_coeffs);
/* The polynomial rnghunt's search examines for a recurrence (src/bin/rnghunt.rs:27-38): bit 0 of 2k
 * successive states from the all-ones vector, reversed, through Berlekamp-Massey.  coeffs_out needs
 * 2k+1 bytes (the reversed sequence of a singular matrix can have complexity above k); entries
 * 0..*degree are the coefficients, highest power first. */
int bbb_lutopt_charpoly(int k, const uint16_t *taps, const uint32_t *row_off, uint8_t *coeffs_out, int *degree);
/* Its acceptance test (rnghunt.rs:40-46): degree == k and primitive, i.e. period 2^k - 1. */
int bbb_lutopt_is_full_period(int k, const uint16_t *taps, const uint32_t *row_off, int *result);
/* Its output file (rnghunt.rs:51-53): k lines of k characters, line r character c = A[r][c]; read
 * back by bbb_lutopt_load_matrix_file and by the reference's util/pack.py, util/verify.py. */
int bbb_lutopt_save_matrix_file(const char *path, int k, const uint16_t *taps, const uint32_t *row_off);

/* ---- the recurrence search of software/rnghunt, on the GPU ----------------------------------- */

/* Candidate number `candidate` of `seed`: a random k x k matrix with 3 or 4 ones per row (weights
 * drawn from [3,4,4,4,4,4,4,4], rnghunt.rs:25) and balanced column weights
 * (BinaryMatrix::random, binary_matrix.rs:81-101), from a counter-based generator so that it is the
 * same matrix on every host and device (the reference's thread_rng is unseeded; see
 * csrc/search_rng.hpp for the construction).  taps_out needs 4k entries, row_off_out k+1. */
int bbb_lutopt_search_candidate(int k, uint64_t seed, uint64_t candidate, uint16_t *taps_out, uint32_t *row_off_out);
typedef struct {
    uint64_t tested;          /* candidates examined */
    uint64_t full_degree;     /* of those: characteristic polynomial of degree k (rnghunt.rs:40) */
    uint64_t order_divides;   /* of those: x^(2^k - 1) = 1 (first check of is_primitive) */
    uint64_t primitive;       /* of those: primitive = accepted */
    uint64_t kernel_ns;       /* duration of the search kernel (HIP events); the call adds the host re-check of a hit */
} bbb_search_stats;
/* Examine candidates first_candidate .. first_candidate + ncandidates - 1 on the GPU, one per
 * wavefront at a time: build the matrix, run 2k steps from the all-ones state, Berlekamp-Massey,
 * degree check, primitivity test (the loop body of rnghunt.rs:23-47).  *found_index receives the
 * SMALLEST accepted candidate number (UINT64_MAX if none), its taps go to taps_out / row_off_out (host,
 * may be NULL) after a re-check with the host arithmetic above.  k must be a multiple of 64 in
 * 64..512 or 16 or 32, with an entry in the factor table.  *stats is a host result. */
int bbb_lutopt_search(int k, uint64_t seed, uint64_t first_candidate, uint64_t ncandidates, uint64_t *found_index,
                      uint16_t *taps_out, uint32_t *row_off_out, bbb_search_stats *stats, int device, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BBB_H */
