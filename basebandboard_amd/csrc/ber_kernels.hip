// ber_kernels.hip -- fused Monte-Carlo trial: PRBS-k bit -> BPSK level + scaled CLT noise ->
// sign slicer -> compare -> count, entirely in bit-sliced form (no sample stream is written).
//
// Reference pieces being chained (paths relative to the reference checkout):
//   PRBS bit            gateware/bbb/prbs.py:32-35
//   LUTOPT + CLTGRNG    gateware/bbb/rng.py:38-40, 96-108
//   noise scaling, sum  gateware/bbb/tx.py:75-81   (12-bit signed registers)
//   slicer              gateware/bbb/rx.py:29      (bit^ = sample >= 0)
// The error/bit counters and the sharding are build-defined (the reference only raises an
// unconsumed `err` pulse: prbs.py:79, rx.py:46).
//
// The channel is a function of (sample, bit) only, so the host tabulates it: for each bit value
// the error indicator over T = sample + 128 (mod 256) is a parity of at most four threshold tests
// [T >= thr] (one for ordinary parameters; 12-bit wrap-around can add more).  On the GPU a
// threshold test on the 8 bit planes of T is 8 V_BITOP3 per 32 samples.
//
// Per wave: 2048 generators.  PRBS state is a second bit-sliced GF(2) recurrence (k planes, kept in
// lane-private LDS as a circular buffer).  Error counts: per-lane u32 -> wave reduction ->
// LDS-free single 64-bit atomic per wave (one wave per block).
#include "bbb_common.hpp"
#include "bitslice_util.hpp"
#include "awgn_launch.hpp"
#include "gen/lutopt256_gen.inc"

#include <cstdlib>

#ifndef BBB_BER_PART
#define BBB_BER_PART 0
#endif
#include "ber_kernels_impl.hpp"

