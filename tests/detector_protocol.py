"""The reference's PRBSErrorDetector test protocol (gateware/bbb/prbs.py:124-163), restated.

Reference design: Bernoulli(0.02) injected errors, a clean preamble of 2k bits, a burst of 3k
errors in the middle to force a resynchronisation, 2k clean bits after it; then, with the one-clock
latency of the registered input (prbs.py:66), `rx_errors[1:] == tx_errors[:-1]` wherever
`reload == 0` (prbs.py:153-163).

Two deliberate differences, both because the reference draws its errors from an UNSEEDED RNG and
its assertion only holds when no injected error lands inside a reload window (an error shifted
into the LFSR while `reload` is high leaves the detector locked to a wrong state, which reports
errors that were never injected -- the reference test fails on such draws too):
  * the draws here are seeded;
  * the clean windows are 4k instead of 2k bits, because for small k the initial reload (armed by
    the all-ones reset of `err_sr`, prbs.py:80) is re-armed by the residue of the reset LFSR state
    and outlasts 2k clocks (19 clocks for k = 7).

Testbench timing (migen simulator): a `yield sig.eq(v)` lands at the next clock edge together with
the synchronous updates, and reads see pre-edge values.  So the input wire is 0 during the first
clock and carries bit i-1 during clock i; `rx_errors[i]`, `reload[i]` are the outputs after
clock i.
"""
import numpy as np


def make_case(k, prbs_bits_fn, seed, literal=False):
    """Returns (wire, tx_errors): the 0/1 input wire per clock and the injected error flags.
    literal: the reference's own windows (prbs.py:133-138: 2k clean bits first, 2k after the burst) instead of 4k."""
    rng = np.random.default_rng(seed)
    nbits = min((1 << k) - 1, 512)
    tx_errors = rng.binomial(1, 0.02, nbits).astype(np.uint8)
    quiet = 2 * k if literal else 4 * k
    tx_errors[:quiet] = 0
    mid = nbits // 2
    tx_errors[mid: mid + 3 * k] = 1
    tx_errors[mid + 3 * k: mid + 3 * k + quiet] = 0
    tx = np.asarray(prbs_bits_fn(k, nbits), dtype=np.uint8)
    wire = np.concatenate([[0], (tx ^ tx_errors)[:-1]]).astype(np.uint8)
    return wire, tx_errors


def check_case(tx_errors, rx_errors, reload):
    """The reference's assertion (prbs.py:158-163)."""
    valid = (1 - np.asarray(reload)).astype(bool)[:-1]
    a = np.asarray(tx_errors)[:-1][valid].tolist()
    b = np.asarray(rx_errors)[1:][valid].tolist()
    assert a == b
    # the detector must actually have been in lock for a good part of the stream,
    # and must have seen (and flagged) the burst
    assert valid.sum() >= len(tx_errors) // 4
    assert (1 - valid.astype(int)).sum() > 0


def errors_inside_reload(k, tx_errors, reload):
    """Injected errors outside the burst that reach the detector while `reload` is high (error i is on the wire during
    clock i + 1).  Such an error is shifted INTO the LFSR: the detector comes out of the reload locked to a wrong state and
    reports errors that were never injected -- the reference's own assertion (prbs.py:158-163) fails on such a draw, which
    with its unseeded RNG and 2k windows happens now and then (an error right behind the 2k preamble meets the tail of the
    start-up reloads; an error behind the 2k post-burst window meets the resynchronisation)."""
    n = len(tx_errors)
    mid = n // 2
    r = np.asarray(reload)
    return [i for i in range(n - 1) if tx_errors[i] and not (mid <= i < mid + 3 * k) and r[i + 1]]
