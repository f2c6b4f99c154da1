"""BPSK-over-PRBS through CLT AWGN, sliced and counted -- the Monte-Carlo loop built from the
reference's TX noise path (gateware/bbb/tx.py:70-81) and RX slicer (gateware/bbb/rx.py:29).

The reference has no error counter and no Eb/N0 notion (prbs.py:79 raises an unconsumed pulse;
the noise level is the 4-bit `noise_var` multiplier, tx.py:52): the counters, the Eb/N0 <->
(amp, noise_var) mapping and the sharding of trials over GPUs are defined here.
"""
import ctypes as C
import math

import torch

from . import _lib
from .prbs import TAPS

SIGMA_G = 8.0          # CLTGRNG standard deviation for n = 256: sqrt(2**(8-2)) (rng.py:63-65)


def ebn0_db(amp, noise_var):
    """Eb/N0 realised by r = +-amp + noise_var * g with Var[g] = 64: amp^2 / (2 sigma^2)."""
    if noise_var == 0:
        return math.inf
    return 10.0 * math.log10(amp * amp / (2.0 * (SIGMA_G * noise_var) ** 2))


def amp_for_ebn0(db, noise_var):
    """Nearest integer BPSK level for a target Eb/N0 at the given noise multiplier."""
    return int(round(SIGMA_G * noise_var * math.sqrt(2.0 * 10.0 ** (db / 10.0))))


def ber_theory(db):
    """Q(sqrt(2 Eb/N0)) -- the Gaussian-channel sanity value (the CLT tails are lighter)."""
    return 0.5 * math.erfc(math.sqrt(10.0 ** (db / 10.0)))


def ber_lattice(amp, noise_var):
    """What a Gaussian of variance 64 predicts for THIS integer channel: the CLT sample g is an integer, so bit 0
    errs when g >= ceil(amp / nv) and bit 1 when g <= -(floor(amp / nv) + 1) (tx.py:75-81, rx.py:29 without
    12-bit wrap); with the continuity correction of half a lattice step that is
    (Q((k0 - 1/2) / 8) + Q((k1 - 1/2) / 8)) / 2.  Comparable with the counters, unlike Q(sqrt(2 Eb/N0)) of the
    nominal label, which ignores the lattice (the two differ by -7 % ... +12 % over 0-10 dB)."""
    if noise_var == 0:
        return 0.0
    k0 = -(-amp // noise_var)
    k1 = amp // noise_var + 1
    q = lambda x: 0.5 * math.erfc(x / math.sqrt(2.0))
    return 0.5 * (q((k0 - 0.5) / SIGMA_G) + q((k1 - 0.5) / SIGMA_G))


def ebn0_db_effective(amp, noise_var):
    """The Eb/N0 at which Q(sqrt(2 Eb/N0)) equals ber_lattice(amp, noise_var): the label to plot the counters
    against a theoretical BPSK curve."""
    target = ber_lattice(amp, noise_var)
    if target <= 0.0:
        return math.inf
    lo, hi = -30.0, 40.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if ber_theory(mid) > target:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


class Trial:
    """One (Eb/N0 point, seed-offset) trial; field meaning as bbb_trial_cfg in include/bbb.h."""

    def __init__(self, nbits, amp, noise_var, prbs_k=31, prbs_state=1, warmup=16, first_bit=0):
        if prbs_k not in TAPS:
            raise ValueError("k={} invalid for PRBS".format(prbs_k))
        self.nbits, self.amp, self.noise_var = int(nbits), int(amp), int(noise_var)
        self.prbs_k, self.prbs_state = int(prbs_k), int(prbs_state)
        self.warmup, self.first_bit = int(warmup), int(first_bit)

    def as_c(self):
        return _lib.TrialCfg(self.prbs_k, self.amp, self.noise_var, 0, self.prbs_state, self.warmup,
                             self.first_bit, self.nbits)


class PreparedTrials:
    """A trial list already marshalled for the C ABI (`prepare`): the ctypes array of bbb_trial_cfg and its length.  Building it costs
    ~2 us per trial in Python -- 0.2 ms for BASELINE configs[4]'s 88 trials, a sixth of the sweep it describes -- so a caller that
    times a call, or repeats one, prepares its list once and passes this instead of the list."""

    def __init__(self, trials):
        self.n = len(trials)
        self.cfgs = (_lib.TrialCfg * max(self.n, 1))(*[t.as_c() for t in trials])

    def __len__(self):
        return self.n


def prepare(trials):
    return trials if isinstance(trials, PreparedTrials) else PreparedTrials(trials)


def run_trials(urng, trials):
    """Run trials (a list of Trial, or `prepare`d) on `urng`'s GPU; returns a list of (bits, errors)."""
    p = prepare(trials)
    if not p.n:
        return []
    out = (_lib.Ber * p.n)()
    urng._bind_stream()
    _lib.check(_lib.lib().bbb_ber_trials(urng._h, p.cfgs, p.n, out), "bbb_ber_trials")
    return [(o.bits, o.errors) for o in out]


def run_trials_into(urng, trials, counters):
    """Accumulate into an int64 CUDA tensor [len(trials), 2] without synchronising (the buffer a
    multi-GPU sweep all-reduces)."""
    p = prepare(trials)
    if counters.dtype != torch.int64 or not counters.is_cuda or not counters.is_contiguous() \
            or counters.numel() < 2 * p.n:
        raise ValueError("counters must be a contiguous int64 CUDA tensor with 2 words per trial")
    if not p.n:
        return counters
    urng._bind_stream()
    _lib.check(_lib.lib().bbb_ber_trials_dev(urng._h, p.cfgs, p.n, C.c_void_p(counters.data_ptr())),
               "bbb_ber_trials_dev")
    return counters


class ContinuedTrials:
    """A trial group continued over several calls (bbb_ber_run_*): `trials` share PRBS, offsets and nbits (= bits per call)
    and read ONE noise stream; a block of `calls_per_block` calls shares one seeding of the generators, and after every
    `calls_per_block`-th call the totals equal `run_trials` over that block's bits, bit for bit.  The Monte-Carlo loop
    that adds bits until it has seen enough errors:

        with ContinuedTrials(urng, trials, 8) as run:
            while min(e for _, e in run.next()) < 100: pass
    """

    def __init__(self, urng, trials, calls_per_block):
        if not trials:
            raise ValueError("no trials")
        self.urng, self.ntrials = urng, len(trials)
        cfgs = (_lib.TrialCfg * len(trials))(*[t.as_c() for t in trials])
        self._r = C.c_void_p()
        urng._bind_stream()
        _lib.check(_lib.lib().bbb_ber_run_open(urng._h, cfgs, len(trials), int(calls_per_block), C.byref(self._r)), "bbb_ber_run_open")

    def next(self, read=True):
        """One more call; returns the run's totals [(bits, errors)] so far (read=False: queued only, returns None)."""
        self.urng._bind_stream()
        out = (_lib.Ber * self.ntrials)() if read else None
        _lib.check(_lib.lib().bbb_ber_run_next(self._r, out), "bbb_ber_run_next")
        return [(o.bits, o.errors) for o in out] if read else None

    def next_into(self, counters):
        """One more call, its counters ADDED to an int64 CUDA tensor [ntrials, 2], without synchronising."""
        if counters.dtype != torch.int64 or not counters.is_cuda or not counters.is_contiguous() or counters.numel() < 2 * self.ntrials:
            raise ValueError("counters must be a contiguous int64 CUDA tensor with 2 words per trial")
        self.urng._bind_stream()
        _lib.check(_lib.lib().bbb_ber_run_next_dev(self._r, C.c_void_p(counters.data_ptr())), "bbb_ber_run_next_dev")
        return counters

    def tell(self):
        """(calls made so far, first bit of the next block to start)"""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().bbb_ber_run_tell(self._r, C.byref(a), C.byref(b)), "bbb_ber_run_tell")
        return a.value, b.value

    def close(self):
        if self._r:
            _lib.lib().bbb_ber_run_close(self._r)
            self._r = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- sharding a sweep over the GPUs of this process: the C ABI's own collective ----------------

def shard_trials(trials, rank, world, mode=_lib.SHARD_TRIALS):
    """What rank `rank` of `world` runs of `trials` under a sharding mode (bbb_sweep_shard; host arithmetic, no
    GPU needed): a list as long as `trials`, entries with nbits = 0 are not run by that rank."""
    n = len(trials)
    cfgs = (_lib.TrialCfg * max(n, 1))(*[t.as_c() for t in trials])
    mine = (_lib.TrialCfg * max(n, 1))()
    _lib.check(_lib.lib().bbb_sweep_shard(cfgs, n, world, rank, mode, mine), "bbb_sweep_shard")
    return [Trial(nbits=m.nbits, amp=m.amp, noise_var=m.noise_var, prbs_k=m.prbs_k, prbs_state=m.prbs_state,
                  warmup=m.warmup, first_bit=m.first_bit) for m in mine[:n]]


def sweep_multi(urngs, trials, mode=_lib.SHARD_BITS):
    """One process, several GPUs: `urngs[r]` is a LUTOPT on device r.  bbb_ber_sweep_multi runs every device's
    share on its own host thread and sums the counters with ONE RCCL all-reduce (uint64, sum).  Returns a list
    of (bits, errors)."""
    p = prepare(trials)
    if not p.n:
        return []
    out = (_lib.Ber * p.n)()
    hs = (C.c_void_p * len(urngs))(*[u._h for u in urngs])
    for u in urngs:
        u._bind_stream()
    _lib.check(_lib.lib().bbb_ber_sweep_multi(hs, len(urngs), p.cfgs, p.n, mode, out), "bbb_ber_sweep_multi")
    return [(o.bits, o.errors) for o in out]


def multi_info():
    """What the last sweep_multi of this process ran on (bbb_multi_last_info): devices, the communicator's rank count, the
    RCCL file in use and whether it was the copy the process already held."""
    m = _lib.MultiInfo()
    _lib.check(_lib.lib().bbb_multi_last_info(C.byref(m)), "bbb_multi_last_info")
    return {"n_devices": m.n_devices, "n_ranks_seen": m.n_ranks_seen, "rccl_reused": bool(m.rccl_reused),
            "rccl_path": m.rccl_path.decode(errors="replace")}


# ---- sharding a sweep over ranks (one process per GPU) -----------------------------------------

def shard(ntrials, rank, world):
    """Static round-robin: trial i runs on rank i % world."""
    return list(range(rank, ntrials, world))


def sweep(trials, runner, rank=0, world=1, group=None):
    """Run this rank's share of `trials` and sum the 64-bit counters over ranks with ONE all-reduce.

    `runner(local_trials, counters_view)` must add (bits, errors) of each local trial into the rows
    of `counters_view` ([len(local), 2] int64, on the device the process group reduces on); on a
    GPU rank it is `lambda ts, c: run_trials_into(urng, ts, c)`.  Integer sums are order
    independent, so the result equals the single-rank result exactly.
    Returns an int64 tensor [len(trials), 2] holding the global counters on every rank.
    """
    import torch.distributed as dist
    mine = shard(len(trials), rank, world)
    local = runner([trials[i] for i in mine], len(mine))
    total = torch.zeros((len(trials), 2), dtype=torch.int64, device=local.device)
    if mine:
        total[torch.tensor(mine, device=local.device)] = local
    if world > 1:
        if dist.get_backend(group) == "gloo" and total.is_cuda:      # CPU rehearsal of a GPU sweep
            host = total.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            total.copy_(host)
        else:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def sweep_bits(trials, runner, rank=0, world=1, group=None):
    """Third form (BBB_SHARD_BITS): every rank runs all `trials` over ITS slice of each trial's bit range, same
    reset state everywhere, ONE all-reduce.  The totals equal the single-rank counters of `trials` exactly, and
    an Eb/N0 sweep keeps its one-pass-per-noise-stream grouping on every rank."""
    import torch.distributed as dist
    mine = shard_trials(trials, rank, world, _lib.SHARD_BITS)
    total = runner(mine, len(mine))
    if world > 1:
        if dist.get_backend(group) == "gloo" and total.is_cuda:      # CPU rehearsal of a GPU sweep
            host = total.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            total.copy_(host)
        else:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def sweep_seeds(trials, runner, world=1, group=None):
    """The other way to shard a Monte-Carlo sweep (points x seeds): EVERY rank
    runs all `trials` on its own generator (a different `init` per rank), so that each rank makes one
    pass over its noise stream for the whole sweep, and the counters of the ranks are summed with ONE
    all-reduce: `world` times the bits per point in the time of one sweep.  Returns int64 [len(trials), 2]."""
    import torch.distributed as dist
    total = runner(list(trials), len(trials))
    if world > 1:
        if dist.get_backend(group) == "gloo" and total.is_cuda:      # CPU rehearsal of a GPU sweep
            host = total.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            total.copy_(host)
        else:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def gpu_runner(urng):
    """The `runner` for `sweep` on a GPU rank.  The counters of consecutive calls are slices of one zeroed block (32 calls'
    worth at a time): a zeroing kernel per call would sit on the stream between two trial kernels that otherwise follow each
    other directly.  The returned tensors are VIEWS into that block (all-reduced in place by the sweeps): copy what is to be
    kept beyond the runner.  The block belongs to the torch stream that was current when it was zeroed: a call under another
    current stream gets a fresh block (zeroed there) -- the old one stays alive as long as its views do."""
    pool = {"buf": None, "next": 0, "stream": None}

    def run(local_trials, n):
        cur = torch.cuda.current_stream(urng.device)
        if pool["buf"] is None or pool["buf"].shape[1] != n or pool["next"] == pool["buf"].shape[0] or pool["stream"] != cur:
            pool["buf"] = torch.zeros((32, n, 2), dtype=torch.int64, device=torch.device("cuda", urng.device))
            pool["next"] = 0
            pool["stream"] = cur
        c = pool["buf"][pool["next"]]
        pool["next"] += 1
        return run_trials_into(urng, local_trials, c)
    return run
