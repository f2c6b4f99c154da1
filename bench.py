#!/usr/bin/env python3
"""Headline benchmark: CLT Gaussian noise (AWGN) sample generation on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...,
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plainly as above -- then this process
  starts the N ranks itself as child processes before touching any GPU (launch_ranks) and passes rank 0's
  JSON line through.

One step = one pass of the hot path over one batch: 1e9 int8 CLT samples of the reference's
LUTOPT-256 -> CLTGRNG generator (BASELINE.json configs[1]; the reference has no "CLT-12 /
xorshift32" generator) written to HBM: one bbb_awgn_stream_next on the
rank's sample stream (the C ABI's sequential-stream object).  Rank r reads its own contiguous
stretch of the ONE sequential reference stream, 2^48 steps apart, a new part every step, so
nothing is cached between steps and ranks are independent shards (weak scaling, no data-path
collective).  Seeding (GF(2) jump-ahead on the GPU), sample kernels and movers are all inside
the timed region.  Before the W warm-up steps the untimed region also holds BENCH_RAMP_STEPS
(default 64) more of the same steps: this part's clock governor needs ~50 ms of load to leave
its idle state (profiles/r03_ramp_clock_per_launch.log: 1.75 GHz at step 2 after an idle second,
2.17 at step 10, 2.38 from step 40 on); `extra.cold_start` reports the K steps straight from idle.

The JSON line also carries
  roofline     achieved HBM-write GB/s of the sample kernel (algorithmic 1 B/sample / its mean
               launch time from hipEvents on the launch stream) against the 8 TB/s HBM peak.
               The kernel is integer-VALU bound (~1250 lane-ops per 32 samples), so this fraction
               is expected to be far below 1; `valu_*` fields give the bound that applies.
               `roofline_other` holds the same record for the PRBS fill / check, detector and TX kernels.
  cpu_baseline the CPU oracle (k=256 fast path, -march=native) timed on this host on a bounded
               sample (rank 0, N = 1 only): all host cores this process may use (count stated), and
               one core beside it.
  extra        PRBS-31 generate+check loopback (per pass), the exact detector, the TX waveform, the
               matrix search and the BER sweep, measured after the timed region.  The sweep's `gbit_s` is ONE
               isolated bbb_ber_trials call (its seeding, zeroing, read-back and synchronisation inside; median of
               five, hot GPU); `back_to_back_gbit_s` is per sweep over sixteen independent sweeps queued back to back
               (each with its own seeding); `ber_sweep_88` is BASELINE configs[4]'s 88 trials (11 points x 8 seeds) --
               on ONE device through bbb_ber_sweep_multi(BBB_SHARD_GROUPS) at N = 1 (with `projected_8_gpu` and its
               arithmetic), the ranks' shares of the same 88 trials + one all-reduce at N > 1 (strong scaling).
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NSAMP = 1_000_000_000
WARM_STATE = 16                  # 2*log2(256): the reference test's warm-up (rng.py:161-162)
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
VALU_OPS_PER_SAMPLE = None       # filled from the generated network's op count


def stream_clock_record():
    """profiles/r05_stream_clock.json: what the sample kernel's clock is inside the noise stream as shipped, without its staging stores
    and without the mover (round 5) -- None when the file is absent"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_stream_clock.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return {"source": "profiles/r05_stream_clock.json", **rec["variants"]}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline():
    """Time the oracle's k=256 fast path on this host (checker code used as the reported CPU baseline only).
    Bounded sample: every core this process may use generates 2e7 samples of the stream at its own offset
    (a few seconds), and one core alone for the single-core figure."""
    import concurrent.futures
    import oracle as O
    so = O.build(native=True, out="/tmp/libbbb_oracle_native.so", force=True)
    lib = O.lib(path=so)
    m = O.Lutopt(path=O.data_path(256), _lib=lib)
    n = 20_000_000
    t0 = time.perf_counter()
    m.awgn(1, WARM_STATE, n, fast=True)
    dt1 = time.perf_counter() - t0
    reps = max(1, min(4, int(6.0 / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    for i in range(reps):
        m.awgn(1 + i, WARM_STATE, n, fast=True)      # same length, different seed each repetition
    dt = time.perf_counter() - t0
    # all cores: one piece of the stream per thread (ctypes releases the GIL around the call).  Every thread writes into
    # its own buffer, allocated and touched BEFORE the clock starts: with fresh buffers inside the timed region the
    # threads spent their time in page faults behind the process's one address-space lock (0.28 Gsample/s on 256 threads)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None                                    # a container's CPU share (cgroup quota), where one is set
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            quota = parse(open(path).read().strip())
            if quota:
                break
        except Exception:
            pass
    if quota:
        ncores = max(1, min(ncores, int(quota + 0.999)))
    nthr = max(1, min(ncores, 512))
    per = 4 if nthr <= 64 else 2                    # pieces per thread
    nt = 10_000_000
    import numpy as np
    bufs = [np.zeros(nt, dtype=np.int8) for _ in range(nthr)]

    def work(i):
        for j in range(per):
            m.awgn(100 + i * per + j, WARM_STATE, nt, fast=True, out=bufs[i])

    with concurrent.futures.ThreadPoolExecutor(nthr) as ex:
        list(ex.map(lambda i: m.awgn(7, WARM_STATE, 100_000, fast=True, out=bufs[i]), range(nthr)))      # threads up, code paged in
        t1 = time.perf_counter()
        list(ex.map(work, range(nthr)))
        dtn = time.perf_counter() - t1
    n_all = nthr * per * nt
    return {"value": round(n_all / dtn / 1e9, 5), "unit": "Gsample/s", "cores": nthr, "kind": "port",
            "sample": f"{nthr * per} x {nt} samples of the same stream, {per} pieces per thread on {nthr} threads (all the cores this "
                      f"process may use: affinity {len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else '?'}, "
                      f"cgroup quota {quota}; the host reports {os.cpu_count()} logical cores); oracle k=256 byte-table path, "
                      f"gcc -O3 -march=native",
            "single_core_value": round(reps * n / dt / 1e9, 5), "single_core_sample": f"{reps} x {n} samples on one core"}


def cpu_baseline_other():
    """The oracle's restatements of configs[2] / configs[3] on one host core (bounded samples), for `extra`."""
    import oracle as O
    lib = O.lib(path="/tmp/libbbb_oracle_native.so")
    nb = 400_000_000
    t0 = time.perf_counter()
    O.prbs_packed(31, nb, fast=True, _lib=lib)
    tp = time.perf_counter() - t0
    m = O.Lutopt(path=O.data_path(256), _lib=lib)
    nbe = 2_000_000
    t0 = time.perf_counter()
    m.ber_trial(1, 31, 1, 128, 8, WARM_STATE, 0, nbe)
    tb = time.perf_counter() - t0
    return {"prbs31_fill_gbit_s": round(nb / tp / 1e9, 3), "ber_trial_mbit_s": round(nbe / tb / 1e6, 2), "cores": 1, "kind": "port",
            "sample": f"{nb} PRBS-31 bits (word-parallel restatement); one {nbe}-bit BPSK trial (LUTOPT-256 + CLT + channel)"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), pass rank 0's stdout (the one JSON line) through and return
    non-zero if any rank fails.  The shape of the reference's only multi-worker program -- workers plus one
    channel back to the parent (software/rnghunt/src/bin/rnghunt.rs:16-18,54-65).  The parent never
    initialises the GPU, and nothing is exec'ed over a process that has."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(pathlib.Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = dict(enumerate(procs))
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))      # the whole launch
    kill_at = None                                                                          # after a failure: grace period, then SIGKILL
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending.values():
                    q.terminate()           # exactly the PIDs started above
                kill_at = time.time() + 20
        if pending and rc == 0 and time.time() > deadline:
            rc = 124
            print("bench.py: launch timed out; stopping the ranks", file=sys.stderr)
            for q in pending.values():
                q.terminate()
            kill_at = time.time() + 20
        if pending and kill_at is not None and time.time() > kill_at:
            for q in pending.values():      # a rank stuck in a rendezvous or a GPU call ignores SIGTERM
                q.kill()
            kill_at = time.time() + 3600
        time.sleep(0.05)
    return rc


def gather_ranks(dt, dt_own, steps, world, dist, device):
    """The launch contract's reduction: MAX of the ranks' timed regions (barrier to barrier) -- plus what a first multi-GPU
    run needs to check itself: how many ranks the process group really holds and every rank's OWN time (its K steps up to
    its own synchronisation, before the closing barrier: a slow rank shows).  Returns (dt_max, info)."""
    import torch
    if world <= 1:
        return dt, {"n_ranks_seen": 1, "per_rank_ms_per_step": [round(dt_own / steps * 1e3, 4)]}
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    own = torch.tensor([dt_own], dtype=torch.float64, device=device)
    every = [torch.zeros_like(own) for _ in range(world)]
    dist.all_gather(every, own)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per = [float(x.item()) for x in every]
    return float(t.item()), {"n_ranks_seen": dist.get_world_size(), "per_rank_ms_per_step": [round(x / steps * 1e3, 4) for x in per]}


def dry_run(args, emit, rank, world):
    """BENCH_DRY_RUN=1: everything of a multi-rank run that is NOT GPU work -- the launch (launch_ranks or a launcher), the
    rendezvous, barriers, the MAX-reduce of the timed region, the counter all-reduces of the sweep and the shape of the JSON
    line -- on the CPU with gloo, the step replaced by a sleep and the trial runner by a stub with known counters.  For
    tests/test_bench_launch.py: the first real N > 1 run should meet no surprise outside the kernels.  The line says
    "dry_run": true and carries no measurement."""
    import torch
    import torch.distributed as dist
    from basebandboard_amd import channel
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))              # (rank r is r + 1 times slower: the MAX must pick the last rank)
    dt_own = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    dt, info = gather_ranks(dt, dt_own, args.steps, world, dist, "cpu")
    # the sweep's collectives with a stub runner: trial i on "seed" r counts (1000, 10 i + r)
    trials = [channel.Trial(nbits=1000, amp=channel.amp_for_ebn0(db, 8), noise_var=8) for db in range(11)]
    stub = lambda ts, n: torch.tensor([[t.nbits, 10 * i + rank] for i, t in enumerate(ts)], dtype=torch.int64)
    seeds_total = channel.sweep_seeds(trials, stub, world=world).tolist()
    bits_total = channel.sweep_bits(trials, lambda ts, n: torch.tensor([[t.nbits, 1] for t in ts], dtype=torch.int64), rank=rank, world=world).tolist()
    if rank == 0:
        emit({"metric": "awgn_clt_gsamples_per_s", "value": None, "unit": "Gsample/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "u32 bit-sliced GF(2) / int8 out", "data": "synthetic", "dry_run": True,
              "config": {"workload": "DRY RUN: no GPU work, steps are sleeps of (rank + 1) ms", "samples_per_step_per_gpu": NSAMP},
              **info,
              "extra": {"ber_sweep": {"seeds": world, "counters": seeds_total,
                                      "reduce": "torch.distributed.all_reduce(int64[11,2], SUM), gloo" if world > 1 else "single rank"},
                        "ber_sweep_bits_sharded": {"counters": bits_total}}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # (0.2 s of device time; over 20 steps the pipeline's fill and drain are 5 % of the region)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU): every rank joins, one all-reduce, rank 0 prints a JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: this process becomes the launcher and never touches
        # the GPU (no torch.cuda / HIP call has happened yet); the ranks are its children.
        sys.exit(launch_ranks(args.gpus))

    # stdout carries ONE line, the JSON: libraries that print there on their own (RCCL's version banner under
    # NCCL_DEBUG=VERSION, which some images export) go to stderr for the whole run -- at the descriptor level, so that
    # C code is covered -- and the result is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if args.launch_check:
        if os.environ.get("BENCH_FAIL_RANK") == str(rank):
            sys.exit(3)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            emit({"launch_check": world, "sum": int(t.item())})
        dist.barrier()
        dist.destroy_process_group()
        return
    if os.environ.get("BENCH_DRY_RUN"):
        return dry_run(args, emit, rank, world)
    # BENCH_BACKEND=gloo + BENCH_SHARE_GPU=1 rehearse the multi-rank code path on a one-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import basebandboard_amd as bbb
    from basebandboard_amd import channel
    u = bbb.LUTOPT.shipped(256, init=1, device=local_rank)
    assert u.specialised, "bench must run the generated gfx950 kernel"
    # The timed loop drains the rank's sample stream through the C ABI's stream object (bbb_awgn_stream_open / _next):
    # the two-kernel form (sample kernel -> count planes in a staging slot, mover -> bytes beside the NEXT kernel), two reads
    # per sample kernel, every next read announced by the library.  BENCH_ONE_KERNEL=1 times plain bbb_awgn_fill_i8 calls
    # in the one-kernel form instead (also reported in `extra`); BENCH_LEVEL=1 / 4 ... picks another level of
    # bbb_lutopt_set_staged for the stream (1: one read per sample kernel).
    staged = not os.environ.get("BENCH_ONE_KERNEL")
    level = int(os.environ.get("BENCH_LEVEL", "0")) if staged else 0
    if level:
        u.set_staged(True, look_ahead=level if level >= 2 else False)
    g = bbb.CLTGRNG(u)
    buf = torch.empty(NSAMP, dtype=torch.int8, device=f"cuda:{local_rank}")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every rank reads its own contiguous stretch of the one stream, 2^48 steps apart (jump-ahead makes any start
    # position as cheap as any other); within a rank the steps follow each other
    first0 = WARM_STATE + (rank << 48)
    first_step = lambda step: first0 + step * NSAMP
    prefetch = not os.environ.get("BENCH_NO_PREFETCH")
    st = g.stream(NSAMP, first_step=first0) if staged else None
    pos = [0]

    def one_step():
        if st is not None:
            st.next(out=buf)
        else:
            g.generate(NSAMP, first_step=first_step(pos[0]), out=buf)
            if prefetch:
                g.prefetch(NSAMP, first_step=first_step(pos[0] + 1))
        pos[0] += 1

    def drop_ahead():
        # the timed region must not inherit arithmetic from the untimed one: a waiting second half of a sample kernel's
        # output is dropped, so that the first timed step launches a kernel (with an odd --steps the last kernel's second
        # half is produced inside the timed region and unused)
        if st is not None:
            st.seek(st.tell())

    # step 0: kept for the parity check (rank 0) -- a prefix and the tail are copied aside on the device now and compared
    # with the oracle AFTER the timed region, so that the GPU does not idle in front of it
    one_step()
    head_d, tail_d = buf[:1_000_000].clone(), buf[NSAMP - 4096:].clone()
    torch.cuda.synchronize()
    # cold start: the K steps as a process finds them when it comes out of an idle GPU (for the record: `extra.cold_start`)
    drop_ahead()
    barrier()
    cold_steps = min(args.steps, 20)            # (the ramp is over after ~40 steps: a longer region would no longer be "cold")
    tc = time.perf_counter()
    for s in range(cold_steps):
        one_step()
    barrier()
    cold_ms = (time.perf_counter() - tc) / cold_steps * 1e3
    # clock ramp: untimed steps until the governor has left its idle state, then the W warm-up steps
    ramp_steps = int(os.environ.get("BENCH_RAMP_STEPS", "64"))
    for s in range(ramp_steps + args.warmup):
        one_step()
    drop_ahead()
    u.profile(True)
    u.profile_read(reset=True); u.profile_read_mover(reset=True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0               # this rank's own K steps (for the record; the metric is barrier to barrier)
    barrier()
    dt = time.perf_counter() - t0
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    mover_ms, movers = u.profile_read_mover(reset=True)
    u.profile(False)
    if st is not None:
        st.close()
    look_ahead = level if level >= 2 else (2 if (staged and not level) else 0)
    dt, rank_info = gather_ranks(dt, dt_own, args.steps, world, dist, f"cuda:{local_rank}" if backend == "nccl" else "cpu")
    verified = None
    if rank == 0:
        import numpy as np
        import oracle as O
        m = O.Lutopt(path=O.data_path(256))
        verified = bool(np.array_equal(head_d.cpu().numpy(), m.awgn(1, WARM_STATE, 1_000_000, fast=True)))
        verified = verified and bool(np.array_equal(tail_d.cpu().numpy(),
                                                    m.awgn(u.state_at(WARM_STATE + NSAMP - 4096), 0, 4096, fast=True)))
    del head_d, tail_d

    value = world * args.steps * NSAMP / dt / 1e9
    kern_avg_ms = kern_ms / max(calls, 1)                 # per LAUNCH of the sample kernel
    per_launch = look_ahead * NSAMP if look_ahead else NSAMP       # samples (= algorithmic bytes) one launch produces
    achieved = per_launch / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0

    # attainable write ceiling on this box: a plain streaming fill of the same 1e9 bytes
    fill_gbs = None
    if rank == 0:
        fb = torch.empty(NSAMP, dtype=torch.int8, device=f"cuda:{local_rank}")
        fb.fill_(1)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        f0.record()
        for _ in range(5):
            fb.fill_(1)
        f1.record()
        torch.cuda.synchronize()
        fill_gbs = 5 * NSAMP / f0.elapsed_time(f1) / 1e6
        del fb

    # HBM bytes and issued instructions from the PMC counters: they need their own rocprofv3 passes (profiles/README.md:
    # --pmc WRITE_SIZE, --pmc FETCH_SIZE, an SQ pass; counter collection serialises the kernels), so the line carries the
    # committed summary of this round's passes and says which file it came from (stale if the kernels changed since)
    traffic = traffic_src = traffic_step = None
    pmc_rec = {}
    for name in ("r05_awgn_pmc.json", "r04_awgn_pmc.json", "r03_awgn_pmc.json"):
        pmc = ROOT / "profiles" / name
        if pmc.exists() and staged:
            try:
                rec = json.load(open(pmc))
                if int(rec.get("samples_per_launch", 0)) != per_launch:
                    continue                      # measured on launches of another size (another level)
                pmc_rec = rec
                traffic = rec["sample_kernel"]["hbm_bytes_per_launch"]["total"]
                traffic_step = rec.get("traffic_per_read")
                traffic_src = f"profiles/{name} (separate --pmc passes over bench.py; kernels run alone under counter collection; stale if they changed since)"
                break
            except Exception:
                traffic = None
    ops_per_step = 1002
    try:
        inc = (ROOT / "basebandboard_amd" / "csrc" / "gen" / "lutopt256_gen.inc").read_text()
        import re
        ops_per_step = int(re.search(r"// (\d+) VALU ops per step", inc).group(1))
    except Exception:
        pass

    extra = {}
    other = []            # roofline records of the other kernels on the path
    if not args.no_extra:
        import ctypes as C
        from basebandboard_amd import _lib
        L = _lib.lib()
        dev = f"cuda:{local_rank}"
        sp = C.c_void_p(torch.cuda.current_stream(local_rank).cuda_stream)
        ev = lambda: torch.cuda.Event(enable_timing=True)

        def hbm(kernel, nbytes, ms, what):
            gbs = nbytes / ms / 1e6
            return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(nbytes), "kernel_ms_avg": round(ms, 4),
                    "what": what}

        # the other forms of the same stream, outside the timed region (hot GPU, 20 steps each): plain bbb_awgn_fill_i8 +
        # bbb_awgn_prefetch calls in the one-kernel form and in the two-kernel form at level 1 (one read per sample kernel)
        def other_form(level_):
            u2 = bbb.LUTOPT.shipped(256, init=1, device=local_rank)
            u2.set_staged(level_ > 0)
            g2 = bbb.CLTGRNG(u2)
            for s_ in range(8):
                g2.generate(NSAMP, first_step=first_step(s_), out=buf); g2.prefetch(NSAMP, first_step=first_step(s_ + 1))
            u2.profile(True); u2.profile_read(reset=True)
            torch.cuda.synchronize(); t_o = time.perf_counter()
            for s_ in range(8, 28):
                g2.generate(NSAMP, first_step=first_step(s_), out=buf); g2.prefetch(NSAMP, first_step=first_step(s_ + 1))
            torch.cuda.synchronize(); t_o = (time.perf_counter() - t_o) / 20
            _, k_o, c_o = u2.profile_read(reset=True)
            return {"ms_per_step": round(t_o * 1e3, 4), "gsample_s": round(NSAMP / t_o / 1e9, 1), "sample_kernel_ms_avg": round(k_o / max(c_o, 1), 4)}
        extra["awgn_other_forms"] = {"one_kernel (bbb_awgn_fill_i8 + bbb_awgn_prefetch, staged off)": other_form(0),
                                     "two_kernels_level_1 (bbb_lutopt_set_staged(h, 1): one read per sample kernel)": other_form(1)}
        # PRBS-31 loopback (BASELINE configs[2]): 1e10 bits written, then read back and checked.  Per pass, device
        # time between hipEvents on the launch stream: the fill, the check right after the fill (the loopback order:
        # it also pays for the write-backs of the fill's last 256 MiB, which are still dirty in the memory-side cache),
        # the check repeated on the now clean buffer; and the loopback cut into pieces that fit that cache.
        nbits = 10_000_000_000
        nbytes = nbits / 8
        pbuf = torch.empty((nbits + 63) // 64, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        fill = lambda first, n: _lib.check(L.bbb_prbs_fill(31, 1, first, n, C.c_void_p(pbuf.data_ptr() + first // 8), local_rank, sp), "bbb_prbs_fill")
        check = lambda first, n: _lib.check(L.bbb_prbs_check_dev(31, 1, first, n, C.c_void_p(pbuf.data_ptr() + first // 8),
                                                                C.c_void_p(cnt.data_ptr()), local_rank, sp), "bbb_prbs_check_dev")
        fill_rb = lambda first, n: _lib.check(L.bbb_prbs_fill_hint(31, 1, first, n, C.c_void_p(pbuf.data_ptr() + first // 8), 1, local_rank, sp), "bbb_prbs_fill_hint")
        for _ in range(4):                      # (the buffer's pages, the region-seed plan, clocks)
            fill_rb(0, nbits); check(0, nbits)
        torch.cuda.synchronize()
        reps = 5
        # the loopback as a loopback caller runs it: the fill told that a check follows (BBB_PRBS_WILL_READ_BACK)
        fh_ms = ch_ms = 0.0
        for _ in range(reps):
            e = [ev() for _ in range(3)]
            e[0].record(); fill_rb(0, nbits); e[1].record(); check(0, nbits); e[2].record()
            torch.cuda.synchronize()
            fh_ms += e[0].elapsed_time(e[1]) / reps; ch_ms += e[1].elapsed_time(e[2]) / reps
        f_ms = c_ms = c2_ms = p_ms = 0.0
        piece = 1 << 31                                  # 256 MiB
        for _ in range(reps):
            e = [ev() for _ in range(6)]
            e[0].record(); fill(0, nbits); e[1].record(); check(0, nbits); e[2].record(); check(0, nbits); e[3].record()
            e[4].record()
            for first in range(0, nbits, piece):
                n = min(piece, nbits - first)
                fill(first, n); check(first, n)
            e[5].record()
            torch.cuda.synchronize()
            f_ms += e[0].elapsed_time(e[1]) / reps; c_ms += e[1].elapsed_time(e[2]) / reps
            c2_ms += e[2].elapsed_time(e[3]) / reps; p_ms += e[4].elapsed_time(e[5]) / reps
        nerr = int(cnt.item())
        tb = lambda ms: round(nbytes / ms / 1e9, 3)
        extra["prbs31_loopback"] = {
            "bits": nbits, "errors": nerr,
            "fill_ms": round(f_ms, 4), "check_after_fill_ms": round(c_ms, 4), "check_clean_ms": round(c2_ms, 4),
            "fill_tb_s": tb(f_ms), "check_after_fill_tb_s": tb(c_ms), "check_clean_tb_s": tb(c2_ms),
            "loopback_ms": round(fh_ms + ch_ms, 4), "loopback_tb_s": round(2 * nbytes / (fh_ms + ch_ms) / 1e9, 3),
            "loopback_hbm_frac": round(2 * nbytes / (fh_ms + ch_ms) / 1e6 / HBM_PEAK_GBS, 4),
            "loopback_form": "bbb_prbs_fill_hint(BBB_PRBS_WILL_READ_BACK) + bbb_prbs_check_dev: the fill writes with non-temporal stores",
            "hinted_fill_ms": round(fh_ms, 4), "check_after_hinted_fill_ms": round(ch_ms, 4),
            "plain_fill_then_check_ms": round(f_ms + c_ms, 4), "plain_fill_then_check_hbm_frac": round(2 * nbytes / (f_ms + c_ms) / 1e6 / HBM_PEAK_GBS, 4),
            "pieced_256MiB_loopback_ms": round(p_ms, 4), "pieced_256MiB_loopback_tb_s": round(2 * nbytes / p_ms / 1e9, 3),
            "gen_gbit_s": round(nbits / f_ms / 1e6, 1), "check_gbit_s": round(nbits / c_ms / 1e6, 1),
            "note": "whole-buffer passes; the checker reads every region from its end on the generator's partition (what the "
                    "memory-side cache still holds); fill beside check on two streams does not overlap (profiles/README.md)"}
        other.append(hbm("prbs_stream_kernel<31,false> (fill)", nbytes, f_ms, "1/8 B per bit written"))
        other.append(hbm("prbs_check_rev_kernel<31> (check right after the fill)", nbytes, c_ms, "1/8 B per bit read"))
        other.append(hbm("prbs_check_rev_kernel<31> (check of a clean buffer)", nbytes, c2_ms, "1/8 B per bit read"))
        del cnt
        # exact self-synchronising detector over the same 1e10-bit stream, with 1e-3 injected errors
        # (chunked FSM with state hand-off)
        det = bbb.PRBSErrorDetector(31, device=local_rank)
        noise = torch.randint(0, 1000, (pbuf.numel(),), device=pbuf.device) == 0
        pbuf ^= noise.to(torch.int64) << 13
        del noise
        for _ in range(40):                     # (workspace, pinned read-back buffer -- the first call takes 2 ms, the second 0.34 -- and the clocks:
            det.run_stream(pbuf, nbits)         #  every call synchronises, and the calls speed up over the first dozen: profiles/r05_det_chunk_sweep.log)
        torch.cuda.synchronize()
        tds = []
        for _ in range(6):
            td0 = time.perf_counter()
            ds = det.run_stream(pbuf, nbits)    # (every call synchronises: it returns the totals)
            tds.append(time.perf_counter() - td0)
        print("detector calls (ms):", " ".join(f"{x * 1e3:.4f}" for x in tds), file=sys.stderr, flush=True)
        td = sum(tds) / len(tds)
        extra["detector_stream"] = {"bits": nbits, "errors": ds["errors"], "resyncs": ds["resyncs"], "chunks": ds["chunks"],
                                    "chunks_rerun": ds["chunks_rerun"], "gbit_s": round(nbits / td / 1e9, 1),
                                    "min_call_gbit_s": round(nbits / min(tds) / 1e9, 1),
                                    "note": "bbb_prbs_detector_stream, totals only, wall time per call over six calls (mean; min_call_gbit_s: the fastest), includes its verify pass, read-back and host synchronisation"}
        r = hbm("det_fused_kernel<31> + verify pass (whole call, wall clock)", nbytes, td * 1e3, "1/8 B per bit read")
        r["true_bound"] = "HBM read of the classification pass (0.20-0.22 ms of the call's 0.29-0.31: 5.7-6.3 TB/s); the serial machine visits ~2 % of the words (DESIGN.md 3.7)"
        other.append(r)
        del pbuf
        # TX output stream: shaped PRBS-31 + scaled CLT noise, int16, 8 samples/bit
        # 1e9 samples per call in the staged form (the plain int8 noise kernel into the staging buffer, the transmitter's
        # arithmetic in the SHAPING mover that empties it beside the next call's noise kernel), and 2^29 per call in the
        # one-kernel form (shaper fused into the sample kernel) as in round 1 (the size those figures were quoted on)
        def tx_rate(ntx, level_tx):
            # level_tx None: the stream object (bbb_tx_stream_*: it chooses level 2 itself); else bbb_lutopt_set_staged + plain calls
            tx = bbb.TX(31, 1, 0, 16, 1, 8, device=local_rank)
            txbuf = torch.empty(ntx, dtype=torch.int16, device=dev)
            if level_tx is None:
                stx = tx.stream(ntx)
                call = lambda i: stx.next(out=txbuf)                                    # noqa: E731
            else:
                stx = None
                tx.urng.set_staged(level_tx > 0, look_ahead=level_tx if level_tx >= 2 else False)
                call = lambda i: tx.generate(ntx, first_sample=i * ntx, out=txbuf)      # noqa: E731
            for i in range(24):                                  # (jump plans; and the clock governor's ramp: see the docstring)
                call(i)
            torch.cuda.synchronize()
            t0e, t1e = ev(), ev()
            t0e.record()
            for i in range(24, 64):
                call(i)
            t1e.record()
            torch.cuda.synchronize()
            if stx is not None:
                stx.close()
            del txbuf
            return t0e.elapsed_time(t1e) / 40
        ntx = 1_000_000_000
        # sequential calls on a handle at bbb_lutopt_set_staged level 2: one noise kernel per two calls (what the noise stream
        # object does for its reads); level 1 = one per call, level 4 = one per four calls (2 x 4 GB of staging)
        tx_ms = tx_rate(ntx, None)
        tx_ms_l1, tx_ms_l4 = tx_rate(ntx, 1), tx_rate(ntx, 4)
        tx_ms_1k = tx_rate(1 << 29, 0)
        extra["tx_waveform"] = {"samples": ntx, "gsample_s": round(ntx / tx_ms / 1e6, 1), "ms_per_call": round(tx_ms, 4),
                                "form": "bbb_tx_stream_next (the stream object: staged, one noise kernel -- count planes -- per two calls, a shaping mover per call)",
                                "level_1_one_noise_kernel_per_call": {"gsample_s": round(ntx / tx_ms_l1 / 1e6, 1), "ms_per_call": round(tx_ms_l1, 4)},
                                "level_4_one_noise_kernel_per_four_calls": {"gsample_s": round(ntx / tx_ms_l4 / 1e6, 1), "ms_per_call": round(tx_ms_l4, 4)},
                                "one_kernel_form_2p29_per_call": {"gsample_s": round((1 << 29) / tx_ms_1k / 1e6, 1), "ms_per_call": round(tx_ms_1k, 4)},
                                "note": "bbb_tx_fill_i16; the one-kernel form has the shaper fused into the sample kernel's round end"}
        r = hbm("awgn256_planes_kernel + unplane_kernel<true> (whole bbb_tx_stream_next call)", 2.0 * ntx, tx_ms, "2 B per sample delivered (4 B of HBM traffic: count planes written and read, int16 output written)")
        r["true_bound"] = ("the noise kernel's guests: a shaping mover per call (3 GB of traffic each: 0.75 ms alone, 1.0-1.3 ms beside the kernel) and the next kernel's "
                           "seeding (0.14 ms alone, 0.5 beside); the transmitter's noise kernel is the 344-register form, beside which both run at the same time "
                           "(DESIGN.md 3.4, profiles/r03_small_footprint_ab.log)")
        other.append(r)
        # PRBSShaper.x alone (noise off): PRBS fill + table rows, 2 B per sample written
        txs = bbb.TX(31, 1, 0, 16, 0, 8, device=local_rank)
        sbuf = torch.empty(1 << 30, dtype=torch.int16, device=dev)
        for i in range(2):
            txs.generate(1 << 30, first_sample=i << 30, out=sbuf)
        torch.cuda.synchronize()
        t0e, t1e = ev(), ev()
        t0e.record()
        for i in range(2, 8):
            txs.generate(1 << 30, first_sample=i << 30, out=sbuf)
        t1e.record()
        torch.cuda.synchronize()
        sh_ms = t0e.elapsed_time(t1e) / 6
        del sbuf, txs
        extra["shaper_only"] = {"samples": 1 << 30, "gsample_s": round((1 << 30) / sh_ms / 1e6, 1), "ms_per_call": round(sh_ms, 4)}
        other.append(hbm("prbs_stream_kernel + shaper_table_kernel + shaper_only_kernel (whole call, noise off)", 2.0 * (1 << 30), sh_ms,
                         "2 B per sample written"))
        # the reference's matrix search (software/rnghunt) on the GPU: candidates per second for k = 256
        from basebandboard_amd import gf2 as _gf2
        _gf2.search(256, seed=rank + 1, first=0, count=256, device=local_rank)
        tsr = time.perf_counter()
        sidx, _, sst = _gf2.search(256, seed=rank + 1, first=1 << 32, count=1 << 17, device=local_rank)
        tsr = time.perf_counter() - tsr
        extra["matrix_search_k256"] = {"candidates_tested": sst["tested"], "full_degree": sst["full_degree"],
                                       "order_divides": sst["order_divides"], "accepted": sst["primitive"], "first_hit": sidx,
                                       "kcand_s": round(sst["tested"] / (sst["kernel_ns"] * 1e-9) / 1e3, 1) if sst["kernel_ns"] else None,
                                       "call_ms": round(tsr * 1e3, 2),
                                       "note": "bbb_lutopt_search: build + 512 steps + Berlekamp-Massey + primitivity per wavefront; "
                                               "kcand_s from the kernel's duration, call_ms includes the host re-check of the hit"}
        # BER sweep (BASELINE configs[3]/[4]): Eb/N0 0..10 dB, 1e9 bits/point.
        # N > 1 (BASELINE configs[4]): points x seeds -- every rank runs all 11 points on its own seed (one noise
        # pass per rank), ONE all-reduce (RCCL) sums the uint64 counters: N times the bits per point in the same time.
        nv = 8
        trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv) for db in range(11)]
        # one seed per rank = the reset state jumped 2^48 r clocks ahead: disjoint stretches of the one cycle (reset states that
        # differ by small integers are XOR-dependent and nothing keeps their streams apart)
        us = u if world == 1 else bbb.LUTOPT.shipped(256, init=u.state_at(rank << 48), device=local_rank)
        # untimed: builds the jump plans (tables per segment length).  At another stream position, so that the timed sweep
        # derives its own start states: the library keeps the last start states of a handle and would hand them back
        # (three of them: the handle keeps two sets of start-state buffers and takes them in turn, and the first calls of a
        # process spend 0.2-40 ms on the host -- plans, buffers, the runtime's own pools: experiments/ber_host2.py)
        for wi in range(3):
            warm = [channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=(wi + 1) << 20) for t in trials]
            channel.sweep_seeds(warm, channel.gpu_runner(us), world=world)
        torch.cuda.synchronize(); barrier()
        # the same sweep sixteen times back to back (bbb_ber_trials_dev does not synchronise), every one at another stream
        # position, so every one derives its own start states: the library does that on internal streams while the kernel of
        # the sweep before runs, and the host's work in front of a call's first launch overlaps it too.  (These sixteen also
        # are the clock ramp in front of the isolated calls below: the governor needs ~50 ms of load to leave its idle state.)
        nrep = 16
        reps_t = [[channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=(2 + i) << 21) for t in trials] for i in range(2 * nrep)]
        runner = channel.gpu_runner(us)
        for i in range(nrep):                       # (untimed: clocks)
            channel.sweep_seeds(reps_t[nrep + i], runner, world=world)
        torch.cuda.synchronize(); barrier()
        tb0 = time.perf_counter()
        for i in range(nrep):
            channel.sweep_seeds(reps_t[i], runner, world=world)
        torch.cuda.synchronize(); barrier()
        tber = (time.perf_counter() - tb0) / nrep
        # ONE isolated call (the GPU idle before and after, its seeding, zeroing and read-back inside): what configs[3] literally is,
        # and every device's share of configs[4].  Through the C ABI's synchronous entry, bbb_ber_trials; five calls at five stream
        # positions, each alone, the median reported (the first of them follows the sixteen above directly: hot clocks).
        iso = []
        for i in range(5):
            ts_i = channel.prepare([channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=(100 + i) << 21) for t in trials] if i else trials)
            torch.cuda.synchronize(); barrier()
            tb0 = time.perf_counter()
            got_i = channel.run_trials(us, ts_i)
            iso.append(time.perf_counter() - tb0)
            if i == 0:
                tot_local = got_i
        tber_isolated = sorted(iso)[len(iso) // 2]
        total = torch.tensor([list(x) for x in tot_local], dtype=torch.int64, device=dev)
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(total, op=dist.ReduceOp.SUM)
            else:
                hc = total.cpu(); dist.all_reduce(hc, op=dist.ReduceOp.SUM); total.copy_(hc)
        tot = total.cpu().tolist()
        extra["ber_sweep"] = {
            "points": [{"ebn0_db": round(channel.ebn0_db(t.amp, nv), 3), "ebn0_db_effective": round(channel.ebn0_db_effective(t.amp, nv), 3),
                        "amp": t.amp, "noise_var": nv, "bits": b_, "errors": e_, "ber": e_ / b_ if b_ else None,
                        "q_theory": channel.ber_theory(channel.ebn0_db(t.amp, nv)), "q_lattice": channel.ber_lattice(t.amp, nv)}
                       for t, (b_, e_) in zip(trials, tot)],
            # (round 3's meaning of gbit_s: ONE isolated call.  Round 4 reported the back-to-back figure under this key)
            "gbit_s": round(11e9 / tber_isolated / 1e9, 2), "seconds": round(tber_isolated, 6),
            "gbit_s_is": "ONE isolated bbb_ber_trials call per rank (11 points x 1e9 bits, one pass of the noise stream), its seeding, the zeroing of its counters, "
                         "its read-back and the host's synchronisation inside; median of five calls at five stream positions, each alone on an idle "
                         "(hot) GPU",
            "isolated_call_gbit_s": round(11e9 / tber_isolated / 1e9, 2), "isolated_call_seconds": round(tber_isolated, 6),
            "isolated_calls_ms": [round(x * 1e3, 4) for x in iso],
            "back_to_back_gbit_s": round(11e9 / tber / 1e9, 2), "back_to_back_seconds": round(tber, 6),
            "back_to_back_is": "per sweep over sixteen sweeps queued back to back (bbb_ber_trials_dev), each at its own stream position with its own seeding",
            "seeds": world, "seeding_in_timed_region": True,
            "labels": "ebn0_db = amp^2 / (2 (8 nv)^2), ignores that the sample is an integer; ebn0_db_effective / q_lattice account for "
                      "the integer decision threshold (channel.ber_lattice) and are the ones comparable with Q(sqrt(2 Eb/N0))",
            "reduce": "torch.distributed.all_reduce(int64[11,2], SUM) over RCCL, one seed per rank" if world > 1 else "single rank"}
        # BASELINE configs[4] at its stated size: 11 points x 8 seeds = 88 trials, the seeds as stretches of the one cycle 2^48 apart
        # (warmup = 16 + (s << 48)), eight groups of eleven.  N = 1: all of them on this device through the C ABI's multi-device entry
        # (bbb_ber_sweep_multi, BBB_SHARD_GROUPS: eight sweeps back to back) -- the figure an N-GPU run of the SAME 88 trials is
        # divided by.  N > 1 (one process per GPU): rank r runs groups r, r + N, ... and ONE all-reduce sums the counters.
        from basebandboard_amd import _lib as _l2
        from basebandboard_amd.channel import sweep_multi, multi_info
        def trials88(pos):
            return [channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=pos << 21, warmup=WARM_STATE + (s_ << 48))
                    for s_ in range(8) for t in trials]
        c88 = torch.zeros((88, 2), dtype=torch.int64, device=dev)
        # (the lists are marshalled for the C ABI BEFORE the timed calls: channel.prepare -- 88 ctypes structs cost Python 0.2 ms,
        # and at N > 1 this rank's share is computed here as well)
        prep88 = {}
        for pos in (200, 201, 210, 211, 212):
            ts88 = trials88(pos)
            prep88[pos] = channel.prepare(ts88 if world == 1 else channel.shard_trials(ts88, rank, world, _l2.SHARD_GROUPS))
        def run88(pos):
            if world == 1:
                return sweep_multi([u], prep88[pos], mode=_l2.SHARD_GROUPS)
            c88.zero_()
            channel.run_trials_into(u, prep88[pos], c88)
            if backend == "nccl":
                dist.all_reduce(c88, op=dist.ReduceOp.SUM)
            else:
                hc = c88.cpu(); dist.all_reduce(hc, op=dist.ReduceOp.SUM); c88.copy_(hc)
            return [tuple(x) for x in c88.cpu().tolist()]
        run88(200); run88(201)
        t88 = []
        for i in range(3):
            torch.cuda.synchronize(); barrier()
            tb0 = time.perf_counter()
            got88 = run88(210 + i)                   # (returns host counters: this rank has synchronised; at N > 1 behind the all-reduce,
            t88.append(time.perf_counter() - tb0)    #  which no rank leaves before every rank has entered it)
        if world > 1:
            # the job's time = the slowest rank's, reduced OUTSIDE the timed calls (a closing barrier inside them would charge its own
            # 30-50 us to a 1.3 ms measurement; the opening barrier stays)
            tt = torch.tensor(t88, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t88 = [float(x) for x in tt.tolist()]
        t88m = sorted(t88)[1]
        extra["ber_sweep_88"] = {
            "trials": 88, "bits": 88_000_000_000, "n_devices": world, "seconds": round(t88m, 6), "gbit_s": round(88e9 / t88m / 1e9, 2),
            "runs_ms": [round(x * 1e3, 4) for x in t88],
            "counters_first_point_first_seed": list(got88[0]),
            "what": ("bbb_ber_sweep_multi(BBB_SHARD_GROUPS) over this one device: eight 11-point sweeps back to back, one read-back (the all-reduce of uint64[176] is queued at more than one device only)"
                     if world == 1 else
                     f"one process per GPU: rank r runs groups r, r + {world}, ... (bbb_sweep_shard, BBB_SHARD_GROUPS) through bbb_ber_trials_dev, ONE all-reduce of int64[88, 2]"),
            "scaling": "strong: the SAME 88 trials at every N; speed-up at N = this record's seconds at N = 1 / at N"}
        if world == 1:
            # the projection for eight devices from what ONE device measures: each device runs ONE isolated sweep through the same
            # entry, then the all-reduce of 176 words over eight ranks (not measurable here: priced at 30 us, RCCL's small-message latency)
            sweep_multi([u], [channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=300 << 21) for t in trials])
            tmi = []
            for i in range(5):
                ts_i = channel.prepare([channel.Trial(nbits=t.nbits, amp=t.amp, noise_var=nv, first_bit=(301 + i) << 21) for t in trials])
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                got = sweep_multi([u], ts_i)
                tmi.append(time.perf_counter() - tb0)
            tm = sorted(tmi)[2]
            got0 = sweep_multi([u], trials)
            extra["ber_sweep_multi_c_abi"] = {"n_devices": 1, "equals_single_device_counters": [list(x) for x in got0] == tot,
                                              "gbit_s": round(11e9 / tm / 1e9, 2), "seconds": round(tm, 6), "calls_ms": [round(x * 1e3, 4) for x in tmi],
                                              "what": "ONE isolated 11-point sweep through bbb_ber_sweep_multi (median of five, each alone)",
                                              "reduce": "ncclAllReduce(ncclUint64, ncclSum) of uint64[22] inside bbb_ber_sweep_multi at more than one device; over ONE device the "
                                                        "communicator is created and asked for its size, the sum is the identity and no collective is queued",
                                              **multi_info()}
            allreduce_8_s = 30e-6
            extra["ber_sweep_88"]["projected_8_gpu"] = {
                "seconds": round(tm + allreduce_8_s, 6), "speedup": round(t88m / (tm + allreduce_8_s), 2),
                "arithmetic": "t(1 GPU, 88 trials) / (t(one isolated sweep through bbb_ber_sweep_multi on one device) + 30 us for the eight-rank all-reduce of 176 words)",
                "measured": False}
        # The same sweep CONTINUED over many calls (bbb_ber_run_*: one seeding of the generators per block of 8 calls, the
        # kernel leaves its states for the next call): what a Monte-Carlo run that keeps adding bits until it has seen enough
        # errors pays per 11 x 1e9 bits.  16 calls timed, the blocks' seedings inside; every rank on its own seed, ONE
        # all-reduce of the totals at the end.
        cont_calls, cont_block = 16, 8
        with channel.ContinuedTrials(us, trials, cont_block) as run:
            cacc = torch.zeros((len(trials), 2), dtype=torch.int64, device=dev)
            for _ in range(cont_block):
                run.next_into(cacc)                 # (jump plans of the block's segment length; clocks)
            cacc.zero_()
            torch.cuda.synchronize(); barrier()
            tc0 = time.perf_counter()
            for _ in range(cont_calls):
                run.next_into(cacc)
            torch.cuda.synchronize(); barrier()
            tcont = time.perf_counter() - tc0
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(cacc, op=dist.ReduceOp.SUM)
            else:                                    # (gloo rehearsal: through the host)
                hc = cacc.cpu()
                dist.all_reduce(hc, op=dist.ReduceOp.SUM)
                cacc.copy_(hc)
        ctot = cacc.cpu().tolist()
        cont_ms = tcont / cont_calls * 1e3
        extra["ber_sweep_continued"] = {
            "calls": cont_calls, "calls_per_seeding": cont_block, "bits_per_point_and_call": 1_000_000_000,
            "ms_per_call": round(cont_ms, 4), "gbit_s": round(world * cont_calls * 11e9 / tcont / 1e9, 2),
            "bits_per_point": ctot[0][0], "errors": [e_ for _, e_ in ctot],
            "what": "bbb_ber_run_next_dev: 11 settings x 1e9 bits per call on one pass of the noise stream, a block of 8 calls shares one seeding "
                    "(the (bit, sample) pairs of a block are those of one bbb_ber_trials call over its 8e9 bits: tests/test_gpu_ber.py)"}
        # roofline record of the trial kernel: integer VALU issue, like the sample kernel (counts from this round's SQ pass)
        vps, kms, ksrc = 1211.0, None, None
        for name in ("r05_ber_pmc.json", "r04_ber_pmc.json"):
            try:
                bp = json.load(open(ROOT / "profiles" / name))
                vps = bp["fused"]["valu_insts_per_step_and_wave"]
                kms = bp["fused"].get("kernel_ms_avg")
                ksrc = name
                break
            except Exception:
                continue
        # the kernel's duration: the committed trace's average (isolated launches, clocks not ramped) or, when that is the larger, this
        # run's whole back-to-back sweep (an upper bound of its kernel: the seeding runs beside the previous kernel)
        if kms and kms > tber * 1e3:
            kms, ksrc = tber * 1e3, "this run: one whole sweep of sixteen queued back to back (upper bound of its kernel)"
        t_k = (kms or cont_ms) * 1e-3
        ber_issued_t = vps / 32.0 * 1e9 / t_k / 1e12
        other.append({"kernel": "ber256_fused_kernel<fast, 11> (one pass of the noise stream for 11 channel settings)", "bound": "valu",
                      "achieved": round(ber_issued_t, 2), "peak": 78.64, "unit": "T lane-op/s", "frac": round(ber_issued_t / 78.64, 3),
                      "valu_frac_issued": round(ber_issued_t / 78.64, 3), "valu_frac_issued_of_1wave_ceiling": round(ber_issued_t / 39.32, 3),
                      "valu_issued_per_step_and_wave": vps, "kernel_ms_avg": round(t_k * 1e3, 4),
                      "kernel_ms_source": (ksrc if ksrc and not ksrc.endswith(".json") else f"profiles/{ksrc} (rocprofv3 kernel trace)") if kms else "bbb_ber_run_next_dev per call (includes the state write-back)",
                      "valu_issued_source": f"SQ_INSTS_VALU of profiles/{[n for n in ('r05_ber_pmc.json', 'r04_ber_pmc.json') if (ROOT / 'profiles' / n).exists()][0]}",
                      "algorithmic_bytes_per_launch": 0,
                      "what": "no sample stream is written: 1061-instruction generator step + 16 more AGPR moves + 8 + 11 x 10 comparator instructions per 32 bits and "
                              "lane; one wave per SIMD, power limited (2.15-2.2 GHz alone: profiles/README.md)"})
        if world > 1:
            # BASELINE configs[4] the other way: every rank runs all points over ITS slice of the bit range, ONE all-reduce; the totals
            # must equal the undivided trials on one device exactly (rank 0 runs those too)
            tb1 = time.perf_counter()
            btot = channel.sweep_bits(trials, channel.gpu_runner(u), rank=rank, world=world)
            torch.cuda.synchronize(); barrier()
            tb1 = time.perf_counter() - tb1
            single = channel.run_trials(u, trials) if rank == 0 else None
            extra["ber_sweep_bits_sharded"] = {"n_ranks": world, "gbit_s": round(11e9 / tb1 / 1e9, 2),
                                               "equals_single_device_counters": (btot.cpu().tolist() == [list(x) for x in single]) if rank == 0 else None,
                                               "reduce": "torch.distributed.all_reduce(int64[11,2], SUM) over RCCL, one bit slice per rank"}
    if rank == 0:
        # issued VALU instructions per step and wave: from the SQ pass of this round (SQ_INSTS_VALU), else the ISA count
        sk = pmc_rec.get("sample_kernel", {})
        issued_per_step = sk.get("valu_insts_per_step_and_wave")
        issued_src = "SQ_INSTS_VALU of the committed counter pass" if issued_per_step else "ISA count (hipcc -S): 918 network + 142 accvgpr moves"
        if not issued_per_step:
            issued_per_step = 1060.5 if staged else 1249.0
        samples_per_s = achieved * 1e9                                   # of the sample kernel while it runs
        net_t = ops_per_step / 32.0 * samples_per_s / 1e12               # network lane-ops only
        issued_t = issued_per_step / 32.0 * samples_per_s / 1e12         # every issued VALU instruction (moves, address arithmetic)
        mover_avg_ms = mover_ms / max(movers, 1)
        if staged and mover_avg_ms > 0:
            mbytes = (pmc_rec.get("mover", {}).get("hbm_bytes_per_launch") or {}).get("total")
            other.insert(0, {"kernel": "unplane_kernel<false> (the mover of the two-kernel form, beside the next sample kernel)", "bound": "hbm",
                             "achieved": round(2.0 * NSAMP / mover_avg_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(2.0 * NSAMP / mover_avg_ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 2 * NSAMP,
                             "kernel_ms_avg": round(mover_avg_ms, 4), "launches_timed": int(movers), "traffic": mbytes,
                             "what": "1 B per sample read (count planes) + 1 B written (the stream); a guest of the sample kernel: one 58-register "
                                     "wave per SIMD, LDS-DMA loads, bound by latency and by the issue slots the host wave leaves (about one in ten cycles)"})
        out = {
            "metric": "awgn_clt_gsamples_per_s", "value": round(value, 3), "unit": "Gsample/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            # (a multi-rank run checks itself: ranks in the process group, every rank's own time -- ms_per_step is their MAX)
            "n_ranks_seen": rank_info["n_ranks_seen"], "per_rank_ms_per_step": rank_info["per_rank_ms_per_step"],
            "vs_baseline": None, "dtype": "u32 bit-sliced GF(2) / int8 out", "data": "synthetic",
            "config": {"workload": "CLT AWGN (LUTOPT-256 -> CLTGRNG adder tree), 1e9 int8 samples/step/GPU, init=1, "
                                   "warm-up 16, sequential reference stream (BASELINE configs[1]; reference-faithful "
                                   "generator, no xorshift/CLT-12 exists in the reference)",
                       "samples_per_step_per_gpu": NSAMP, "seeding_in_timed_region": True,
                       "form": (f"bbb_awgn_stream_next on the rank's sample stream: two kernels -- sample kernel -> count planes in a staging slot, "
                                f"mover -> bytes, the movers beside the NEXT sample kernel; one sample-kernel launch (and one seeding) per "
                                f"{max(look_ahead, 1)} consecutive steps, the timed region starts on a launch; all of it inside the timed region")
                               if staged else "bbb_awgn_fill_i8 + bbb_awgn_prefetch, one kernel",
                       "clock_ramp_steps": ramp_steps,
                       "clock_ramp_note": "untimed steps of the same workload in front of the W warm-up steps: the clock governor needs ~50 ms of load "
                                          "to leave its idle state (profiles/r03_ramp_clock_per_launch.log); extra.cold_start is 20 of the same steps straight from idle",
                       "verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_is": "sample kernel only: algorithmic bytes per launch / its mean duration (hipEvents on its stream; a launch "
                                        "that is dispatched while the previous sample kernel still holds every SIMD is counted from that "
                                        "kernel's completion -- its time on the machine, which is also the spacing of the kernels' END "
                                        "timestamps in the rocprofv3 trace: tools/trace_spacing.py, profiles/README.md)",
                         # the whole step: sample kernel + mover + seeding, per 1e9 samples delivered
                         "traffic_step": traffic_step,
                         "kernel": "awgn256_planes_kernel" if staged else "awgn256_kernel<false>", "kernel_ms_avg": round(kern_avg_ms, 4),
                         "seed_ms_avg": round(seed_ms / max(calls, 1), 4), "launches_timed": int(calls),
                         "algorithmic_bytes_per_launch": per_launch,
                         "steps_per_launch": per_launch // NSAMP,
                         "streaming_fill_gb_s": round(fill_gbs, 1) if fill_gbs else None,
                         "frac_of_streaming_fill": round(achieved / fill_gbs, 4) if fill_gbs else None,
                         "true_bound": "integer VALU issue (bit-sliced XOR / majority network) and, with the guests beside it, the chip's power "
                                       "limit (the shader clock sits at 2.2 GHz under this load, 2.39 with either the staging stores or the mover's "
                                       "traffic taken away: `clock_in_stream`), not HBM",
                         # committed measurement (per-wave clock stamps in one-off variants of the product kernel), not taken live
                         "clock_in_stream": stream_clock_record(),
                         "valu_lane_ops_per_sample": round(ops_per_step / 32.0, 2),
                         "valu_issued_per_step_and_wave": round(issued_per_step, 1), "valu_issued_source": issued_src,
                         "valu_net_tlaneops_s": round(net_t, 2),
                         "valu_issued_tlaneops_s": round(issued_t, 2),
                         # three denominators, named: the chip's nominal VALU issue peak (256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz: one
                         # wave-instruction per 2 cycles per SIMD); the one-wave-per-SIMD ceiling the 256-plane state forces on this kernel
                         # (a single wave issues one VALU instruction per 4 cycles); and the V_BITOP3 rate MEASURED with two and more waves
                         # per SIMD on this part (profiles/r01_design_ubench.log: 2.85 cycles per instruction = 55-57 T lane-op/s)
                         "valu_peak_tlaneops_s": 78.64,
                         "valu_peak_1wave_tlaneops_s": 39.32,
                         "valu_attainable_measured_tlaneops_s": 56.0,
                         "valu_frac": round(net_t / 78.64, 3),
                         "valu_frac_issued": round(issued_t / 78.64, 3),
                         "valu_frac_issued_of_1wave_ceiling": round(issued_t / 39.32, 3),
                         # (the same figure under the name the round-4 verdict asked for: the bound this line is judged on is VALU issue at
                         # one wave per SIMD, not HBM -- `bound` / `frac` stay what the contract defines them as)
                         "valu_frac_of_1wave_ceiling": round(issued_t / 39.32, 3),
                         "bound_that_binds": "valu issue at one wave per SIMD (256-plane bit-sliced state: 512 registers per wave)",
                         "valu_frac_issued_of_measured_attainable": round(issued_t / 56.0, 3)},
        }
        extra["cold_start"] = {"ms_per_step": round(cold_ms, 4), "gsample_s": round(NSAMP / cold_ms / 1e6, 1),
                               "what": f"{cold_steps} of the same steps timed straight after the parity copy, GPU coming out of idle, no clock ramp"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if extra:
                extra["cpu_baseline_other"] = cpu_baseline_other()
        if other:
            out["roofline_other"] = other
        if extra:
            out["extra"] = extra
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
