#!/usr/bin/env python3
"""Headline benchmark: CLT Gaussian noise (AWGN) sample generation on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: 1e9 int8 CLT samples of the reference's
LUTOPT-256 -> CLTGRNG generator (BASELINE.json configs[1]; the reference has no "CLT-12 /
xorshift32" generator) written to HBM.  Step s of rank r generates
stream positions [16 + (s*world + r)*1e9, +1e9): a different part of the SAME sequential stream
every step and every rank, so nothing is cached between steps and ranks are independent shards
(weak scaling, no data-path collective).  Seeding (GF(2) jump-ahead on the GPU) is inside the
timed region; the seeding of step s+1 is announced with bbb_awgn_prefetch right after step s is
launched, so that it runs beside step s's sample kernel (BENCH_NO_PREFETCH=1 turns that off).

The JSON line also carries
  roofline     achieved HBM-write GB/s of the sample kernel (algorithmic 1 B/sample / its mean
               launch time from hipEvents on the launch stream) against the 8 TB/s HBM peak.
               The kernel is integer-VALU bound (~1250 lane-ops per 32 samples), so this fraction
               is expected to be far below 1; `valu_*` fields give the bound that applies.
  cpu_baseline the CPU oracle (k=256 fast path, 1 thread, -march=native) timed on this host
               on a bounded sample (rank 0, N = 1 only).
  extra        PRBS-31 generate+check loopback and a BER sweep, measured after the timed region.
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


def cpu_baseline():
    """Time the oracle's k=256 fast path on this host (checker code used as the reported CPU
    baseline only).  Bounded sample: ~2e8 samples (10-30 s)."""
    import numpy as np
    import oracle as O
    so = O.build(native=True, out="/tmp/libbbb_oracle_native.so", force=True)
    lib = O.lib(path=so)
    m = O.Lutopt(path=O.data_path(256), _lib=lib)
    n = 20_000_000
    t0 = time.perf_counter()
    m.awgn(1, WARM_STATE, n, fast=True)
    dt = time.perf_counter() - t0
    reps = max(1, min(10, int(15.0 / max(dt, 1e-3))))
    t0 = time.perf_counter()
    for i in range(reps):
        m.awgn(1 + i, WARM_STATE, n, fast=True)      # same length, different seed each repetition
    dt = time.perf_counter() - t0
    # the same restatement on a share of the host's cores (one stream offset per thread via a different seed;
    # ctypes releases the GIL around the call).  Reported beside the single-core figure, not instead of it.
    import concurrent.futures
    nthr = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with concurrent.futures.ThreadPoolExecutor(nthr) as ex:
        t1 = time.perf_counter()
        list(ex.map(lambda i: m.awgn(100 + i, WARM_STATE, n, fast=True), range(nthr)))
        dtn = time.perf_counter() - t1
    return {"value": round(reps * n / dt / 1e9, 5), "unit": "Gsample/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x {n} samples of the same stream (oracle k=256 byte-table path, gcc -O3 -march=native, "
                      f"host has {os.cpu_count()} logical cores)",
            "threads_value": round(nthr * n / dtn / 1e9, 5), "threads": nthr}


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
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
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
            print(json.dumps({"launch_check": world, "sum": int(t.item())}))
        dist.barrier()
        dist.destroy_process_group()
        return
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
    g = bbb.CLTGRNG(u)
    buf = torch.empty(NSAMP, dtype=torch.int8, device=f"cuda:{local_rank}")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def first_step(step):
        return WARM_STATE + (step * world + rank) * NSAMP

    # parity spot check before timing (rank 0): a prefix of step 0 against the oracle
    verified = None
    g.generate(NSAMP, first_step=first_step(0), out=buf)
    if rank == 0:
        import numpy as np
        import oracle as O
        m = O.Lutopt(path=O.data_path(256))
        verified = bool(np.array_equal(buf[:1_000_000].cpu().numpy(), m.awgn(1, WARM_STATE, 1_000_000, fast=True)))
        tail0 = NSAMP - 4096
        verified = verified and bool(np.array_equal(buf[tail0:].cpu().numpy(),
                                                    m.awgn(u.state_at(WARM_STATE + tail0), 0, 4096, fast=True)))
    prefetch = not os.environ.get("BENCH_NO_PREFETCH")
    for s in range(1, args.warmup + 1):
        g.generate(NSAMP, first_step=first_step(s), out=buf)
        if prefetch:
            g.prefetch(NSAMP, first_step=first_step(s + 1))
    u.profile(True)
    u.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        g.generate(NSAMP, first_step=first_step(args.warmup + 1 + s), out=buf)
        if prefetch:      # seeding of the following step, issued now so that it runs beside this step's kernel
            g.prefetch(NSAMP, first_step=first_step(args.warmup + 2 + s))
    barrier()
    dt = time.perf_counter() - t0
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    u.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    value = world * args.steps * NSAMP / dt / 1e9
    kern_avg_ms = kern_ms / max(calls, 1)
    achieved = NSAMP / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0

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

    # PMC-measured HBM write bytes per launch, if a profile summary of this command is committed
    traffic = None
    pmc = ROOT / "profiles" / "r01_awgn256_pmc.json"
    if pmc.exists():
        try:
            traffic = json.load(open(pmc)).get("awgn256_kernel_hbm_bytes_per_launch")
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
    if not args.no_extra:
        # PRBS-31 loopback (BASELINE configs[2]): 1e10 bits written then read back and checked
        nbits = 10_000_000_000
        p = bbb.PRBS(31, device=local_rank)
        det = bbb.PRBSErrorDetector(31, device=local_rank)
        pbuf = torch.empty((nbits + 63) // 64, dtype=torch.int64, device=f"cuda:{local_rank}")
        p.generate(nbits, out=pbuf)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = 5
        nerr = None
        gen_ms = chk_ms = 0.0
        for _ in range(reps):
            e0.record(); p.generate(nbits, out=pbuf); e1.record()
            nerr = det.count_errors(pbuf, nbits); e2.record()
            torch.cuda.synchronize()
            gen_ms += e0.elapsed_time(e1); chk_ms += e1.elapsed_time(e2)
        extra["prbs31_loopback"] = {
            "bits": nbits, "errors": int(nerr),
            "gen_gbit_s": round(nbits * reps / gen_ms / 1e6, 1), "check_gbit_s": round(nbits * reps / chk_ms / 1e6, 1),
            "gen_hbm_write_gb_s": round(nbits / 8 * reps / gen_ms / 1e6, 1),
            "check_hbm_read_gb_s": round(nbits / 8 * reps / chk_ms / 1e6, 1),
            "hbm_frac_write": round(nbits / 8 * reps / gen_ms / 1e6 / HBM_PEAK_GBS, 4),
            "hbm_frac_read": round(nbits / 8 * reps / chk_ms / 1e6 / HBM_PEAK_GBS, 4)}
        # exact self-synchronising detector over the same 1e10-bit stream, with 1e-3 injected errors
        # (chunked FSM with state hand-off)
        noise = torch.randint(0, 1000, (pbuf.numel(),), device=pbuf.device) == 0
        pbuf ^= noise.to(torch.int64) << 13
        del noise
        det.run_stream(pbuf, nbits)
        torch.cuda.synchronize()
        td = time.perf_counter()
        ds = det.run_stream(pbuf, nbits)
        torch.cuda.synchronize()
        td = time.perf_counter() - td
        extra["detector_stream"] = {"bits": nbits, "errors": ds["errors"], "resyncs": ds["resyncs"], "chunks": ds["chunks"],
                                    "chunks_rerun": ds["chunks_rerun"], "gbit_s": round(nbits / td / 1e9, 1),
                                    "note": "bbb_prbs_detector_stream, totals only, includes its verify passes and host syncs"}
        del pbuf
        # TX output stream: shaped PRBS-31 + scaled CLT noise, int16, 8 samples/bit
        ntx = 1 << 29
        tx = bbb.TX(31, 1, 0, 16, 1, 8, device=local_rank)
        txbuf = torch.empty(ntx, dtype=torch.int16, device=f"cuda:{local_rank}")
        tx.generate(ntx, out=txbuf)
        torch.cuda.synchronize()
        t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0e.record()
        for i in range(3):
            tx.generate(ntx, first_sample=(i + 1) * ntx, out=txbuf)
        t1e.record()
        torch.cuda.synchronize()
        extra["tx_waveform"] = {"samples": ntx, "gsample_s": round(3 * ntx / t0e.elapsed_time(t1e) / 1e6, 1),
                                "note": "bbb_tx_fill_i16: PRBS fill + CLT noise fill + shaper/combine kernel, int16 out"}
        del txbuf
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
        # BER sweep (BASELINE configs[3]/[4]): Eb/N0 0..10 dB, 1e9 bits/point, sharded over ranks
        # (point i -> rank i % world), ONE all-reduce (RCCL) of the uint64 counters.
        # N > 1 (BASELINE configs[4]): points x seeds -- every rank runs all 11 points on its own seed (one noise
        # pass per rank), the counters are summed over ranks: N times the bits per point in the same time.
        nv = 8
        trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv) for db in range(11)]
        us = u if world == 1 else bbb.LUTOPT.shipped(256, init=1 + rank, device=local_rank)
        channel.sweep_seeds(trials, channel.gpu_runner(us), world=world)       # untimed: builds the jump plans
        torch.cuda.synchronize(); barrier()
        tb = time.perf_counter()
        total = channel.sweep_seeds(trials, channel.gpu_runner(us), world=world)
        torch.cuda.synchronize(); barrier()
        tber = time.perf_counter() - tb
        tot = total.cpu().tolist()
        extra["ber_sweep"] = {
            "points": [{"ebn0_db": round(channel.ebn0_db(t.amp, nv), 3), "amp": t.amp, "noise_var": nv, "bits": b, "errors": e,
                        "ber": e / b if b else None, "q_theory": channel.ber_theory(channel.ebn0_db(t.amp, nv))}
                       for t, (b, e) in zip(trials, tot)],
            "gbit_s": round(sum(b for b, _ in tot) / tber / 1e9, 2), "seconds": round(tber, 4),
            "seeds": world,
            "reduce": "torch.distributed.all_reduce(int64[11,2], SUM) over RCCL, one seed per rank" if world > 1 else "single rank"}

    if rank == 0:
        out = {
            "metric": "awgn_clt_gsamples_per_s", "value": round(value, 3), "unit": "Gsample/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 bit-sliced GF(2) / int8 out", "data": "synthetic",
            "config": {"workload": "CLT AWGN (LUTOPT-256 -> CLTGRNG adder tree), 1e9 int8 samples/step/GPU, init=1, "
                                   "warm-up 16, sequential reference stream (BASELINE configs[1]; reference-faithful "
                                   "generator, no xorshift/CLT-12 exists in the reference)",
                       "samples_per_step_per_gpu": NSAMP, "seeding_in_timed_region": True,
                       "seeding_overlapped_by_prefetch_hint": bool(prefetch),
                       "verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "awgn256_kernel", "kernel_ms_avg": round(kern_avg_ms, 4),
                         "seed_ms_avg": round(seed_ms / max(calls, 1), 4), "launches_timed": int(calls),
                         "algorithmic_bytes_per_launch": NSAMP,
                         "streaming_fill_gb_s": round(fill_gbs, 1) if fill_gbs else None,
                         "frac_of_streaming_fill": round(achieved / fill_gbs, 4) if fill_gbs else None,
                         "true_bound": "integer VALU (bit-sliced XOR/majority network), not HBM",
                         "valu_lane_ops_per_sample": round(ops_per_step / 32.0, 2),
                         "valu_net_tlaneops_s": round(ops_per_step / 32.0 * achieved / 1e3, 2),
                         # issue ceiling at ONE wave per SIMD (the kernel needs the whole register file): one VALU
                         # instruction per 4 cycles per SIMD (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU quad-cycles,
                         # profiles/r01_d_pmc_sq.json) = 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4; the shader clock
                         # observed under this kernel is 2.16 GHz (GRBM_GUI_ACTIVE), i.e. 35.4 at the real clock
                         "valu_peak_1wave_tlaneops_s": 39.32,
                         "valu_frac_net": round(ops_per_step / 32.0 * achieved / 1e3 / 39.32, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if extra:
                extra["cpu_baseline_other"] = cpu_baseline_other()
        if extra:
            out["extra"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
