"""The multi-rank sweep (basebandboard_amd.channel.sweep) on CPU: two `gloo` ranks shard the trial
list round-robin, each runs its share, ONE all-reduce sums the 64-bit counters; the result must
equal the single-rank counters exactly.  The per-trial compute is injected (the oracle here: this
file is a test; on GPU ranks it is the HIP kernel via gpu_runner)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _trials():
    from basebandboard_amd import Trial
    return [Trial(nbits=3000 + 500 * i, amp=60 + 20 * i, noise_var=8, prbs_k=(7, 9, 31)[i % 3], first_bit=100 * i)
            for i in range(7)]


def _oracle_runner():
    import sys
    sys.path.insert(0, str(ROOT))
    import oracle as O
    m = O.Lutopt(path=O.data_path(256))

    def run(local_trials, n):
        out = torch.zeros((n, 2), dtype=torch.int64)
        for i, t in enumerate(local_trials):
            b, e = m.ber_trial(1, t.prbs_k, t.prbs_state, t.amp, t.noise_var, t.warmup, t.first_bit, t.nbits)
            out[i, 0], out[i, 1] = b, e
        return out
    return run


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basebandboard_amd import sweep
    total = sweep(_trials(), _oracle_runner(), rank=rank, world=world)
    q.put((rank, total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_equals_single_rank():
    from basebandboard_amd import sweep
    single = sweep(_trials(), _oracle_runner(), rank=0, world=1).tolist()
    assert all(b > 0 for b, _ in single)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == single


def test_shard_is_a_partition():
    from basebandboard_amd import shard
    for n in (0, 1, 7, 88):
        for world in (1, 2, 8):
            parts = [shard(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def _oracle_runner_seed(seed):
    import sys
    sys.path.insert(0, str(ROOT))
    import oracle as O
    m = O.Lutopt(path=O.data_path(256))

    def run(local_trials, n):
        out = torch.zeros((n, 2), dtype=torch.int64)
        for i, t in enumerate(local_trials):
            b, e = m.ber_trial(seed, t.prbs_k, t.prbs_state, t.amp, t.noise_var, t.warmup, t.first_bit, t.nbits)
            out[i, 0], out[i, 1] = b, e
        return out
    return run


def _worker_seeds(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basebandboard_amd.channel import sweep_seeds
    total = sweep_seeds(_trials(), _oracle_runner_seed(1 + rank), world=world)
    q.put((rank, total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_seed_sharded_sweep_sums_the_seeds():
    """points x seeds (BASELINE configs[4]): every rank runs all points on its own seed; the reduced counters
    are the sums of the per-seed single-rank sweeps."""
    from basebandboard_amd.channel import sweep_seeds
    per_seed = [sweep_seeds(_trials(), _oracle_runner_seed(s), world=1).tolist() for s in (1, 2)]
    expect = [[a[0] + b[0], a[1] + b[1]] for a, b in zip(*per_seed)]
    assert per_seed[0] != per_seed[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_seeds, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == expect


def _worker_bits(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basebandboard_amd.channel import sweep_bits
    total = sweep_bits(_trials(), _oracle_runner(), rank=rank, world=world)
    q.put((rank, total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bit_sliced_sweep_equals_single_rank():
    """BBB_SHARD_BITS: every rank runs all trials over its slice of each trial's bit range (bbb_sweep_shard, the
    same arithmetic bbb_ber_sweep_multi uses); the reduced counters ARE the single-rank counters."""
    from basebandboard_amd.channel import sweep_bits
    single = sweep_bits(_trials(), _oracle_runner(), rank=0, world=1).tolist()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bits, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == single


def test_c_abi_shard_arithmetic():
    """bbb_sweep_shard (host only): the shares of the three modes are what include/bbb.h says, for ragged sizes."""
    from basebandboard_amd import Trial, shard
    from basebandboard_amd import _lib
    from basebandboard_amd.channel import shard_trials
    trials = [Trial(nbits=nb, amp=50 + i, noise_var=8, first_bit=fb)
              for i, (nb, fb) in enumerate([(10**9 + 7, 0), (3, 5), (0, 9), (1, 0), (8, 10**15), (12345, 77), (2**40 + 1, 2**40)])]
    for world in (1, 2, 3, 8):
        shares = {m: [shard_trials(trials, r, world, m) for r in range(world)]
                  for m in (_lib.SHARD_TRIALS, _lib.SHARD_SEEDS, _lib.SHARD_BITS)}
        for r in range(world):
            # round robin = the Python `shard` used by the process-per-GPU sweep
            mine = [i for i, t in enumerate(shares[_lib.SHARD_TRIALS][r]) if t.nbits]
            assert mine == [i for i in shard(len(trials), r, world) if trials[i].nbits]
            for a, b in zip(shares[_lib.SHARD_SEEDS][r], trials):
                assert (a.nbits, a.first_bit, a.amp) == (b.nbits, b.first_bit, b.amp)
        for i, t in enumerate(trials):
            assert sum(shares[_lib.SHARD_TRIALS][r][i].nbits for r in range(world)) == t.nbits
            # bit slices: contiguous, in rank order, covering the trial exactly, sizes within one bit of each other
            pos = t.first_bit
            sizes = []
            for r in range(world):
                s = shares[_lib.SHARD_BITS][r][i]
                assert s.first_bit == pos and (s.amp, s.noise_var, s.prbs_state, s.warmup) == (t.amp, t.noise_var, t.prbs_state, t.warmup)
                pos += s.nbits
                sizes.append(s.nbits)
            assert pos == t.first_bit + t.nbits and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_trials(trials, 2, 2)
    with pytest.raises(ValueError):
        shard_trials(trials, 0, 1, 7)


def test_c_abi_shard_groups_keeps_a_sweep_on_one_rank():
    """BBB_SHARD_GROUPS (round 5; host only): consecutive trials on one noise / PRBS stream are one group, group q runs on rank
    q % ndev -- BASELINE configs[4]'s 11 points x 8 seeds is one 11-point sweep per device on eight devices, eight sweeps on one;
    trials that differ in any stream parameter start a new group; a rank's share keeps the trials' order and parameters."""
    from basebandboard_amd import Trial
    from basebandboard_amd import _lib
    from basebandboard_amd.channel import shard_trials
    t88 = [Trial(nbits=10**9, amp=90 + i, noise_var=8, warmup=16 + (s << 48)) for s in range(8) for i in range(11)]
    for world in (1, 2, 3, 8):
        seen = [0] * 88
        for r in range(world):
            mine = shard_trials(t88, r, world, _lib.SHARD_GROUPS)
            for i, (m, t) in enumerate(zip(mine, t88)):
                assert (m.amp, m.warmup, m.first_bit) == (t.amp, t.warmup, t.first_bit)
                assert m.nbits == (t.nbits if (i // 11) % world == r else 0)
                seen[i] += m.nbits != 0
        assert seen == [1] * 88
    # group boundaries: any of prbs_k, prbs_state, warmup, first_bit, nbits
    mixed = [Trial(nbits=100, amp=1, noise_var=8), Trial(nbits=100, amp=2, noise_var=8), Trial(nbits=101, amp=3, noise_var=8),
             Trial(nbits=101, amp=4, noise_var=8, first_bit=5), Trial(nbits=101, amp=5, noise_var=8, first_bit=5, prbs_k=9),
             Trial(nbits=101, amp=6, noise_var=8, first_bit=5, prbs_k=9, prbs_state=3), Trial(nbits=101, amp=7, noise_var=8, first_bit=5, prbs_k=9, prbs_state=3)]
    groups = [0, 0, 1, 2, 3, 4, 4]
    for world in (2, 5):
        for r in range(world):
            mine = shard_trials(mixed, r, world, _lib.SHARD_GROUPS)
            assert [bool(m.nbits) for m in mine] == [g % world == r for g in groups]


def test_thread_per_device_orchestration_under_thread_sanitizer(tmp_path):
    """csrc/sweep_threads.hpp (the host side of bbb_ber_sweep_multi: a thread per device, shares, error hand-back) built for
    the host with -fsanitize=thread and a stub in place of the kernel launches (tests/san_sweep.cpp): no race, totals equal
    the undivided trials for 1, 2, 3, 8 ranks in all three modes.  The launches themselves are rehearsed on the GPU box
    (tests/test_gpu_ber.py::test_thread_per_device_body_rehearsed_on_one_gpu)."""
    import subprocess
    exe = tmp_path / "san_sweep"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", str(ROOT / "tests" / "san_sweep.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("no ThreadSanitizer runtime for g++ here")
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok tsan"), (out.stdout + out.stderr)[-3000:]
    assert "WARNING: ThreadSanitizer" not in out.stderr
