"""`python bench.py --gpus N` must run by itself: the parent starts the N ranks as child processes before it
touches any GPU, passes rank 0's JSON line through and fails when a rank fails (the reference's only
multi-worker program has this shape: workers and one channel back, software/rnghunt/src/bin/rnghunt.rs:16-18,54-65)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, cwd=str(ROOT), env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_parent_launches_the_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": 2, "sum": 3}


def test_a_failing_rank_fails_the_launch():
    r = _run(["--gpus", "3", "--launch-check"], env={"BENCH_FAIL_RANK": "2"}, timeout=120)
    assert r.returncode == 3 and "rank 2 exited" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("via", ("self", "torchrun"))
def test_two_ranks_end_to_end_without_a_gpu(via):
    """`bench.py --gpus 2` from the launch to the JSON line with the GPU work stubbed (BENCH_DRY_RUN=1: steps are sleeps of
    (rank + 1) ms, the trial runner returns known counters) -- started by bench.py itself and the way the driver starts it
    (python -m torch.distributed.run).  Checks what a first multi-GPU run cannot afford to get wrong outside the kernels:
    both ranks joined (n_ranks_seen), the timed region is the MAX over ranks, the counters went through ONE sum, the line
    has the contract's fields and nothing else is on stdout."""
    env = {"BENCH_DRY_RUN": "1"}
    if via == "self":
        r = _run(["--gpus", "2", "--steps", "20", "--warmup", "2"], env=env)
    else:
        e = dict(os.environ, **env)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            e.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29611", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2"],
                           cwd=str(ROOT), env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "n_ranks_seen", "per_rank_ms_per_step"):
        assert k in out
    assert out["dry_run"] is True and out["value"] is None
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 20 and out["scaling"] == "weak"
    per = out["per_rank_ms_per_step"]
    assert len(per) == 2 and per[1] > 1.5 * per[0] and per[0] >= 1.0   # rank 1 sleeps twice as long
    assert out["ms_per_step"] >= max(per) - 1e-3                       # barrier to barrier: at least the slowest rank's own time
    # seeds form: every rank's counters summed once -- trial i: bits 2 x 1000, errors (10 i + 0) + (10 i + 1)
    assert out["extra"]["ber_sweep"]["counters"] == [[2000, 20 * i + 1] for i in range(11)]
    # bit-sliced form: the ranks' slices add up to the whole trial
    assert [c[0] for c in out["extra"]["ber_sweep_bits_sharded"]["counters"]] == [1000] * 11


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_box():
    """The multi-rank code path (sharded stream positions, barrier, max-over-ranks timing, all-reduce) with two
    ranks sharing cuda:0 and gloo as the rendezvous backend -- what a 1-GPU box can rehearse of `--gpus 2`."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo", "NCCL_DEBUG": "VERSION"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert r.stdout.strip() == lines[0]            # nothing else on stdout (RCCL's banner goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["n_ranks_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2 and out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-3
    assert out["config"]["verified_vs_oracle"] is True
    assert "cpu_baseline" not in out
    pts = out["extra"]["ber_sweep"]["points"]
    assert out["extra"]["ber_sweep"]["seeds"] == 2 and all(p["bits"] == 2_000_000_000 for p in pts)


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_fields():
    r = _run(["--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(r.stdout.strip().splitlines()) == 1
    out = json.loads(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0 < rf["valu_frac"] < rf["valu_frac_issued"] < rf["valu_frac_issued_of_measured_attainable"] < rf["valu_frac_issued_of_1wave_ceiling"] < 1
    assert rf["kernel_ms_avg"] / rf["steps_per_launch"] <= out["ms_per_step"] * 1.05
    assert out["config"]["clock_ramp_steps"] >= 0 and out["extra"]["cold_start"]["ms_per_step"] > 0
