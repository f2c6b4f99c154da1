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
