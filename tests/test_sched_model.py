"""The library's host scheduler -- basebandboard_amd/csrc/bbb_api.hip, UNCHANGED: staging slots, slot generations, the
prefetch swap, look-ahead, the stream objects, the transmitter's bit buffers, BER trials and continued trials -- compiled
for the host against a model of HIP streams and events (tests/sched_model/) and driven with random call sequences.  Every
kernel launch is replaced by a stub that records which device buffers the real kernel reads and writes; the model keeps
vector clocks and reports every access that the streams, events and host synchronisations do not order: a write in front
of a buffer's last reader, a read in front of its writer.  That is the hazard the GPU cannot be sanitised for and that
round 3's soak test met on seed 3 of 3.

Three builds prove that the model sees what it has to (the third: -DBBB_SCHED_MODEL_REVERT_MOVER_CHAIN, round 4's rule that a
slot's "free" event stands for ALL movers that read it, also one on the caller's other stream): with -DBBB_SCHED_MODEL_REVERT_UNTAKEN_HINT the scheduler is the one of
commit f1ee557 in that respect (round 3's race: an announcement that was never taken leaves its seeding on one arithmetic
stream, the next seeding goes to the other) and with -DBBB_SCHED_MODEL_REVERT_STALE_SKIP the one before the round-2
advisor's fix (a prefetch's "the seeding waited for this slot's mover" outliving later movers on the slot) -- both must be
FOUND; the scheduler as it stands must come through >= 10 000 sequences clean, under AddressSanitizer +
UndefinedBehaviorSanitizer (leaks included) and under ThreadSanitizer."""
import json
import subprocess

import pytest

from conftest import ROOT

SRC = [str(ROOT / "tests" / "sched_model" / "model.cpp"), str(ROOT / "tests" / "sched_model" / "driver.cpp")]
TAPS = str(ROOT / "basebandboard_amd" / "data" / "lutopt_256.taps")


BUILDS = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
    "tsan": ["-fsanitize=thread"],
    "BBB_SCHED_MODEL_REVERT_UNTAKEN_HINT": ["-DBBB_SCHED_MODEL_REVERT_UNTAKEN_HINT"],
    "BBB_SCHED_MODEL_REVERT_STALE_SKIP": ["-DBBB_SCHED_MODEL_REVERT_STALE_SKIP"],
    "BBB_SCHED_MODEL_REVERT_MOVER_CHAIN": ["-DBBB_SCHED_MODEL_REVERT_MOVER_CHAIN"],
}


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    """the five builds, compiled side by side"""
    d = tmp_path_factory.mktemp("sched_model")
    procs = {}
    for name, flags in BUILDS.items():
        cmd = ["g++", "-std=c++17", "-O1", "-g", *flags, "-I", str(ROOT / "tests" / "sched_model"), "-x", "c++",
               str(ROOT / "basebandboard_amd" / "csrc" / "bbb_api.hip"), "-x", "none", *SRC, "-o", str(d / name), "-ldl", "-lpthread"]
        procs[name] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for name, pr in procs.items():
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, (name, err[-4000:])
    return {name: d / name for name in BUILDS}


def run(exe, nseq, seed, mode="all", max_bad=1000000, timeout=900):
    r = subprocess.run([str(exe), TAPS, str(nseq), str(seed), mode, str(max_bad)], capture_output=True, text=True, timeout=timeout)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(line[-1]) if line else None)


def test_scheduler_orders_every_access_over_ten_thousand_sequences(exes):
    exe = exes["asan"]
    total = 0
    for seed, nseq, mode in ((1, 7000, "all"), (2, 4000, "hints")):
        r, out = run(exe, nseq, seed, mode)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
        assert out["sequences_with_unordered_access"] == 0 and out["sequences"] == nseq and out["operations_checked"] > 10 * nseq
        total += out["sequences"]
    assert total >= 10_000


def test_scheduler_under_thread_sanitizer(exes):
    exe = exes["tsan"]
    r, out = run(exe, 1500, 3)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert out["sequences_with_unordered_access"] == 0 and "ThreadSanitizer" not in r.stderr


@pytest.mark.parametrize("macro,mode,what", [
    ("BBB_SCHED_MODEL_REVERT_UNTAKEN_HINT", "all", "seeding"),
    ("BBB_SCHED_MODEL_REVERT_UNTAKEN_HINT", "hints", "seeding"),
    ("BBB_SCHED_MODEL_REVERT_STALE_SKIP", "hints", "awgn256_planes_kernel"),
    ("BBB_SCHED_MODEL_REVERT_MOVER_CHAIN", "all", "awgn256_planes_kernel"),
])
def test_the_model_finds_the_races_of_rounds_two_and_three(exes, macro, mode, what):
    """The scheduler with one of its ordering rules taken out: the model must report unordered accesses, and of the kind the
    rule is about (two seedings writing the same start-state buffers from different streams; a sample kernel overwriting a
    staging slot its mover still reads -- rounds 2 and 3; round 4, the mover on the caller's stream: a sample kernel
    overwriting a slot that a mover on the caller's OTHER stream still reads after a re-bind the caller did not order)."""
    r, out = run(exes[macro], 6000 if "MOVER_CHAIN" in macro else 3000, 1, mode, max_bad=3)
    assert r.returncode == 1 and out["sequences_with_unordered_access"] > 0
    assert "UNORDERED ACCESS" in r.stderr and what in r.stderr
