"""The library's host scheduler -- basebandboard_amd/csrc/bbb_api.hip, UNCHANGED: staging slots, slot generations, the
prefetch swap, look-ahead, the stream objects, the transmitter's bit buffers, BER trials and continued trials -- compiled
for the host against a model of HIP streams and events (tests/sched_model/) and driven with random call sequences.  Every
kernel launch is replaced by a stub that records which device buffers the real kernel reads and writes; the model keeps
vector clocks and reports every access that the streams, events and host synchronisations do not order: a write in front
of a buffer's last reader, a read in front of its writer.  That is the hazard the GPU cannot be sanitised for and that
round 3's soak test met on seed 3 of 3.

Five MUTANTS prove that the model sees what it has to: copies of the file with ONE ordering rule removed by exact text (the
product source carries no test macro since round 5) -- round 3's race (an announcement that was never taken leaves its seeding
on one arithmetic stream, the next seeding goes to the other), the round-2 advisor's case (a prefetch's "the seeding waited for
this slot's mover" outliving later movers on the slot), round 4's rule that a slot's "free" event stands for ALL movers that read
it, round 5's turn-taking of the BER trials' generator buffers and of the transmitter's data-bit buffers -- all must be FOUND; the scheduler as it stands must come
through >= 10 000 sequences clean, under AddressSanitizer + UndefinedBehaviorSanitizer (leaks included) and under ThreadSanitizer."""
import json
import subprocess

import pytest

from conftest import ROOT

SRC = [str(ROOT / "tests" / "sched_model" / "model.cpp"), str(ROOT / "tests" / "sched_model" / "driver.cpp")]
TAPS = str(ROOT / "basebandboard_amd" / "data" / "lutopt_256.taps")


BUILDS = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
    "tsan": ["-fsanitize=thread"],
}

# MUTANTS: the scheduler with ONE of its ordering rules taken out.  Until round 4 these were #ifdef holes in the product source
# (BBB_SCHED_MODEL_REVERT_*); now the product file carries no test macro and the rule is removed from a COPY of it, by exact text:
# every (old, new) pair must match exactly once -- a rule that is reworded in bbb_api.hip fails the test until its mutant follows.
MUTANTS = {
    # round 3's race (commit f1ee557's scheduler in that respect): an announcement that was never taken leaves its seeding on one
    # arithmetic stream, the next seeding goes to the other
    "untaken_hint": [("""    if (pf.seeded) BBB_HIP(hipStreamWaitEvent(side, pf.seeded, 0));
    else BBB_HIP(hipEventCreateWithFlags(&pf.seeded, hipEventDisableTiming));
""", """    if (!pf.seeded) BBB_HIP(hipEventCreateWithFlags(&pf.seeded, hipEventDisableTiming));
""")],
    # the round-2 advisor's case: a prefetch's "the seeding waited for this slot's mover" outliving later movers on the slot
    "stale_skip": [("h->pf_waited_slot == slot && h->pf_waited_gen == h->stage_gen[slot];", "h->pf_waited_slot == slot;")],
    # round 4: a slot's "free" event stands for ALL movers that read it, also one on the caller's other stream
    "mover_chain": [("""    if (h->stage_busy[slot]) BBB_HIP(hipStreamWaitEvent(ms, h->stage_free[slot], 0));
    hipEvent_t m0 = nullptr, m1 = nullptr;""", """    hipEvent_t m0 = nullptr, m1 = nullptr;""")],
    # round 5: a BER trial's generator buffers are taken in turn; the seeding of trial s + 2 must wait for the kernel of trial s
    "ber_buffer_reuse": [("""            if (h->bs_pending[sb]) BBB_HIP(hipStreamWaitEvent(ss, h->bs_read[sb], 0));    // the trial before last read this pair
""", "")],
    # round 5: the transmitter's data bits take the staging slot's two bit buffers in turn INSTEAD of waiting for the slot's last mover;
    # without the turn the bits of the slot's next kernel are written while that mover still reads them
    "tx_bits_turn": [("bits_buf = h->mbits_turn[slot] ^= 1u;", "bits_buf = 0;")],
}


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    """the two sanitizer builds of the file as it is and one plain build per mutant, compiled side by side"""
    d = tmp_path_factory.mktemp("sched_model")
    api = (ROOT / "basebandboard_amd" / "csrc" / "bbb_api.hip").read_text()
    csrc = str(ROOT / "basebandboard_amd" / "csrc")
    procs = {}
    for name, flags in BUILDS.items():
        cmd = ["g++", "-std=c++17", "-O1", "-g", *flags, "-I", str(ROOT / "tests" / "sched_model"), "-x", "c++",
               str(ROOT / "basebandboard_amd" / "csrc" / "bbb_api.hip"), "-x", "none", *SRC, "-o", str(d / name), "-ldl", "-lpthread"]
        procs[name] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for name, edits in MUTANTS.items():
        text = api
        for old, new in edits:
            assert text.count(old) == 1, f"mutant {name}: its rule occurs {text.count(old)} times in bbb_api.hip (expected once)"
            text = text.replace(old, new)
        # (the copy lives elsewhere: its quoted includes are found through -I csrc)
        src = d / f"bbb_api_{name}.cpp"
        src.write_text(text)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-I", str(ROOT / "tests" / "sched_model"), "-I", csrc, str(src), *SRC, "-o", str(d / name), "-ldl", "-lpthread"]
        procs[name] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for name, pr in procs.items():
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, (name, err[-4000:])
    return {name: d / name for name in procs}


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
    ("untaken_hint", "all", "seeding"),
    ("untaken_hint", "hints", "seeding"),
    ("stale_skip", "hints", "awgn256_planes_kernel"),
    ("mover_chain", "all", "awgn256_planes_kernel"),
    ("ber_buffer_reuse", "all", "seed_"),
    ("tx_bits_turn", "all", "memset"),
])
def test_the_model_finds_the_races_of_rounds_two_and_three(exes, macro, mode, what):
    """The scheduler with one of its ordering rules taken out: the model must report unordered accesses, and of the kind the
    rule is about (two seedings writing the same start-state buffers from different streams; a sample kernel overwriting a
    staging slot its mover still reads -- rounds 2 and 3; round 4, the mover on the caller's stream: a sample kernel
    overwriting a slot that a mover on the caller's OTHER stream still reads after a re-bind the caller did not order)."""
    r, out = run(exes[macro], 6000 if macro == "mover_chain" else 3000, 1, mode, max_bad=3)
    assert r.returncode == 1 and out["sequences_with_unordered_access"] > 0
    assert "UNORDERED ACCESS" in r.stderr and what in r.stderr


def test_model_reports_the_use_of_a_destroyed_stream(tmp_path):
    """Round 4's crash was not an ordering fault but a LIFETIME fault: a process-wide cache remembered a hipStream_t and
    synchronised with it after its owner had destroyed it.  The model's streams now carry a lifetime -- hipStreamDestroy poisons
    the handle, every later use is a report -- and tests/sched_model/lifetime_check.cpp holds it to that bug's shape: the
    stream-remembering cache must be reported on every kind of use, the event-based one that replaced it must not."""
    exe = tmp_path / "lifetime_check"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", str(ROOT / "tests" / "sched_model"), str(ROOT / "tests" / "sched_model" / "model.cpp"),
                        str(ROOT / "tests" / "sched_model" / "lifetime_check.cpp"), "-o", str(exe), "-ldl", "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok lifetimes"), (out.stdout + out.stderr)[-3000:]
