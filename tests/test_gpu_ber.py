"""Fused BER trial kernel vs the oracle restatement (counters are integers: exact)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def oracle_trial(oracle, t, init=1):
    m = oracle.Lutopt(path=oracle.data_path(256))
    return m.ber_trial(init, t.prbs_k, t.prbs_state, t.amp, t.noise_var, t.warmup, t.first_bit, t.nbits)


@pytest.mark.parametrize("k", (7, 9, 11, 15, 20, 23, 31))
def test_trial_matches_oracle_all_prbs(gpu, oracle, k):
    u = gpu.LUTOPT.shipped(256)
    t = gpu.Trial(nbits=200_001, amp=100, noise_var=8, prbs_k=k)
    (bits, errs), = gpu.run_trials(u, [t])
    assert (bits, errs) == oracle_trial(oracle, t)
    assert bits == t.nbits and errs > 0


@pytest.mark.parametrize("amp,nv", [(0, 15), (64, 8), (181, 16 - 1), (300, 9), (2047, 15), (1900, 15), (500, 0), (1, 1)])
def test_trial_channel_parameters(gpu, oracle, amp, nv):
    """Includes parameter pairs whose 12-bit sums wrap (tx.py:80-81): several decision thresholds."""
    u = gpu.LUTOPT.shipped(256)
    t = gpu.Trial(nbits=150_000, amp=amp, noise_var=nv, prbs_k=31, warmup=16)
    (bits, errs), = gpu.run_trials(u, [t])
    assert (bits, errs) == oracle_trial(oracle, t)


@pytest.mark.parametrize("nbits,first,warm", [(1, 0, 0), (2, 5, 16), (63, 0, 16), (131073, 999_999, 16), (400_000, 1_500_000, 3)])
def test_trial_offsets_and_sizes(gpu, oracle, nbits, first, warm):
    u = gpu.LUTOPT.shipped(256, init=0xABCDEF0123456789)
    t = gpu.Trial(nbits=nbits, amp=90, noise_var=7, prbs_k=23, prbs_state=0x1234, warmup=warm, first_bit=first)
    (bits, errs), = gpu.run_trials(u, [t])
    assert (bits, errs) == oracle_trial(oracle, t, init=0xABCDEF0123456789)


def fast_trials(oracle, init, ts):
    """[(bits, errors)] of trials that share one noise / PRBS stream, from the oracle's BULK paths -- the byte-table sample stream and
    the word-parallel PRBS, both pinned against the literal restatements in tests/test_oracle.py -- and the channel of tx.py:75-81 /
    rx.py:29 in numpy: millions of bits per second where the literal bbo_ber_trial does one (it is held to that one below, at a
    small size)."""
    t = ts[0]
    m = oracle.Lutopt(path=oracle.data_path(256))
    g = m.awgn(init, t.warmup + t.first_bit, t.nbits, fast=True).astype(np.int32)
    _, s0 = oracle.prbs_packed(t.prbs_k, t.first_bit, state=t.prbs_state, fast=True)
    words, _ = oracle.prbs_packed(t.prbs_k, t.nbits, state=s0, fast=True)
    bit = np.unpackbits(words.view(np.uint8), bitorder="little")[: t.nbits].astype(np.int32)
    w12 = lambda v: ((v + 2048) & 4095) - 2048
    out = []
    for x_ in ts:
        x = w12(np.where(bit == 1, x_.amp, -x_.amp) + w12(g * x_.noise_var))
        out.append((t.nbits, int(np.count_nonzero((x >= 0).astype(np.int32) != bit))))
    return out


@pytest.mark.parametrize("nbits,k,first", [(9_000_001, 31, 0), (4_200_000, 9, 123_457), (30_000_000, 23, 5_000_003)])
def test_generators_beyond_the_head_of_the_seeding(gpu, oracle, nbits, k, first):
    """Round 5's two-launch seeding: the first 65536 start states come from seed_head_kernel, everything above from the tail kernel's
    top-level tables (d = 1, 2, ... per 65536 generators), which also writes the bit planes; the PRBS states come from ONE state per
    consumer lane and 31 steps with B^64.  Trials of 65 626 ... 468 750 generators (segments of 64 bits), all settings of a group,
    against the oracle's sequential pass."""
    u = gpu.LUTOPT.shipped(256, init=0x1234_5678_9ABC_DEF0_0FED_CBA9)
    ts = [gpu.Trial(nbits=nbits, amp=a, noise_var=8, prbs_k=k, prbs_state=5, first_bit=first) for a in (91, 128, 181)]
    init = 0x1234_5678_9ABC_DEF0_0FED_CBA9
    got = gpu.run_trials(u, ts)
    assert got == fast_trials(oracle, init, ts)
    # the same trials again: the library hands back the start states it kept (no seeding), same counters
    assert gpu.run_trials(u, ts) == got
    # (the bulk pricing against the literal restatement of the whole trial)
    small = gpu.Trial(nbits=60_001, amp=128, noise_var=8, prbs_k=k, prbs_state=5, first_bit=1000)
    m = oracle.Lutopt(path=oracle.data_path(256))
    assert fast_trials(oracle, init, [small]) == [m.ber_trial(init, k, 5, 128, 8, 16, 1000, 60_001)]


def test_sweep_multi_groups_on_one_device(gpu, oracle):
    """BBB_SHARD_GROUPS through bbb_ber_sweep_multi on one device (BASELINE configs[4]'s shape in small: 3 seeds x 4 points, the seeds
    as stretches of the one cycle 2^48 apart): every group one pass of its own noise stream, counters equal bbb_ber_trials' and,
    for the first and the last seed, the oracle's (its jump to 2 x 2^48 is the product's own state_at: the oracle steps)."""
    from basebandboard_amd import _lib
    from basebandboard_amd.channel import sweep_multi
    u = gpu.LUTOPT.shipped(256)
    ts = [gpu.Trial(nbits=300_011, amp=a, noise_var=8, warmup=16 + (s << 48)) for s in range(3) for a in (91, 114, 143, 181)]
    got = sweep_multi([u], ts, mode=_lib.SHARD_GROUPS)
    assert got == gpu.run_trials(u, ts)
    m = oracle.Lutopt(path=oracle.data_path(256))
    assert got[:4] == [m.ber_trial(1, 31, 1, t.amp, 8, 16, 0, t.nbits) for t in ts[:4]]
    far = u.state_at(2 << 48)                       # seed 2 = the reset state 2^49 clocks on
    assert got[8:] == [m.ber_trial(far, 31, 1, t.amp, 8, 16, 0, t.nbits) for t in ts[8:]]


def test_grouped_trials_equal_individual_trials(gpu, oracle):
    """Trials that share one noise/PRBS stream are evaluated in one pass (up to 12 per launch); the
    counters must equal those of the same trials run one at a time, and the oracle's."""
    u = gpu.LUTOPT.shipped(256)
    chan = [(0, 15), (64, 8), (91, 8), (300, 9), (2047, 15), (1900, 15), (500, 0), (1, 1), (181, 15), (128, 8),
            (143, 8), (161, 8), (203, 8), (286, 8)]                       # 14 -> one group of 12 and one of 2
    ts = [gpu.Trial(nbits=120_001, amp=a, noise_var=nv, prbs_k=15, first_bit=777) for a, nv in chan]
    grouped = gpu.run_trials(u, ts)
    single = [gpu.run_trials(u, [t])[0] for t in ts]
    assert grouped == single
    for i in (0, 4, 5, 13):
        assert grouped[i] == oracle_trial(oracle, ts[i])
    # a different stream in the middle splits the groups but changes nothing else
    mixed = ts[:3] + [gpu.Trial(nbits=5000, amp=90, noise_var=8, prbs_k=7)] + ts[3:6]
    res = gpu.run_trials(u, mixed)
    assert res[:3] == grouped[:3] and res[4:] == grouped[3:6]
    assert res[3] == oracle_trial(oracle, mixed[3])


def test_split_trial_sums_to_whole(gpu):
    """Counters are additive over disjoint bit ranges: what sharding across GPUs relies on."""
    u = gpu.LUTOPT.shipped(256)
    whole = gpu.Trial(nbits=3_000_000, amp=110, noise_var=8)
    parts = [gpu.Trial(nbits=1_000_000, amp=110, noise_var=8, first_bit=i * 1_000_000) for i in range(3)]
    res = gpu.run_trials(u, [whole] + parts)
    assert res[0][0] == sum(r[0] for r in res[1:]) == 3_000_000
    assert res[0][1] == sum(r[1] for r in res[1:])


def test_sweep_against_q_function(gpu):
    """Eb/N0 sweep sanity (build-defined layer): BER within 25 % of Q(sqrt(2 Eb/N0)) up to 7 dB
    with 2e7 bits per point; monotone decreasing."""
    from basebandboard_amd import channel
    u = gpu.LUTOPT.shipped(256)
    nv = 8
    trials, dbs = [], []
    for db in range(0, 8):
        amp = channel.amp_for_ebn0(db, nv)
        dbs.append(channel.ebn0_db(amp, nv))
        trials.append(gpu.Trial(nbits=20_000_000, amp=amp, noise_var=nv))
    res = gpu.run_trials(u, trials)
    bers = [e / b for b, e in res]
    assert all(x > y for x, y in zip(bers, bers[1:]))
    for db, ber in zip(dbs, bers):
        assert abs(ber / channel.ber_theory(db) - 1.0) < 0.25, (db, ber, channel.ber_theory(db))


def test_run_trials_into_accumulates(gpu):
    u = gpu.LUTOPT.shipped(256)
    ts = [gpu.Trial(nbits=100_000, amp=80, noise_var=8), gpu.Trial(nbits=50_000, amp=120, noise_var=8)]
    c = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    gpu.run_trials_into(u, ts, c)
    gpu.run_trials_into(u, ts, c)
    torch.cuda.synchronize()
    once = gpu.run_trials(u, ts)
    assert c.cpu().tolist() == [[2 * b, 2 * e] for b, e in once]


def test_trial_errors(gpu):
    u = gpu.LUTOPT.shipped(256)
    with pytest.raises(ValueError, match="invalid for PRBS"):
        gpu.Trial(nbits=10, amp=1, noise_var=1, prbs_k=8)
    with pytest.raises(ValueError):
        gpu.run_trials(u, [gpu.Trial(nbits=10, amp=5000, noise_var=1)])
    with pytest.raises(ValueError):
        gpu.run_trials(u, [gpu.Trial(nbits=10, amp=10, noise_var=16)])
    assert gpu.run_trials(u, []) == []


def test_two_scan_form_of_the_grouped_kernel(oracle):
    """The grouped kernel has two forms: one scan per setting on X = bit ? ~T : T (taken whenever the two
    thresholds mirror each other, i.e. for every non-wrapping channel) and the plain two-scan form, which
    BBB_BER_NO_FAST=1 forces in the experiments build of the library (libbbb_hip_exp.so, -DBBB_EXPERIMENTS; the
    product build ignores the environment).  Same counters, in child processes because the switch is read once."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import json, basebandboard_amd as g\n"
        "g._lib.select_build('experiments')\n"
        "u = g.LUTOPT.shipped(256)\n"
        "ts = [g.Trial(nbits=300_001, amp=a, noise_var=nv, prbs_k=k, warmup=16) for a, nv, k in"
        " [(100, 8, 31), (91, 8, 31), (37, 3, 31), (250, 15, 31), (64, 8, 31), (1, 1, 31), (77, 7, 31)]]\n"
        "print(json.dumps(g.run_trials(u, ts)))\n")
    outs = []
    for env in ({}, {"BBB_BER_NO_FAST": "1"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
    m = oracle.Lutopt(path=oracle.data_path(256))
    for (bits, errs), (a, nv) in zip(outs[0], [(100, 8), (91, 8), (37, 3), (250, 15), (64, 8), (1, 1), (77, 7)]):
        assert (bits, errs) == m.ber_trial(1, 31, 1, a, nv, 16, 0, 300_001)


def test_cpp_caller_of_the_c_abi(gpu, oracle):
    """examples/bbb_mc: no Python, no torch -- library, hipMalloc and printf.  Its counters for a small sweep
    equal the oracle's; its loopback line reports a clean stream."""
    import subprocess
    from conftest import ROOT
    exe = ROOT / "examples" / "bbb_mc"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "examples")])
    r = subprocess.run([str(exe), "--bits", "200000", "--from", "2", "--to", "6", "--step", "2", "--k", "15", "--nv", "7", "--loopback", "3000000"],
                       cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.splitlines() if l and not l.startswith("#")]
    assert len(rows) == 3
    m = oracle.Lutopt(path=oracle.data_path(256))
    for row in rows:
        amp, bits, errs = int(row[1]), int(row[2]), int(row[3])
        assert (bits, errs) == m.ber_trial(1, 15, 1, amp, 7, 16, 0, 200_000)
    assert "loopback: 3000000 bits, 0 errors" in r.stdout
    # points x seeds: two reset states summed
    r = subprocess.run([str(exe), "--bits", "100000", "--from", "4", "--to", "4", "--k", "15", "--nv", "7", "--init", "a5", "--seeds", "2"],
                       cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    row = [l.split() for l in r.stdout.splitlines() if l and not l.startswith("#")][0]
    amp = int(row[1])
    # seed d = the reset state advanced 2^48 d clocks (disjoint stretches of the one cycle), not init + d
    seed1 = gpu.LUTOPT.shipped(256, init=0xa5).state_at(1 << 48)
    e = [m.ber_trial(s, 15, 1, amp, 7, 16, 0, 100_000) for s in (0xa5, seed1)]
    assert (int(row[2]), int(row[3])) == (e[0][0] + e[1][0], e[0][1] + e[1][1])


def test_sweep_multi_on_one_device_equals_ber_trials(gpu):
    """bbb_ber_sweep_multi (host thread per device, ncclCommInitAll, ONE ncclAllReduce of uint64 counters) with
    ndev = 1: RCCL is loaded and the collective runs; all three sharding modes return bbb_ber_trials' counters
    bit for bit."""
    from basebandboard_amd import _lib
    from basebandboard_amd.channel import sweep_multi
    u = gpu.LUTOPT.shipped(256)
    trials = [gpu.Trial(nbits=2_000_003, amp=gpu.channel.amp_for_ebn0(db, 8), noise_var=8) for db in range(0, 11, 2)]
    trials += [gpu.Trial(nbits=77_777, amp=40, noise_var=5, prbs_k=9, first_bit=123)]
    want = gpu.run_trials(u, trials)
    assert all(b == t.nbits for (b, _), t in zip(want, trials)) and want[0][1] > 0
    for mode in (_lib.SHARD_TRIALS, _lib.SHARD_SEEDS, _lib.SHARD_BITS):
        assert sweep_multi([u], trials, mode) == want
    assert sweep_multi([u], trials) == want                  # cached communicator, second use
    with pytest.raises(ValueError):
        sweep_multi([u, u], trials)                          # two handles on one device
    assert _lib.lib().bbb_multi_release() == 0
    assert sweep_multi([u], trials) == want                  # communicators are rebuilt on demand


@pytest.mark.parametrize("world", (2, 8))
def test_bit_sliced_shares_sum_to_the_whole_on_the_gpu(gpu, world):
    """The shares bbb_ber_sweep_multi hands to `world` devices (bbb_sweep_shard, BBB_SHARD_BITS), run one after
    the other on this GPU, add up to the undivided trials' counters exactly."""
    from basebandboard_amd import _lib
    from basebandboard_amd.channel import shard_trials
    u = gpu.LUTOPT.shipped(256)
    trials = [gpu.Trial(nbits=3_000_001, amp=gpu.channel.amp_for_ebn0(db, 8), noise_var=8) for db in (0, 3, 6)]
    want = gpu.run_trials(u, trials)
    tot = [[0, 0] for _ in trials]
    for r in range(world):
        for i, (b, e) in enumerate(gpu.run_trials(u, shard_trials(trials, r, world, _lib.SHARD_BITS))):
            tot[i][0] += b
            tot[i][1] += e
    assert [tuple(t) for t in tot] == want


def test_cli_json_multi_and_awgn_modes(gpu, oracle):
    """examples/bbb_mc: --ebn0 A:B:STEP with --json, the bbb_ber_sweep_multi route (--multi 1: host thread per
    device + RCCL all-reduce, here over one device) and the AWGN fill mode -- counters and samples equal the oracle's."""
    import json
    import subprocess
    from conftest import ROOT
    exe = ROOT / "examples" / "bbb_mc"
    subprocess.check_call(["make", "-C", str(ROOT / "examples")], stdout=subprocess.DEVNULL)
    m = oracle.Lutopt(path=oracle.data_path(256))
    outs = []
    for extra in ([], ["--multi", "1"], ["--multi", "1", "--shard", "trials"]):
        r = subprocess.run([str(exe), "--bits", "300000", "--ebn0", "1:7:3", "--prbs", "31", "--nv", "8", "--json", "1",
                            "--loopback", "5000000"] + extra, cwd=str(ROOT), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        pts = [l for l in lines if "ebn0_db" in l]
        assert len(pts) == 3
        for p in pts:
            assert (p["bits"], p["errors"]) == m.ber_trial(1, 31, 1, p["amp"], 8, 16, 0, 300_000)
            assert p["ebn0_db_effective"] <= p["ebn0_db"] + 0.7
        summ = [l for l in lines if l.get("mode") == "ber_sweep"][0]
        assert summ["total_bits"] == 900_000 and summ["gbit_trials_s"] > 0
        assert ("ncclAllReduce" in summ["reduce"]) == bool(extra)
        lb = [l for l in lines if l.get("mode") == "prbs_loopback"][0]
        assert lb["check_errors"] == 0 and lb["detector_errors"] == 0 and lb["bits"] == 5_000_000
        outs.append(pts)
    assert outs[0] == outs[1] == outs[2]
    r = subprocess.run([str(exe), "--nsamples", "3000000", "--steps", "2", "--json", "1"], cwd=str(ROOT), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a = json.loads(r.stdout.strip().splitlines()[-1])
    assert a["mode"] == "awgn_fill" and a["gsample_s"] > 0 and 0 < a["hbm_roofline_frac"] < 1
    assert a["head"] == m.awgn(1, 16, 64, fast=True).tolist() and a["samples_per_launch"] == 3_000_000
    # a size at which the two-kernel form applies: the stream's own level (two reads per sample kernel), explicit levels, and
    # 0 = plain bbb_awgn_fill_i8 calls in the one-kernel form
    nbig = (1 << 24) + 16
    for staged, per in ((None, 2 * nbig), (2, 2 * nbig), (1, nbig), (0, nbig)):
        r = subprocess.run([str(exe), "--nsamples", str(nbig), "--steps", "4", "--json", "1"] + ([] if staged is None else ["--staged", str(staged)]), cwd=str(ROOT),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        a = json.loads(r.stdout.strip().splitlines()[-1])
        assert a["samples_per_launch"] == per and a["gsample_s"] > 0 and a["head"] == m.awgn(1, 16, 64, fast=True).tolist()
    r = subprocess.run([str(exe), "--gpus", "64"], cwd=str(ROOT), capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "visible" in r.stderr
    # --shard groups: BASELINE configs[4]'s shape as ONE call -- the sweep once per seed (stretches of the one cycle 2^48 apart) as groups
    # that stay on one device; the rows are the sums over the seeds, here three seeds on the one device, against the oracle
    r = subprocess.run([str(exe), "--bits", "200000", "--ebn0", "2:6:2", "--json", "1", "--multi", "1", "--shard", "groups", "--seeds", "3"],
                       cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]      # (RCCL prints its banner to stdout)
    pts = [l for l in lines if "ebn0_db" in l]
    u = gpu.LUTOPT.shipped(256)
    for p in pts:
        want = [m.ber_trial(u.state_at(s << 48) if s else 1, 31, 1, p["amp"], 8, 16, 0, 200_000) for s in range(3)]
        assert (p["bits"], p["errors"]) == (sum(b for b, _ in want), sum(e for _, e in want))
    summ = [l for l in lines if l.get("mode") == "ber_sweep"][0]
    assert summ["seeds"] == 3 and summ["shard"] == "groups" and summ["equals_single_device_counters"] is True and summ["total_bits"] == 3 * 3 * 200_000


def test_thread_per_device_body_rehearsed_on_one_gpu():
    """bbb_ber_sweep_multi runs one host thread per device; a one-GPU box has never run more than `work(0)` on the calling
    thread.  The experiments build (libbbb_hip_exp.so) with BBB_MULTI_REHEARSAL=1 accepts handles that share a device and
    replaces the RCCL all-reduce by a host-side sum: 2 and 8 host threads inside ber_run at once (plan caches, per-device
    statics, per-thread error text), in all three sharding modes -- the totals must equal the same trials run undivided.
    A child process: the product build refuses (and must keep refusing) two handles on one device."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import json, torch, basebandboard_amd as g\n"
        "g._lib.select_build('experiments')\n"
        "from basebandboard_amd.channel import sweep_multi\n"
        "from basebandboard_amd import _lib\n"
        "base = g.LUTOPT.shipped(256)\n"
        "ts = [g.Trial(nbits=2_000_003, amp=g.channel.amp_for_ebn0(db, 8), noise_var=8) for db in (0, 2, 4, 6, 8)] + [g.Trial(nbits=77_777, amp=40, noise_var=5, prbs_k=9, first_bit=123)]\n"
        "want = g.run_trials(base, ts)\n"
        "res = {'want': want}\n"
        "for nd in (2, 8):\n"
        "    same = [g.LUTOPT.shipped(256) for _ in range(nd)]\n"
        "    res[f'bits{nd}'] = sweep_multi(same, ts, _lib.SHARD_BITS)\n"
        "    res[f'trials{nd}'] = sweep_multi(same, ts, _lib.SHARD_TRIALS)\n"
        "    seeds = [g.LUTOPT.shipped(256, init=base.state_at(d << 48)) for d in range(nd)]\n"
        "    got = sweep_multi(seeds, ts, _lib.SHARD_SEEDS)\n"
        "    tot = [[0, 0] for _ in ts]\n"
        "    for u in seeds:\n"
        "        for i, (b, e) in enumerate(g.run_trials(u, ts)):\n"
        "            tot[i][0] += b; tot[i][1] += e\n"
        "    res[f'seeds{nd}'] = [got, tot]\n"
        "    res[f'device{nd}'] = torch.cuda.current_device()\n"
        "print(json.dumps(res))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), env={**os.environ, "BBB_MULTI_REHEARSAL": "1"}, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for nd in (2, 8):
        assert res[f"bits{nd}"] == res["want"], nd
        assert res[f"trials{nd}"] == res["want"], nd
        got, tot = res[f"seeds{nd}"]
        assert got == tot and got != res["want"]
        assert res[f"device{nd}"] == 0


# ---- a trial group continued over several calls (bbb_ber_run_*) -------------------------------------------------

@pytest.mark.parametrize("k,nper,m,first", [(31, 100_003, 4, 0), (7, 12_345, 3, 77), (20, 70_000, 1, 5), (23, 262_144, 8, 1 << 20)])
def test_continued_trials_equal_one_trial_over_the_block(gpu, oracle, k, nper, m, first):
    """After every m-th call the totals of a continued run are those of ONE trial over the blocks so far -- against the
    oracle (sequential CPU restatement) and against bbb_ber_trials on the same ranges."""
    u = gpu.LUTOPT.shipped(256, init=0x1234567)
    settings = [(100, 8), (64, 8), (37, 3)]
    ts = [gpu.Trial(nbits=nper, amp=a, noise_var=nv, prbs_k=k, prbs_state=5, warmup=16, first_bit=first) for a, nv in settings]
    m_or = oracle.Lutopt(path=oracle.data_path(256))
    with gpu.ContinuedTrials(u, ts, m) as run:
        seen = []
        for call in range(2 * m):
            tot = run.next()
            seen.append(tot)
            if (call + 1) % m == 0:
                nb = (call + 1) * nper
                for (bits, errs), (a, nv) in zip(tot, settings):
                    assert (bits, errs) == m_or.ber_trial(0x1234567, k, 5, a, nv, 16, first, nb)
                whole = gpu.run_trials(u, [gpu.Trial(nbits=nb, amp=a, noise_var=nv, prbs_k=k, prbs_state=5, warmup=16, first_bit=first)
                                           for a, nv in settings])
                assert tot == whole
        assert run.tell() == (2 * m, first + 2 * m * nper)
        # between block ends every call adds about nper bits, and the counters never go down
        for prev, cur in zip(seen, seen[1:]):
            for (b0, e0), (b1, e1) in zip(prev, cur):
                assert 0 < b1 - b0 <= nper + 2 * m * 64 and e1 >= e0


def test_continued_trials_leave_the_handle_alone(gpu, oracle):
    """Other calls on the handle between two calls of a run (a fill, other trials, a staged stream) do not disturb it: the run
    owns its state buffers."""
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    t = gpu.Trial(nbits=50_000, amp=90, noise_var=8)
    m_or = oracle.Lutopt(path=oracle.data_path(256))
    c = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
    with gpu.ContinuedTrials(u, [t], 3) as run:
        for call in range(3):
            run.next_into(c)
            g.generate(100_000, first_step=123)
            gpu.run_trials(u, [gpu.Trial(nbits=33_333, amp=50, noise_var=5, first_bit=9)])
        torch.cuda.synchronize()
    assert tuple(c.cpu().tolist()[0]) == m_or.ber_trial(1, 31, 1, 90, 8, 16, 0, 150_000)


def test_continued_trials_reject_mixed_groups(gpu):
    u = gpu.LUTOPT.shipped(256)
    with pytest.raises(ValueError):
        gpu.ContinuedTrials(u, [gpu.Trial(nbits=10, amp=10, noise_var=1), gpu.Trial(nbits=11, amp=10, noise_var=1)], 2)
    with pytest.raises(ValueError):
        gpu.ContinuedTrials(u, [gpu.Trial(nbits=10, amp=10, noise_var=1)], 0)
    with pytest.raises(ValueError):       # a wrap-around setting cannot share a launch
        gpu.ContinuedTrials(u, [gpu.Trial(nbits=1000, amp=1900, noise_var=15), gpu.Trial(nbits=1000, amp=10, noise_var=1)], 2)
    with gpu.ContinuedTrials(u, [gpu.Trial(nbits=1000, amp=1900, noise_var=15)], 2) as run:      # alone it runs (general kernel)
        assert run.next()[0][0] > 0


def test_trials_queued_back_to_back_equal_trials_run_one_by_one(gpu, oracle):
    """bbb_ber_trials_dev does not synchronise: a caller that queues trials back to back has the start states of trial s + 1
    derived (on internal streams, into the second set of buffers) while the kernel of trial s runs.  Seven trials of different
    positions, lengths and PRBS in one go, then each alone with a synchronisation in between, then the oracle; with a fill and
    a staged stream read in between, which use the same second set of start-state buffers."""
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    ts = [gpu.Trial(nbits=3_000_000 + 777 * i, amp=60 + 9 * i, noise_var=8, first_bit=(i * 37) % 5 * 1_000_003, prbs_k=(31, 23, 9)[i % 3],
                    warmup=16 + i) for i in range(7)]
    c = torch.zeros((len(ts), 2), dtype=torch.int64, device="cuda")
    for i, t in enumerate(ts):
        gpu.run_trials_into(u, [t], c[i:i + 1])
        if i == 2:
            g.generate(200_000, first_step=5)
        if i == 4:
            with g.stream((1 << 24) + 4096, first_step=99) as st:
                st.next()
    torch.cuda.synchronize()
    got = [tuple(x) for x in c.cpu().tolist()]
    v = gpu.LUTOPT.shipped(256)
    alone = []
    for t in ts:
        alone.append(gpu.run_trials(v, [t])[0])
        torch.cuda.synchronize()
    assert got == [tuple(x) for x in alone]
    m_or = oracle.Lutopt(path=oracle.data_path(256))
    t = ts[3]
    assert got[3] == m_or.ber_trial(1, t.prbs_k, 1, t.amp, t.noise_var, t.warmup, t.first_bit, t.nbits)
