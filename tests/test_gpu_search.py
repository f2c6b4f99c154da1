"""The recurrence search of software/rnghunt on the GPU (bbb_lutopt_search) against the host arithmetic
and the Python restatement: same accepted candidates, same counts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def host_scan(gf2, k, seed, lo, hi):
    full, acc = 0, []
    for c in range(lo, hi):
        rows = gf2.search_candidate(k, seed, c)
        _, deg = gf2.lutopt_charpoly(rows)
        full += deg == k
        if deg == k and gf2.is_full_period(rows):
            acc.append(c)
    return full, acc


@pytest.mark.parametrize("k,seed,n", [(16, 1, 700), (16, 77, 700), (32, 3, 1500), (64, 5, 1500)])
def test_search_finds_the_first_accepted_candidate(gpu, k, seed, n):
    from basebandboard_amd import gf2
    from oracle import gf2poly as og
    full, acc = host_scan(gf2, k, seed, 0, n)
    idx, rows, st = gf2.search(k, seed=seed, first=0, count=n)
    if not acc:
        assert idx is None and st["tested"] == n and st["full_degree"] == full and st["primitive"] == 0
        return
    assert idx == acc[0]
    assert rows == gf2.search_candidate(k, seed, idx) and og.is_full_period(rows)
    # the range before the hit, exhaustively: every candidate examined, none accepted, same degree count
    full0, acc0 = host_scan(gf2, k, seed, 0, idx)
    idx0, _, st0 = gf2.search(k, seed=seed, first=0, count=idx)
    assert idx0 is None and not acc0
    assert st0["tested"] == idx and st0["full_degree"] == full0 and st0["primitive"] == 0
    # and a window starting behind it finds the next one
    if len(acc) > 1:
        idx1, _, _ = gf2.search(k, seed=seed, first=acc[0] + 1, count=n - acc[0] - 1)
        assert idx1 == acc[1]


@pytest.mark.parametrize("k,count", [(128, 4000), (192, 6000), (256, 8000), (512, 20000)])
def test_search_large_orders(gpu, k, count):
    """The orders the reference searched (its tool is set to n = 192, rnghunt.rs:14): whatever comes back has
    been re-checked by the host arithmetic inside the call; here also by the Python restatement, and a
    sample of the rejected candidates before it is confirmed rejected."""
    from basebandboard_amd import gf2
    from oracle import gf2poly as og
    idx, rows, st = gf2.search(k, seed=2017, first=0, count=count)
    assert st["tested"] > 0 and st["full_degree"] <= st["tested"] and st["order_divides"] <= st["full_degree"]
    if idx is None:
        assert st["tested"] == count and st["primitive"] == 0
        pytest.skip(f"no full-period matrix among {count} candidates for k={k}: {st}")
    p, L = og.lutopt_charpoly(rows)
    assert L == k and og.is_primitive(p)
    rng = np.random.default_rng(k)
    for c in rng.integers(0, idx, size=min(idx, 25)):
        assert not gf2.is_full_period(gf2.search_candidate(k, 2017, int(c)))
    if k & (k - 1):
        return                                  # (generator handles exist for power-of-two orders only: the CLT tree)
    # usable as a generator: full linear complexity of bit 0
    import basebandboard_amd as bbb
    u = bbb.LUTOPT.from_packed(rows, init=(1 << k) - 1, device=-1)
    bits = [u.state_at(t + 1) & 1 for t in range(2 * k)]
    assert gf2.berlekamp_massey(bits)[0] == k


def test_search_errors(gpu):
    from basebandboard_amd import gf2
    with pytest.raises(Exception, match="search supports|no factorisation"):
        gf2.search(20, count=10)
    assert gf2.search(16, count=0)[0] is None


@pytest.mark.parametrize("k", (32, 128))
def test_found_matrix_gets_its_own_kernel(gpu, oracle, tmp_path, k):
    """search -> specialise -> generate: a matrix that the search returns has no shipped kernel; LUTOPT.specialise
    generates the straight-line network for it, compiles it with hipcc and attaches it.  The stream must be the
    table-driven kernel's and the oracle's; and it is an order of magnitude faster."""
    import shutil
    import time
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this machine")
    from basebandboard_amd import gf2
    idx, rows, _ = gf2.search(k, seed=11, first=0, count=20000)
    assert idx is not None
    import basebandboard_amd as bbb
    u = bbb.LUTOPT.from_packed(rows, init=(1 << k) - 1)
    g = bbb.CLTGRNG(u)
    n = 3_000_017
    slow = g.generate(n, first_step=9)
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.generate(n, first_step=9 + n); torch.cuda.synchronize(); t_slow = time.perf_counter() - t0
    u.specialise(build_dir=tmp_path)
    fast = g.generate(n, first_step=9)
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.generate(n, first_step=9 + n); torch.cuda.synchronize(); t_fast = time.perf_counter() - t0
    assert torch.equal(slow, fast)
    m = oracle.Lutopt(packed=rows)
    assert np.array_equal(fast[:200_000].cpu().numpy(), m.awgn((1 << k) - 1, 9, 200_000))
    assert t_fast < t_slow / 3
    # a second handle finds the cached library
    u2 = bbb.LUTOPT.from_packed(rows, init=5)
    u2.specialise(build_dir=tmp_path)
    assert np.array_equal(bbb.CLTGRNG(u2).generate(10_000).cpu().numpy(), m.awgn(5, 0, 10_000))


def test_found_k256_matrix_runs_ber_trials_and_tx(gpu, oracle, tmp_path):
    """Search -> use, for the Monte-Carlo loop as well: a k = 256 matrix that is NOT the shipped one gets, through
    LUTOPT.specialise() (= bbb_lutopt_attach_custom_library), its own sample kernel AND its own fused BER kernels;
    counters equal the oracle's for that matrix.  Before, bbb_ber_trials refused every matrix but the shipped n256."""
    bbb = gpu
    # a permuted copy of the shipped matrix: conjugation by a permutation keeps the period, changes every tap list
    import random
    rnd = random.Random(7)
    perm = list(range(256))
    rnd.shuffle(perm)
    inv = [0] * 256
    for i, p in enumerate(perm):
        inv[p] = i
    base = bbb.recurrences.load_packed(bbb.recurrences.matrix_path(256))
    rows = [sorted(perm[c] for c in base[inv[r]]) for r in range(256)]
    assert rows != base
    u = bbb.LUTOPT.from_packed(rows, init=0xABCDEF)
    assert not u.specialised
    t = [bbb.Trial(nbits=200_003, amp=a, noise_var=nv, prbs_k=k) for a, nv, k in ((100, 8, 31), (64, 8, 31), (37, 3, 9))]
    with pytest.raises(bbb._lib.BbbError) as e:
        bbb.run_trials(u, t)
    assert e.value.code == bbb._lib.BBB_EUNSUP
    u.specialise(build_dir=tmp_path)
    m = oracle.Lutopt(packed=rows)
    got = bbb.run_trials(u, t)
    assert got == [m.ber_trial(0xABCDEF, x.prbs_k, 1, x.amp, x.noise_var, 16, 0, x.nbits) for x in t]
    assert np.array_equal(bbb.CLTGRNG(u).generate(100_000, first_step=16).cpu().numpy(), m.awgn(0xABCDEF, 16, 100_000))
    # full groups: the run-time library holds the short instance list (Fast<12> serves 5 ... 12 settings), i.e. the kernel
    # with seven resident settings and FIVE streamed through the two scalar-load windows, around a network the shipped
    # library's tests never see.  Twelve, eleven and five settings on one stream, against the oracle.
    amps = (91, 102, 114, 128, 143, 161, 181, 203, 228, 255, 287, 322)
    for n in (12, 11, 5):
        grp = [bbb.Trial(nbits=300_007, amp=a, noise_var=8, prbs_k=31) for a in amps[:n]]
        assert bbb.run_trials(u, grp) == [m.ber_trial(0xABCDEF, 31, 1, x.amp, 8, 16, 0, x.nbits) for x in grp], n


def test_cli_search_writes_the_reference_format_and_the_file_loads(gpu, oracle, tmp_path):
    """examples/bbb_mc --search K --seed S --out FILE: the reference's rnghunt loop (software/rnghunt/src/bin/rnghunt.rs:13-66)
    around bbb_lutopt_search -- windows of candidates until one is accepted, the matrix written in the `out` format
    (rnghunt.rs:51-53), read back and re-checked; the file then drives --matrix, and its first samples equal the oracle's on
    that matrix."""
    import json
    import subprocess
    from conftest import ROOT
    exe = ROOT / "examples" / "bbb_mc"
    subprocess.check_call(["make", "-C", str(ROOT / "examples")], stdout=subprocess.DEVNULL)
    out = tmp_path / "out"
    r = subprocess.run([str(exe), "--search", "32", "--seed", "5", "--count", "4096", "--out", str(out)], cwd=str(ROOT),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["mode"] == "search" and rec["k"] == 32 and rec["full_period"] is True and rec["reloaded_k"] == 32 and rec["tested"] > 0
    rows = out.read_text().split()
    assert len(rows) == 32 and all(len(x) == 32 and set(x) <= {"0", "1"} and 3 <= x.count("1") <= 4 for x in rows)
    # the same candidate through the library call, and the file as a generator
    idx, packed, _ = gpu.gf2.search(32, seed=5, first=0, count=rec["candidate"] + 1)
    assert idx == rec["candidate"]
    assert [[c for c, ch in enumerate(x) if ch == "1"] for x in rows] == [sorted(p) for p in packed]
    r = subprocess.run([str(exe), "--matrix", str(out), "--nsamples", "100000", "--steps", "1", "--json", "1"], cwd=str(ROOT),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a = json.loads(r.stdout.strip().splitlines()[-1])
    m = oracle.Lutopt(path=str(out))
    assert a["head"] == m.awgn(1, 16, 64).tolist()
