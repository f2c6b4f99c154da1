"""LUTOPT + CLTGRNG kernels vs the oracle and the golden vectors (through the C ABI).  Bit exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_stream(oracle, n, init, first, count, fast=True):
    m = oracle.Lutopt(path=oracle.data_path(n))
    return m.awgn(init, first, count, fast=fast and n == 256)


def test_n256_golden_prefix(gpu, golden_lutopt):
    """First 4096 samples from init=1 equal the reference's embedded models (rng.py:134-135,173-181)."""
    u = gpu.LUTOPT.shipped(256)
    assert u.specialised
    g = gpu.CLTGRNG(u)
    got = g.generate(4096).cpu().numpy()
    assert got.tolist() == golden_lutopt["256"]["clt_out"]
    assert u.state_at(1) == int(golden_lutopt["256"]["states_hex"][0], 16)
    assert u.state_at(4096) == int(golden_lutopt["256"]["state_last_hex"], 16)


def test_n256_second_seed(gpu, golden_lutopt):
    g2 = golden_lutopt["256_seed2"]
    u = gpu.LUTOPT.shipped(256, init=int(g2["init"], 16))
    got = gpu.CLTGRNG(u).generate(512).cpu().numpy()
    assert got.tolist() == g2["clt_out"]
    assert u.state_at(512) == int(g2["state_last_hex"], 16)


@pytest.mark.parametrize("nsamples,first", [(1, 0), (15, 0), (16, 7), (17, 0), (63, 0), (64, 16), (65, 16),
                                            (4097, 3), (131072, 16), (1_000_003, 16), (3_000_000, 1_234_567)])
def test_n256_matches_oracle(gpu, oracle, nsamples, first):
    u = gpu.LUTOPT.shipped(256)
    got = gpu.CLTGRNG(u).generate(nsamples, first_step=first).cpu().numpy()
    exp = ref_stream(oracle, 256, 1, first, nsamples)
    assert np.array_equal(got, exp)


def test_n256_far_offset(gpu, oracle):
    """first_step far beyond what the sequential oracle can reach: jump-ahead composes
    (state_at(a + b) from a handle seeded with state_at(a)), and the stream from there matches
    the oracle started at that state."""
    u = gpu.LUTOPT.shipped(256)
    far = 10**15 + 12345
    s_far = u.state_at(far)
    assert gpu.LUTOPT.shipped(256, init=u.state_at(10**15)).state_at(12345) == s_far
    m = oracle.Lutopt(path=oracle.data_path(256))
    assert m.run_int(u.state_at(10**15 + 12000), 345) == s_far
    got = gpu.CLTGRNG(u).generate(500_000, first_step=far).cpu().numpy()
    assert np.array_equal(got, m.awgn(s_far, 0, 500_000, fast=True))


def test_n256_large_seams_and_checksum(gpu, oracle):
    """2^28 samples: every generator segment start (seam) and a strided sample of whole segments
    are compared with the sequential oracle; plus determinism (same call twice, identical bytes)."""
    n = 1 << 28
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    a = g.generate(n, first_step=16)
    b = g.generate(n, first_step=16).clone()
    assert torch.equal(a, b)
    m = oracle.Lutopt(path=oracle.data_path(256))
    a_np = a.cpu().numpy()
    head = m.awgn(1, 16, 200_000, fast=True)
    assert np.array_equal(a_np[:200_000], head)
    rng = np.random.default_rng(0)
    for off in list(rng.integers(0, n - 5000, size=40)) + [n - 5000]:
        exp = m.awgn(u.state_at(16 + int(off)), 0, 5000, fast=True)
        assert np.array_equal(a_np[off: off + 5000], exp), off
    # moments of the CLT output (rng.py:63-65 / clt-grng-evaluate.py:18-31): mean 0, variance 64
    x = a_np[: 1 << 24].astype(np.float64)
    assert abs(x.mean()) < 0.02 and abs(x.var() - 64.0) < 0.2


@pytest.mark.parametrize("n", (16, 32, 64, 128))
def test_small_matrices_match_golden_and_oracle(gpu, oracle, golden_lutopt, n):
    """The reference's own test matrices (n16 in test_lutopt, n32 in test_cltgrng) on the GPU: the shipped
    n16 ... n128 have generated kernels of their own (csrc/awgn_small.hip)."""
    u = gpu.LUTOPT.shipped(n)
    assert not u.specialised
    g = gpu.CLTGRNG(u)
    gold = golden_lutopt[str(n)]
    got = g.generate(256).cpu().numpy()
    assert got.tolist() == gold["clt_out"]
    for i, h in enumerate(gold["states_hex"][:16]):
        assert u.state_at(i + 1) == int(h, 16)
    got = g.generate(70_001, first_step=2 * int(np.log2(n))).cpu().numpy()
    exp = ref_stream(oracle, n, 1, 2 * int(np.log2(n)), 70_001)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("n", (16, 32, 64, 128))
@pytest.mark.parametrize("nsamples,first,init", [(1, 0, 1), (15, 3, 1), (17, 0, 5), (4097, 1000, 1), (1_000_003, 12, 0xBEEF), (3_000_000, 10**9 + 7, 1)])
def test_small_generated_kernels_ragged(gpu, oracle, n, nsamples, first, init):
    u = gpu.LUTOPT.shipped(n, init=init)
    got = gpu.CLTGRNG(u).generate(nsamples, first_step=first).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(n))
    exp = m.awgn(u.state_at(first), 0, nsamples)
    assert np.array_equal(got, exp)
    assert got.min() >= -(n // 2) and got.max() < n // 2            # a log2(n)-bit signed value (rng.py:78)


@pytest.mark.parametrize("n", (16, 64, 128))
def test_other_small_matrices_take_the_table_driven_kernel(gpu, oracle, n):
    """A matrix that is not the shipped one (here: a candidate of the search) has no generated kernel; the
    table-driven path must give its stream."""
    from basebandboard_amd import gf2
    rows = gf2.search_candidate(n, 3, 11)
    u = gpu.LUTOPT.from_packed(rows, init=(1 << n) - 1)
    got = gpu.CLTGRNG(u).generate(20_011, first_step=5).cpu().numpy()
    m = oracle.Lutopt(packed=rows)
    assert np.array_equal(got, m.awgn((1 << n) - 1, 5, 20_011))


def test_n512_int16(gpu, oracle):
    u = gpu.LUTOPT.shipped(512)
    g = gpu.CLTGRNG(u)
    assert g.dtype == torch.int16
    got = g.generate(20_000, first_step=18).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(512))
    x = u.state_at(18)
    exp = []
    for _ in range(300):
        x = m.step_int(x)
        exp.append(m.clt_wrap(m.clt_tree(x)))
    assert got[:300].tolist() == exp


def test_n256_generic_path_equals_specialised(gpu, oracle):
    """A row-permuted copy of n256 is a different matrix (table-driven kernel); the shipped one
    hits the straight-line kernel.  Both must agree with the oracle on their own matrix."""
    packed = gpu.recurrences.n256
    perm = packed[1:] + packed[:1]
    u = gpu.LUTOPT.from_packed(perm, init=12345)
    assert not u.specialised
    got = gpu.CLTGRNG(u).generate(50_000, first_step=3).cpu().numpy()
    m = oracle.Lutopt(packed=perm)
    assert np.array_equal(got, m.awgn(12345, 3, 50_000))


def test_lutopt_from_matrix_and_errors(gpu):
    a = gpu.LUTOPT.shipped(16).a
    u = gpu.LUTOPT(a, init=1)
    assert u.packed == gpu.recurrences.n16
    v = gpu.LUTOPT.from_packed([[0]] * 24)          # LUTOPT takes any k (rng.py:21-40) ...
    with pytest.raises(ValueError):
        gpu.CLTGRNG(v)                              # ... CLTGRNG a power of two (rng.py:72-76)
    with pytest.raises(ValueError):
        gpu.LUTOPT.from_packed([[0]] * 600)         # beyond the 512 this library is built for
    with pytest.raises(ValueError):
        gpu.LUTOPT.from_packed([[99]] * 16)         # tap out of range
    with pytest.raises(ValueError):
        gpu.LUTOPT.from_packed(gpu.recurrences.n16, init=1 << 16)


def test_extreme_state_wraps_to_minus_128(gpu, golden_lutopt):
    """Bits set exactly on the +1 positions: tree value +128, 8-bit signed output -128 (rng.py:78)."""
    ex = golden_lutopt["256_extreme"]
    x = int(ex["x_hex"], 16)
    words = np.array([[(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64).view(np.int64)
    t = gpu.CLTGRNG.tree(torch.from_numpy(words).cuda(), 256).cpu().numpy()
    assert t[0] == ex["clt_tree"] == 128


def test_clt_tree_random_words(gpu, oracle):
    """software/clt-grng/clt-grng-evaluate.py's experiment: tree of random 256-bit words; compare
    each value with the literal tree and the moments with theory (sigma^2 = 64)."""
    rng = np.random.default_rng(7)
    w = rng.integers(0, 2**63, size=(100_000, 4), dtype=np.int64) | (rng.integers(0, 2, size=(100_000, 4)).astype(np.int64) << 63)
    t = gpu.CLTGRNG.tree(torch.from_numpy(w).cuda(), 256).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(256))
    for i in range(0, 100_000, 997):
        x = sum(int(np.uint64(w[i, q])) << (64 * q) for q in range(4))
        assert t[i] == m.clt_tree(x) == m.clt_popcount(x)
    assert abs(t.mean()) < 0.1 and abs(t.astype(np.float64).var() - 64.0) < 1.0


def test_n256_beyond_4g_samples(gpu, oracle):
    """More than 2^32 samples in one call (64-bit offsets everywhere): seams and the tail against
    the oracle started from jump-ahead states."""
    n = (1 << 32) + 12_345
    u = gpu.LUTOPT.shipped(256)
    a = gpu.CLTGRNG(u).generate(n, first_step=16)
    m = oracle.Lutopt(path=oracle.data_path(256))
    rng = np.random.default_rng(42)
    offs = [0, (1 << 32) - 3000, n - 4000] + [int(x) for x in rng.integers(0, n - 4000, size=12)]
    for off in offs:
        got = a[off: off + 4000].cpu().numpy()
        assert np.array_equal(got, m.awgn(u.state_at(16 + off), 0, 4000, fast=True)), off
    del a
    torch.cuda.empty_cache()


def test_prefetch_hint_changes_nothing_but_timing(gpu, oracle):
    """bbb_awgn_prefetch seeds the announced fill on a side stream; results must be identical, a
    mismatching fill must ignore the hint, and hints may be chained."""
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    m = oracle.Lutopt(path=oracle.data_path(256))
    n = 3_000_000
    ref = [m.awgn(u.state_at(16 + i * n), 0, 50_000, fast=True) for i in range(4)]
    g.prefetch(n, first_step=16)
    outs = []
    for i in range(4):
        x = g.generate(n, first_step=16 + i * n)
        if i < 3:
            g.prefetch(n, first_step=16 + (i + 1) * n)
        outs.append(x[:50_000].cpu().numpy())
    for a, b in zip(outs, ref):
        assert np.array_equal(a, b)
    g.prefetch(n, first_step=999)                      # announced but something else is asked for
    x = g.generate(1000, first_step=5).cpu().numpy()
    assert np.array_equal(x, m.awgn(1, 5, 1000, fast=True))
    y = g.generate(n, first_step=999)[:1000].cpu().numpy()
    assert np.array_equal(y, m.awgn(u.state_at(999), 0, 1000, fast=True))
    # BER trials after a prefetch still see the right planes
    g.prefetch(n, first_step=16)
    t = gpu.Trial(nbits=50_000, amp=100, noise_var=8)
    assert gpu.run_trials(u, [t])[0] == m.ber_trial(1, 31, 1, 100, 8, 16, 0, 50_000)


def test_baseline_configs_2_and_4_at_full_size(gpu, oracle):
    """BASELINE.json configs[1] at its full size, every byte: 1e9 samples of the sequential reference stream
    (init = 1, 16 warm-up steps as the reference's test skips, rng.py:161-162) against the oracle's single
    sequential pass (~25 s of one host core).  Covers every segment seam, every round and the ragged tail.
    The same oracle stream then prices ALL ELEVEN points of configs[3] (Eb/N0 0..10 dB, 1e9 bits each, PRBS-31,
    one sample per bit): the channel of tx.py:75-81 / rx.py:29 evaluated on the host over the joint histogram of
    (oracle sample, oracle PRBS bit) must give the fused kernel's error counts exactly; one of the points is also
    priced sample by sample."""
    n = 1_000_000_000
    u = gpu.LUTOPT.shipped(256)
    got = gpu.CLTGRNG(u).generate(n, first_step=16).cpu().numpy()
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp3 = m.awgn(1, 16, 3 * n, fast=True)          # ONE sequential oracle pass over three reads' worth (~75 s of one host core)
    exp = exp3[:n]
    assert got.shape == exp.shape and np.array_equal(got, exp)
    del got
    # The TIMED path of bench.py, at its size: bbb_awgn_stream_next on a fresh handle -- the two-kernel form, two reads per
    # sample kernel (the stream's own choice of level), every next read announced -- 1e9 samples per read, three reads (a
    # kernel's first and second half and the next kernel's first), every byte against that one oracle pass.
    us = gpu.LUTOPT.shipped(256)
    buf = torch.empty(n, dtype=torch.int8, device="cuda")
    with gpu.CLTGRNG(us).stream(n, first_step=16) as st:
        for r in range(3):
            st.next(out=buf)
            part = buf.cpu().numpy()
            assert np.array_equal(part, exp3[r * n:(r + 1) * n]), r
        assert st.tell() == 16 + 3 * n
    del buf, part, exp3
    # sigma^2 = 2^(log2 n - 2) = 64 (software/clt-grng/clt-grng-evaluate.py:18-21), mean 0
    x = exp[: 50_000_000].astype(np.float64)
    assert abs(x.mean()) < 0.01 and abs(x.var() - 64.0) < 0.1
    del x
    words, _ = oracle.prbs_packed(31, n, fast=True)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)
    nv = 8
    trials = [gpu.Trial(nbits=n, amp=gpu.channel.amp_for_ebn0(db, nv), noise_var=nv, prbs_k=31, warmup=16) for db in range(11)]
    counts = gpu.run_trials(u, trials)
    # joint histogram: hist[b][g + 128] = number of positions with PRBS bit b and CLT sample g
    idx = exp.view(np.uint8) ^ np.uint8(0x80)
    hist = [np.bincount(idx[~bits], minlength=256).astype(np.int64), np.bincount(idx[bits], minlength=256).astype(np.int64)]
    assert hist[0].sum() + hist[1].sum() == n
    g = np.arange(-128, 128, dtype=np.int64)
    wrap12 = lambda v: ((v + 2048) & 4095) - 2048
    for t, (nb, ne) in zip(trials, counts):
        errors = 0
        for b in (0, 1):
            rx = wrap12((t.amp if b else -t.amp) + wrap12(g * nv))     # tx.py:75-81
            errors += int(hist[b][(rx >= 0) != bool(b)].sum())          # rx.py:29
        assert (nb, ne) == (n, errors), (t.amp, ne, errors)
        # the Gaussian prediction for this integer channel (the CLT tails are slightly lighter)
        assert 0.8 < (ne / nb) / gpu.channel.ber_lattice(t.amp, nv) < 1.05, (t.amp, ne / nb, gpu.channel.ber_lattice(t.amp, nv))
    # one point sample by sample as well (12-bit registers written out)
    t, (nb, ne) = trials[5], counts[5]
    noise = exp.astype(np.int16) * np.int16(nv)
    lvl = np.where(bits, np.int16(t.amp), np.int16(-t.amp))
    rx = ((lvl + noise + 2048) & 4095) - 2048
    assert ne == int(np.count_nonzero((rx >= 0) != bits))


@pytest.mark.parametrize("n", (16, 4096, 1_000_000 - 64, 1_000_003))
def test_int16_form_of_the_n256_stream(gpu, oracle, n):
    """bbb_awgn_fill_i16 on the k = 256 generator: the same samples, sign-extended (multiples of 16 take the
    generated int8 kernel plus a widening pass, other lengths the table-driven kernel)."""
    import ctypes as C
    from basebandboard_amd import _lib
    u = gpu.LUTOPT.shipped(256, init=0x1234567)
    out = torch.full((n + 32,), 777, dtype=torch.int16, device="cuda")
    _lib.check(_lib.lib().bbb_awgn_fill_i16(u._h, C.c_void_p(out.data_ptr()), n, 40), "bbb_awgn_fill_i16")
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp = m.awgn(0x1234567, 40, n, fast=True).astype(np.int16)
    got = out.cpu().numpy()
    assert np.array_equal(got[:n], exp)
    assert (got[n + 16:] == 777).all()                     # nothing written beyond the rounded-up length


@pytest.mark.parametrize("nsamples,first", [(1, 0), (7, 3), (8, 0), (9, 16), (63, 1), (64, 0), (65, 5), (4099, 18),
                                            (1_000_003, 18), (3_000_000, 2_345_678)])
def test_n512_generated_kernel_matches_oracle(gpu, oracle, nsamples, first):
    """The shipped n512 matrix (software/rnghunt/matrices/512) on its generated packed-state kernel (awgn512.hip): 9-bit
    signed samples in int16 (rng.py:78), ragged sizes and offsets, two reset states."""
    m = oracle.Lutopt(path=oracle.data_path(512))
    for init in (1, int("0123456789abcdef" * 8, 16)):
        u = gpu.LUTOPT.shipped(512, init=init)
        got = gpu.CLTGRNG(u).generate(nsamples, first_step=first).cpu().numpy()
        st = m.states(init, first, min(nsamples, 20_000))
        exp = m.clt_tree_bulk(st).astype(np.int64)
        exp = ((exp + 256) % 512) - 256
        assert got.dtype == np.int16 and np.array_equal(got[:len(exp)].astype(np.int64), exp)
        if nsamples > 20_000:            # the tail, from a jumped state
            tail = m.states(u.state_at(first + nsamples - 5000), 0, 5000)
            texp = ((m.clt_tree_bulk(tail).astype(np.int64) + 256) % 512) - 256
            assert np.array_equal(got[-5000:].astype(np.int64), texp)


def test_n512_stream_with_prefetch_hints(gpu, oracle):
    """bbb_awgn_prefetch on the n512 handle (start states of the next fill seeded on the side stream): right hints, a wrong
    one and none; every fill equals an un-hinted handle's, and a stretch of a hinted fill the oracle."""
    m = oracle.Lutopt(path=oracle.data_path(512))
    u = gpu.LUTOPT.shipped(512, init=5)
    g = gpu.CLTGRNG(u)
    d = gpu.CLTGRNG(gpu.LUTOPT.shipped(512, init=5))
    n = 3_000_008
    outs = []
    for s in range(6):
        outs.append(g.generate(n, first_step=18 + s * n))
        if s in (0, 1, 4):
            g.prefetch(n, first_step=18 + (s + 1) * n)
        elif s == 2:
            g.prefetch(n, first_step=99)                    # wrong
    outs.append(g.generate(n + 8, first_step=7))
    torch.cuda.synchronize()
    for s, o in enumerate(outs[:6]):
        assert torch.equal(o, d.generate(n, first_step=18 + s * n)), s
    assert torch.equal(outs[6], d.generate(n + 8, first_step=7))
    st = m.states(u.state_at(18 + 2 * n), 0, 10_000)
    exp = ((m.clt_tree_bulk(st).astype(np.int64) + 256) % 512) - 256
    assert np.array_equal(outs[2][:10_000].cpu().numpy().astype(np.int64), exp)


def test_n512_rate_and_moments(gpu):
    """2^28 samples: variance 2^(9-2) = 128, mean 0 (rng.py:63-65); and the generated kernel is an order of magnitude
    faster than the table-driven one it replaces (~6 Gsample/s)."""
    import time
    u = gpu.LUTOPT.shipped(512)
    g = gpu.CLTGRNG(u)
    n = 1 << 28
    for i in range(3):                                   # (the first calls build the jump plan)
        x = g.generate(n, first_step=18 + i * n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = g.generate(n, first_step=18 + 3 * n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f = x[: 1 << 24].to(torch.float64)
    assert abs(float(f.mean())) < 0.05 and abs(float(f.var()) / 128.0 - 1.0) < 0.01
    print(f"n512: {n / dt / 1e9:.1f} Gsample/s")
    assert n / dt > 60e9
