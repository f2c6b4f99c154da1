"""The two-kernel ("staged") form of the k = 256 sample stream (bbb_lutopt_set_staged): the sample kernel leaves its
pieces in a staging buffer as full lines, a second kernel moves them to their place on an internal stream, the next
fill's arithmetic overlaps that.  Same bytes as the one-kernel form and as the oracle, whatever is interleaved."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BIG = 1 << 24          # fills of at least this many samples take the staged form


def test_staged_fill_equals_oracle_and_one_kernel_form(gpu, oracle):
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256)
    u.set_staged(True)
    g = gpu.CLTGRNG(u)
    d = gpu.CLTGRNG(gpu.LUTOPT.shipped(256))
    for n, first in ((BIG, 0), (BIG + 5, 16), (30_000_017, 12_345), (BIG - 16, 3)):       # the last one is below the threshold
        got = g.generate(n, first_step=first)
        snap = got.clone()                              # queued on the caller's stream right behind the fill
        exp = m.awgn(u.state_at(first), 0, 300_000, fast=True)
        assert np.array_equal(snap[:300_000].cpu().numpy(), exp)
        tail = m.awgn(u.state_at(first + n - 4096), 0, 4096, fast=True)
        assert np.array_equal(snap[-4096:].cpu().numpy(), tail)
        assert torch.equal(snap, d.generate(n, first_step=first))
        assert torch.equal(got, snap)


def test_back_to_back_staged_fills_with_prefetch(gpu, oracle):
    """The streaming pattern of bench.py: every fill announces the next one; the movers and the seedings run beside
    the following fill's arithmetic.  Four different buffers, all checked at the end."""
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256, init=0x1234567)
    u.set_staged(True)
    g = gpu.CLTGRNG(u)
    n = BIG + 4096
    bufs = []
    for s in range(4):
        bufs.append(g.generate(n, first_step=16 + s * n))
        g.prefetch(n, first_step=16 + (s + 1) * n)
    torch.cuda.synchronize()
    ref = m.awgn(0x1234567, 16, 4 * n, fast=True)
    for s, b in enumerate(bufs):
        assert np.array_equal(b.cpu().numpy(), ref[s * n:(s + 1) * n])
    # the same buffer again and again (what the bench does): the mover of fill s must not overtake a reader of fill s-1
    buf = torch.empty(n, dtype=torch.int8, device="cuda")
    sums = []
    for s in range(4):
        g.generate(n, first_step=16 + s * n, out=buf)
        sums.append(buf.to(torch.int64).sum())          # a reader on the caller's stream between two fills
    assert [int(x) for x in sums] == [int(ref[s * n:(s + 1) * n].astype(np.int64).sum()) for s in range(4)]


def test_staged_fills_interleaved_with_other_calls(gpu, oracle):
    """Staged fills run on an internal stream, everything else on the caller's: the hand-over between them."""
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256)
    u.set_staged(True)
    g = gpu.CLTGRNG(u)
    n = BIG + 48
    a = g.generate(n, first_step=16)
    t = gpu.Trial(nbits=400_003, amp=100, noise_var=8)
    assert gpu.run_trials(u, [t])[0] == m.ber_trial(1, 31, 1, 100, 8, 16, 0, 400_003)
    b = g.generate(n, first_step=16 + n)
    w = u.generate_words(1000, first_step=7)
    small = g.generate(5000, first_step=99)               # one-kernel form on the caller's stream
    c = g.generate(n, first_step=16 + 2 * n)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                          # the caller re-binds the handle to another stream
        dd = g.generate(n, first_step=16 + 3 * n)
        dsum = dd.to(torch.int64).sum()
    torch.cuda.synchronize()
    ref = m.awgn(1, 16, 4 * n, fast=True)
    for s, x in enumerate((a, b, c, dd)):
        assert np.array_equal(x.cpu().numpy(), ref[s * n:(s + 1) * n])
    assert int(dsum) == int(ref[3 * n:].astype(np.int64).sum())
    assert np.array_equal(small.cpu().numpy(), m.awgn(1, 99, 5000, fast=True))
    assert np.array_equal(w.cpu().numpy().view(np.uint32), m.words_u32(1, 7, 1000))


def test_staged_tx_equals_one_kernel_form(gpu, oracle, golden_shaper):
    n = BIG + 1000
    x = gpu.TX(31, 1, 0, 16, 1, 8)
    x.urng.set_staged(True)
    y = gpu.TX(31, 1, 0, 16, 1, 8)
    outs = [x.generate(n, first_sample=i * n) for i in range(3)]
    refs = [y.generate(n, first_sample=i * n) for i in range(3)]
    torch.cuda.synchronize()
    for a, b in zip(outs, refs):
        assert torch.equal(a, b)
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp = oracle.tx(m, 1, golden_shaper["rcf_coeffs"][16], 31, 200_000, first_sample=n, noise_var=8, warmup=16)
    assert np.array_equal(outs[1][:200_000].cpu().numpy(), exp)
