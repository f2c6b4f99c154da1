"""bbb_awgn_stream_*: the CLTGRNG sample stream as an object that is drained sequentially (the reference's generator emits
exactly one such stream, rng.py:70-108).  Byte-exact against the oracle for whole reads, ragged tails, a broken sequence
(other calls on the handle between two reads, a seek), int16 / n512 and the table-driven path."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BIG = 1 << 24


def test_stream_equals_oracle_whole_reads_and_ragged_tail(gpu, oracle):
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256, init=0xC0FFEE)
    g = gpu.CLTGRNG(u)
    n = BIG + 4096 + 16
    with g.stream(n, first_step=16) as st:
        parts = [st.next() for _ in range(3)]
        parts.append(st.read(1_000_003))              # a ragged read: the stream continues behind it
        parts.append(st.next())
        parts.append(st.read(17))
        parts.append(st.read(BIG + 1))                # a large read of another length
        assert st.tell() == 16 + 4 * n + 1_000_003 + 17 + BIG + 1
        torch.cuda.synchronize()
    total = sum(p.numel() for p in parts)
    ref = m.awgn(0xC0FFEE, 16, total, fast=True)
    off = 0
    for i, p in enumerate(parts):
        assert np.array_equal(p.cpu().numpy(), ref[off:off + p.numel()]), i
        off += p.numel()


def test_stream_survives_a_broken_sequence(gpu, oracle):
    """Other calls on the handle between two reads (a fill elsewhere, BER trials, a word fill, a seek): the stream's
    bytes are those of its positions whatever the announcement it had made has become."""
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    n = BIG + 48
    st = g.stream(n, first_step=16)
    a = st.next()
    other = g.generate(2 * BIG + 5, first_step=7_000_000_001)              # steals the announced start states' slot
    b = st.next()
    t = gpu.Trial(nbits=300_001, amp=100, noise_var=8)
    assert gpu.run_trials(u, [t])[0] == m.ber_trial(1, 31, 1, 100, 8, 16, 0, 300_001)
    c = st.next()
    w = u.generate_words(500, first_step=3)
    st.seek(5_000_000_000)
    d = st.next()
    e = st.next()
    st.close()
    torch.cuda.synchronize()
    ref = m.awgn(1, 16, 3 * n, fast=True)
    for i, x in enumerate((a, b, c)):
        assert np.array_equal(x.cpu().numpy(), ref[i * n:(i + 1) * n]), i
    far = m.awgn(u.state_at(5_000_000_000), 0, 400_000, fast=True)
    assert np.array_equal(d[:400_000].cpu().numpy(), far)
    assert np.array_equal(e[:300_000].cpu().numpy(), m.awgn(u.state_at(5_000_000_000 + n), 0, 300_000, fast=True))
    assert np.array_equal(other[:200_000].cpu().numpy(), m.awgn(u.state_at(7_000_000_001), 0, 200_000, fast=True))
    assert np.array_equal(w.cpu().numpy().view(np.uint32), m.words_u32(1, 3, 500))


def test_stream_owns_the_mode_and_restores_it(gpu):
    l = gpu._lib.lib()
    u = gpu.LUTOPT.shipped(256)
    g = gpu.CLTGRNG(u)
    plain = g.generate(BIG + 16, first_step=16)                          # one-kernel form (staged off)
    st = g.stream(BIG + 16, first_step=16)
    with pytest.raises(ValueError):
        g.stream(BIG, first_step=0)                                       # one stream per handle
    x = st.next()
    st.close()
    assert torch.equal(x, plain)
    st2 = g.stream(BIG + 16, first_step=16)                               # ... and again after close
    assert torch.equal(st2.next(), plain)
    st2.close()
    # a handle on which the caller chose look-ahead keeps it while a stream is open and afterwards
    u.set_staged(True, look_ahead=2)
    with g.stream(BIG + 16, first_step=16) as st3:
        y = [st3.next() for _ in range(3)]
    assert torch.equal(y[0], plain)
    again = g.generate(BIG + 16, first_step=16)
    assert torch.equal(again, plain)
    # null / bad arguments
    s = C.c_void_p()
    assert l.bbb_awgn_stream_open(u._h, 0, 0, 1, C.byref(s)) == gpu._lib.BBB_EINVAL
    assert l.bbb_awgn_stream_open(u._h, 100, 0, 3, C.byref(s)) == gpu._lib.BBB_EINVAL
    assert l.bbb_awgn_stream_next(None, None) == gpu._lib.BBB_EINVAL
    assert l.bbb_awgn_stream_close(None) == gpu._lib.BBB_OK


@pytest.mark.parametrize("k", [32, 512])
def test_stream_on_other_orders(gpu, oracle, k):
    """The table-driven path (n32) and the packed n512 kernel (int16): the stream is the same object."""
    m = oracle.Lutopt(path=oracle.data_path(k))
    u = gpu.LUTOPT.shipped(k)
    g = gpu.CLTGRNG(u)
    n = 200_000 if k == 32 else (1 << 20) + 8
    first = 2 * (k.bit_length() - 1)
    with g.stream(n, first_step=first) as st:
        parts = [st.next(), st.next(), st.read(1001)]
        torch.cuda.synchronize()
    pos = first
    for p in parts:
        got = p.cpu().numpy().astype(np.int64)
        if k == 32:
            exp = m.awgn(u.state_at(pos), 0, p.numel()).astype(np.int64)
        else:
            cnt = min(p.numel(), 20_000)
            exp = ((m.clt_tree_bulk(m.states(u.state_at(pos), 0, cnt)).astype(np.int64) + 256) % 512) - 256
            got = got[:cnt]
        assert np.array_equal(got, exp)
        pos += p.numel()


def test_waveform_stream_equals_the_plain_calls(gpu, oracle, golden_shaper):
    """bbb_tx_stream_*: TX.x read sequentially -- whole reads, a ragged one, other calls on the handle in between, a seek --
    against bbb_tx_fill_i16 on a plain handle for the same positions, and a stretch against the oracle."""
    n = BIG + 4096
    x = gpu.TX(31, 1, 0, 16, 1, 8)
    y = gpu.TX(31, 1, 0, 16, 1, 8)
    with x.stream(n, first_sample=100) as st:
        parts = [st.next() for _ in range(3)]
        parts.append(st.read(1_000_003))
        parts.append(st.next())
        noise = gpu.CLTGRNG(x.urng).generate(BIG + 32, first_step=9_000_000_000)      # breaks the announced sequence
        parts.append(st.next())
        assert st.tell() == 100 + 5 * n + 1_000_003
        st.seek(3_000_000_017)
        far = st.next()
        torch.cuda.synchronize()
    assert x.urng is not None and gpu._lib.lib().bbb_tx_stream_close(None) == gpu._lib.BBB_OK
    pos = 100
    for i, p in enumerate(parts):
        ref = y.generate(p.numel(), first_sample=pos, stream_on=False)
        assert torch.equal(p, ref), i
        pos += p.numel()
    assert torch.equal(far, y.generate(n, first_sample=3_000_000_017, stream_on=False))
    m = oracle.Lutopt(path=oracle.data_path(256))
    exp = oracle.tx(m, 1, golden_shaper["rcf_coeffs"][16], 31, 100_000, first_sample=100 + n, noise_var=8, warmup=16)
    assert np.array_equal(parts[1][:100_000].cpu().numpy(), exp)
    assert np.array_equal(noise[:50_000].cpu().numpy(), m.awgn(u_state(m, 9_000_000_000, x), 0, 50_000, fast=True))


def u_state(m, step, x):
    return x.urng.state_at(step)


def test_waveform_stream_refuses_a_second_stream_and_restores_the_level(gpu):
    x = gpu.TX(31, 1, 0, 16, 1, 8)
    st = x.stream(BIG)
    with pytest.raises(ValueError, match="already has an open stream"):
        gpu.CLTGRNG(x.urng).stream(BIG)
    st.close()
    with gpu.CLTGRNG(x.urng).stream(BIG) as s2:      # the handle is free again
        s2.next()
    torch.cuda.synchronize()


def test_waveform_stream_without_noise(gpu):
    """noise_en = 0: the stream is the shaper's output (nothing is staged, no hint is made): equal to the plain calls."""
    n = 3_000_003
    x = gpu.TX(31, 1, 0, 16, 0, 8)
    y = gpu.TX(31, 1, 0, 16, 0, 8)
    with x.stream(n, first_sample=5) as st:
        a, b = st.next(), st.read(1001)
        assert st.tell() == 5 + n + 1001
    torch.cuda.synchronize()
    assert torch.equal(a, y.generate(n, first_sample=5)) and torch.equal(b, y.generate(1001, first_sample=5 + n))
