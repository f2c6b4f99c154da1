"""Seeded random sizes / offsets / parameters for every entry point, GPU vs oracle (bit exact)."""
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KS = (7, 9, 11, 15, 20, 23, 31)


def test_fuzz_prbs(gpu, oracle):
    rng = np.random.default_rng(2024)
    for _ in range(60):
        k = int(rng.choice(KS))
        nbits = int(rng.integers(1, 3_000_000))
        first = int(rng.integers(0, 2**45))
        init = int(rng.integers(1, 1 << k))
        p = gpu.PRBS(k, init=init)
        got = p.generate(nbits, first_bit=first).cpu().numpy().view(np.uint64)
        exp, _ = oracle.prbs_packed(k, nbits, state=p.state_at(first), fast=True)
        assert np.array_equal(got, exp), (k, nbits, first, init)
        buf = torch.from_numpy(got.view(np.int64).copy()).cuda()
        flips = np.unique(rng.integers(0, nbits, size=int(rng.integers(0, 50))))
        m = np.zeros(buf.numel(), dtype=np.uint64)
        np.bitwise_xor.at(m, flips // 64, np.uint64(1) << (flips % 64).astype(np.uint64))
        buf ^= torch.from_numpy(m.view(np.int64)).cuda()
        assert gpu.PRBSErrorDetector(k).count_errors(buf, nbits, first_bit=first, init=init) == len(flips)


def test_fuzz_awgn(gpu, oracle):
    rng = np.random.default_rng(7)
    m = oracle.Lutopt(path=oracle.data_path(256))
    for _ in range(25):
        n = int(rng.integers(1, 2_000_000))
        first = int(rng.integers(0, 10**14))
        init = int.from_bytes(rng.bytes(32), "little") | 1
        u = gpu.LUTOPT.shipped(256, init=init)
        got = gpu.CLTGRNG(u).generate(n, first_step=first).cpu().numpy()
        exp = m.awgn(u.state_at(first), 0, n, fast=True)
        assert np.array_equal(got, exp), (n, first)


def test_fuzz_ber(gpu, oracle):
    rng = np.random.default_rng(11)
    m = oracle.Lutopt(path=oracle.data_path(256))
    u = gpu.LUTOPT.shipped(256)
    trials = []
    for _ in range(12):
        trials.append(gpu.Trial(nbits=int(rng.integers(1, 120_000)), amp=int(rng.integers(0, 2048)),
                                noise_var=int(rng.integers(0, 16)), prbs_k=int(rng.choice(KS)),
                                prbs_state=1, warmup=int(rng.integers(0, 64)), first_bit=int(rng.integers(0, 200_000))))
    res = gpu.run_trials(u, trials)
    for t, r in zip(trials, res):
        assert r == m.ber_trial(1, t.prbs_k, t.prbs_state, t.amp, t.noise_var, t.warmup, t.first_bit, t.nbits), vars(t)


def test_fuzz_tx(gpu, oracle):
    rng = np.random.default_rng(3)
    m = oracle.Lutopt(path=oracle.data_path(256))
    for _ in range(12):
        k = int(rng.choice(KS))
        sel = int(rng.integers(0, 32))
        nv = int(rng.integers(0, 16))
        n = int(rng.integers(1, 400_000))
        first = int(rng.integers(0, 300_000))
        be, ne, src = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        tx = gpu.TX(k, be, src, sel, ne, nv)
        got = tx.generate(n, first_sample=first).cpu().numpy()
        exp = oracle.tx(m, 1, tx.prbs_shaper.coefficients[sel], k, n, first_sample=first, source=src, bit_en=be,
                        noise_en=ne, noise_var=nv, warmup=16)
        assert np.array_equal(got, exp), (k, sel, nv, n, first, be, ne, src)


def test_capture_decoder(gpu):
    """software/memdump/decode.py: 8192 little-endian int16 samples -> (dat > 0)[::4]."""
    rng = np.random.default_rng(5)
    x = rng.integers(-300, 300, size=8192).astype(np.int16)
    raw = struct.pack("<8192h", *x.tolist())
    bits = gpu.RX.decode_capture(raw)
    assert np.array_equal(bits, (x.astype(float) > 0).astype(np.uint8)[::4])


def test_determinism_and_resume(gpu):
    """the same seed twice gives identical bytes; (state, offset) is the whole checkpoint --
    a stream generated in three pieces with different handles equals the stream generated at once."""
    n = 5_000_011
    a = gpu.CLTGRNG(gpu.LUTOPT.shipped(256, init=0xC0FFEE)).generate(n, first_step=16)
    b = gpu.CLTGRNG(gpu.LUTOPT.shipped(256, init=0xC0FFEE)).generate(n, first_step=16)
    assert torch.equal(a, b)
    cuts = (0, 1_234_567, 3_000_001, n)
    parts = [gpu.CLTGRNG(gpu.LUTOPT.shipped(256, init=0xC0FFEE)).generate(cuts[i + 1] - cuts[i], first_step=16 + cuts[i]) for i in range(3)]
    assert torch.equal(torch.cat(parts), a)
    p = gpu.PRBS(23, init=0x1ABCD)
    whole = p.generate(64 * 100_000)
    pieces = [gpu.PRBS(23, init=0x1ABCD).generate(64 * 25_000, first_bit=64 * 25_000 * i) for i in range(4)]
    assert torch.equal(torch.cat(pieces), whole)
    assert gpu.PRBS(23, init=p.state_at(64 * 50_000)).generate(64 * 50_000).equal(whole[50_000:])
