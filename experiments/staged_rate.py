"""Round 2: one-kernel vs staged (two-kernel) form of the sample stream: steady-state rate of back-to-back fills,
AWGN int8 (1e9 samples per fill) and TX int16 (2^29 samples per call); outputs compared byte for byte."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")

N = 1_000_000_000
def awgn(staged, steps=10, prefetch=True):
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
    g = bbb.CLTGRNG(u)
    buf = torch.empty(N, dtype=torch.int8, device="cuda")
    first = lambda s: 16 + s * N
    for s in range(3):
        g.generate(N, first_step=first(s), out=buf)
        if prefetch: g.prefetch(N, first_step=first(s + 1))
    u.profile(True); u.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(3, 3 + steps):
        g.generate(N, first_step=first(s), out=buf)
        if prefetch: g.prefetch(N, first_step=first(s + 1))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    return dt / steps * 1e3, kern_ms / calls, buf

ms0, k0, b0 = awgn(False); ref = b0.clone()
ms1, k1, b1 = awgn(True)
print(f"AWGN direct: {ms0:.4f} ms/step ({N/ms0/1e6:.1f} Gsample/s), sample kernel {k0:.4f} ms")
print(f"AWGN staged: {ms1:.4f} ms/step ({N/ms1/1e6:.1f} Gsample/s), sample kernel {k1:.4f} ms, identical output: {torch.equal(ref, b1)}")
del b0, b1, ref

ntx = 1 << 29
def txrate(staged):
    tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(staged)
    buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
    tx.generate(ntx, out=buf); tx.generate(ntx, first_sample=ntx, out=buf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2, 8):
        tx.generate(ntx, first_sample=i * ntx, out=buf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    return dt * 1e3, buf
t0_, x0 = txrate(False); r = x0.clone()
t1_, x1 = txrate(True)
print(f"TX direct: {t0_:.4f} ms/call ({ntx/t0_/1e6:.1f} Gsample/s)")
print(f"TX staged: {t1_:.4f} ms/call ({ntx/t1_/1e6:.1f} Gsample/s), identical output: {torch.equal(r, x1)}")
