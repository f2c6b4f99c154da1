"""The staged noise stream (1e9 per step, look-ahead 1 and 2) and the TX stream (1e9 per call, staged) with the seeding's
tables staged in LDS in 2 or 4 pieces or read in place from global memory (BBB_SEED_PARTS = 2 / 4 / 0, experiments build)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
bbb._lib.select_build("experiments")
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for m in (1, 2):
    u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=m if m > 1 else False)
    g = bbb.CLTGRNG(u)
    first = lambda s: 16 + s * N
    for s in range(4):
        g.generate(N, first_step=first(s), out=buf); g.prefetch(N, first_step=first(s + 1))
    u.profile(True); u.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(4, 24):
        g.generate(N, first_step=first(s), out=buf); g.prefetch(N, first_step=first(s + 1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    print(f"parts={os.environ.get('BBB_SEED_PARTS')} m={m}: {dt*1e3:.4f} ms/step = {N/dt/1e9:.1f} Gsample/s, sample kernel {kern_ms/calls/m:.4f} ms per 1e9", flush=True)
    del u, g
tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(True)
b16 = torch.empty(N, dtype=torch.int16, device="cuda")
for i in range(4):
    tx.generate(N, first_sample=i * N, out=b16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4, 14):
    tx.generate(N, first_sample=i * N, out=b16)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"parts={os.environ.get('BBB_SEED_PARTS')} TX staged 1e9: {dt*1e3:.4f} ms/call = {N/dt/1e9:.1f} Gsample/s", flush=True)
