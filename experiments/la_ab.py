"""Look-ahead A/B in one process: m = 1, 2, 1, 2, 4 on the staged noise stream, 40 steps of 1e9 each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for m in [int(x) for x in os.environ.get("MS", "1,2,1,2,4,1,2").split(",")]:
    u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=m if m > 1 else False)
    g = bbb.CLTGRNG(u)
    first = lambda s: 16 + s * N
    for s in range(4):
        g.generate(N, first_step=first(s), out=buf); g.prefetch(N, first_step=first(s + 1))
    u.profile(True); u.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(4, 44):
        g.generate(N, first_step=first(s), out=buf); g.prefetch(N, first_step=first(s + 1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    print(f"m={m}: {dt*1e3:.4f} ms/step = {N/dt/1e9:.1f} Gsample/s, sample kernel {kern_ms/calls/m:.4f} ms per 1e9 ({calls} launches)", flush=True)
    del u, g
