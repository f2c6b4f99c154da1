"""Round 5: fills that are NOT announced (plain bbb_awgn_fill_i8 calls, each at another stream position, no prefetch): every fill derives
its start states in line -- the one-kernel form and the staged form at level 1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if len(sys.argv) > 1 and sys.argv[1] == 'exp':
    bbb._lib.select_build('experiments')
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for staged in (False, True):
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
    g = bbb.CLTGRNG(u)
    for s in range(40):
        g.generate(N, first_step=16 + s * N, out=buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(40, 80):
        g.generate(N, first_step=16 + s * N, out=buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"unannounced fills, staged={staged}: {dt * 1e3:.4f} ms per 1e9 samples = {N / dt / 1e9:.1f} Gsample/s", flush=True)
