"""Round 2: the parts of the staged form in isolation (a device sync after every fill: nothing overlaps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
for staged in (False, True):
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
    g = bbb.CLTGRNG(u)
    buf = torch.empty(N, dtype=torch.int8, device="cuda")
    for s in range(2):
        g.generate(N, first_step=16 + s * N, out=buf); torch.cuda.synchronize()
    u.profile(True); u.profile_read(reset=True)
    tot = 0.0
    for s in range(2, 8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.generate(N, first_step=16 + s * N, out=buf)
        torch.cuda.synchronize(); tot += time.perf_counter() - t0
    seed_ms, kern_ms, calls = u.profile_read(reset=True)
    print(f"staged={staged}: isolated fill {tot/6*1e3:.4f} ms wall; seeding {seed_ms/calls:.4f} ms, sample kernel alone {kern_ms/calls:.4f} ms")
