"""Which guest costs the staged sample kernel what: prefetch on/off x staged on/off, kernel time from the library's events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for staged in (False, True):
    for prefetch in (False, True):
        u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
        g = bbb.CLTGRNG(u)
        first = lambda s: 16 + s * N
        for s in range(3):
            g.generate(N, first_step=first(s), out=buf)
            if prefetch: g.prefetch(N, first_step=first(s + 1))
        u.profile(True); u.profile_read(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(3, 13):
            g.generate(N, first_step=first(s), out=buf)
            if prefetch: g.prefetch(N, first_step=first(s + 1))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        seed_ms, kern_ms, calls = u.profile_read(reset=True)
        print(f"staged={staged} prefetch={prefetch}: {dt*1e3:.4f} ms/step, sample kernel {kern_ms/calls:.4f} ms, seeding (or wait for it) {seed_ms/calls:.4f} ms")
        del u, g
