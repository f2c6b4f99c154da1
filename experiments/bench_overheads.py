"""Round 3: what the bench line's timed region costs beyond the steady state: the profiling events inside it and the
pipeline's fill and drain over K = 20 steps.  Stream object (level 2), 1e9 samples per read."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
u = bbb.LUTOPT.shipped(256); g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device="cuda")
st = g.stream(N, first_step=16)
for s in range(80):
    st.next(out=buf)
torch.cuda.synchronize()
for prof in (False, True, False, True):
    for K in (20, 200):
        u.profile(prof)
        u.profile_read(reset=True); u.profile_read_mover(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(K):
            st.next(out=buf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        extra = ""
        if prof:
            sm, km, calls = u.profile_read(reset=True); mm, mv = u.profile_read_mover(reset=True)
            extra = f"; sample kernel {km / max(calls, 1):.4f} ms x {calls}, mover {mm / max(mv, 1):.4f} ms x {mv}"
        u.profile(False)
        print(f"profile={prof} K={K}: {dt / K * 1e3:.4f} ms/step = {N * K / dt / 1e9:.1f} Gsample/s{extra}", flush=True)
