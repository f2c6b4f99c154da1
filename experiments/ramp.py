"""Round 3: step time against step index from a cold (idle) GPU -- how long the clock governor takes to reach the
steady state the sample kernel then runs at.  Events on the caller's stream behind every fill (it waits for the mover)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
if os.environ.get('BBB_EXP'): _l.select_build('experiments')
import basebandboard_amd as bbb
N = 1_000_000_000
LA = int(os.environ.get('RAMP_LA', '0'))
u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=LA if LA >= 2 else False); g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for s in range(3):
    g.generate(N, first_step=16 + s * N, out=buf)
torch.cuda.synchronize()
for idle in ((2.0, 0.2, 0.0) if not os.environ.get('RAMP_QUICK') else (0.0,)):
    time.sleep(idle)
    K = int(os.environ.get("RAMP_K", "600"))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for s in range(K):
        g.generate(N, first_step=16 + s * N, out=buf)
        g.prefetch(N, first_step=16 + (s + 1) * N)
        ev[s + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    cum = [ev[0].elapsed_time(ev[i]) for i in range(K + 1)]
    print(f"after {idle} s idle: ms per step, steps 0-4: {[round(x, 3) for x in t[:5]]}")
    for lo, hi in ((0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, 600)):
        if hi > K: break
        print(f"   steps {lo:3d}-{hi:3d}: {sum(t[lo:hi]) / (hi - lo):.4f} ms/step   (t = {cum[lo]:.1f} .. {cum[hi]:.1f} ms)")
