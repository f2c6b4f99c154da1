"""Round 3: what slows the mover beside the sample kernel?  The experiments build's BBB_EXP_PLANES_FLAGS (1: the sample kernel
computes but stores nothing; 2: plain instead of non-temporal stores; 4: wave priority 0 instead of 3) against the mover's own
time (events on its stream) and the step time.  One process per setting (the knob is read at the first launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
_l.select_build('experiments')
import basebandboard_amd as bbb
N = 1_000_000_000
LA = int(os.environ.get('RAMP_LA', '2'))
u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=LA if LA >= 2 else False); g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device="cuda")
def loop(k, s0):
    for s in range(s0, s0 + k):
        g.generate(N, first_step=16 + s * N, out=buf)
        g.prefetch(N, first_step=16 + (s + 1) * N)
loop(80, 0)
torch.cuda.synchronize()
u.profile(True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); loop(80, 80); b.record(); torch.cuda.synchronize()
seed_ms, kern_ms, calls = u.profile_read()
mv_ms, movers = u.profile_read_mover()
print(f"flags={os.environ.get('BBB_EXP_PLANES_FLAGS', '0')} level={LA}: {a.elapsed_time(b) / 80:.4f} ms/step; sample kernel {kern_ms / max(calls, 1):.4f} ms x {calls}, "
      f"seeding {seed_ms / max(calls, 1):.4f} ms, mover {mv_ms / max(movers, 1):.4f} ms x {movers}", flush=True)
