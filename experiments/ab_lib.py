"""Round 3: same-box A/B of two builds of the library (AB_LIB = path of the .so to load): steady-state step time of the
noise stream at one and two reads per sample kernel, and the sample kernel's / mover's own times (profiling events)."""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
if os.environ.get("AB_LIB"):
    _l.LIB_PATH = pathlib.Path(os.environ["AB_LIB"]).resolve()
import basebandboard_amd as bbb
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for LA in ((2,) if os.environ.get("AB_ONLY2") else (2, 0)):
    u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=LA if LA >= 2 else False); g = bbb.CLTGRNG(u)
    def loop(k, s0):
        for s in range(s0, s0 + k):
            g.generate(N, first_step=16 + s * N, out=buf)
            g.prefetch(N, first_step=16 + (s + 1) * N)
    loop(100, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); loop(200, 100); b.record(); torch.cuda.synchronize()
    plain = a.elapsed_time(b) / 200
    u.profile(True)
    a.record(); loop(80, 300); b.record(); torch.cuda.synchronize()
    seed_ms, kern_ms, calls = u.profile_read(); mv_ms, movers = u.profile_read_mover()
    u.profile(False)
    print(f"{os.environ.get('AB_LIB', 'product')} level {max(LA, 1)}: {plain:.4f} ms/step; with events {a.elapsed_time(b) / 80:.4f}: sample kernel {kern_ms / max(calls, 1):.4f} ms x {calls}, "
          f"mover {mv_ms / max(movers, 1):.4f} ms x {movers}", flush=True)
    del g, u
