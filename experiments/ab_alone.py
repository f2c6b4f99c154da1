"""Round 3: the sample kernel ALONE (a synchronise after every fill: no mover, no seeding beside it) in two builds (AB_LIB)."""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
if os.environ.get("AB_LIB"):
    _l.LIB_PATH = pathlib.Path(os.environ["AB_LIB"]).resolve()
import basebandboard_amd as bbb
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
LEVEL = int(os.environ.get('AB_LEVEL', '2'))      # 2: the form the noise stream's default level runs (one kernel per two fills)
u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=LEVEL if LEVEL >= 2 else False); g = bbb.CLTGRNG(u)
for s in range(60):                       # clock ramp
    g.generate(N, first_step=16 + s * N, out=buf)
torch.cuda.synchronize()
u.profile(True)
for s in range(60, 100):
    g.generate(N, first_step=16 + s * N, out=buf)
    torch.cuda.synchronize()
seed_ms, kern_ms, calls = u.profile_read(); mv_ms, movers = u.profile_read_mover()
print(f"{os.environ.get('AB_LIB', 'product')}: alone (level {LEVEL}): sample kernel {kern_ms / calls / max(LEVEL, 1):.4f} ms per 1e9 x {calls}, seeding {seed_ms / calls:.4f}, mover {mv_ms / movers:.4f} ms x {movers}", flush=True)
