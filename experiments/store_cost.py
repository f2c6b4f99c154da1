"""Round 5 (verdict item 5): what does the 2 B per sample of staging traffic cost the sample kernel?  awgn256_planes_kernel ALONE (every
call synchronised: its mover runs after it, nothing beside it), the bench's launch size (2e9 samples: L = 960), with its stores
and without (experiments build, BBB_EXP_PLANES_FLAGS=1: the count planes go to an empty asm instead of memory -- same instructions
otherwise), alternating in one process.  Per launch from the per-wave stamps: cycles per wave, the wave's lifetime, the clock it ran
at, the launch's span."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basebandboard_amd import _lib
_lib.select_build("experiments")
import basebandboard_amd as bbb
L = _lib.lib()
L.bbb_exp_set_awgn_debug.argtypes = [C.c_void_p]
L.bbb_exp_set_awgn_debug.restype = None
N = 1_000_000_000
dbg = torch.zeros(16 * 1024, dtype=torch.int64, device="cuda")
L.bbb_exp_set_awgn_debug(C.c_void_p(dbg.data_ptr()))
buf = torch.empty(N, dtype=torch.int8, device="cuda")
u = bbb.LUTOPT.shipped(256)
u.set_staged(True, look_ahead=2)
g = bbb.CLTGRNG(u)
pos = [16]


def launch():                      # one sample kernel (2e9 samples) + its two movers, then idle
    for _ in range(2):
        g.generate(N, first_step=pos[0], out=buf)
        pos[0] += N
    torch.cuda.synchronize()


def stamps():
    raw = dbg.cpu().numpy()
    d = raw[:4 * 1024].reshape(-1, 4)[:1018]
    cyc = (d[:, 1] - d[:, 0]).astype(np.float64)
    us = (d[:, 3] - d[:, 2]).astype(np.float64) / 100.0
    span = (d[:, 3].max() - d[:, 2].min()) / 100.0
    return cyc.mean(), us.mean(), (cyc / us).mean() / 1e3, span


os.environ["BBB_EXP_PLANES_FLAGS"] = "0"
for _ in range(30):                # clocks
    launch()
rows = {0: [], 1: []}
for rep in range(12):
    for flag in (0, 1):
        os.environ["BBB_EXP_PLANES_FLAGS"] = str(flag)
        launch(); launch()         # (the second of two: the first follows the other variant)
        rows[flag].append(stamps())
for flag, name in ((0, "with its stores   "), (1, "stores suppressed ")):
    a = np.array(rows[flag])
    print(f"{name}: cycles per wave {a[:, 0].mean():10.0f}  wave lifetime {a[:, 1].mean():8.1f} us  clock {a[:, 2].mean():5.3f} GHz  "
          f"launch span {a[:, 3].mean():8.1f} us (min {a[:, 3].min():.1f}, max {a[:, 3].max():.1f}) over {len(a)} launches", flush=True)
a0, a1 = np.array(rows[0]), np.array(rows[1])
print(f"stores cost: {100 * (a0[:, 0].mean() / a1[:, 0].mean() - 1):+.2f} % cycles, {100 * (a0[:, 2].mean() / a1[:, 2].mean() - 1):+.2f} % clock, "
      f"{100 * (a0[:, 3].mean() / a1[:, 3].mean() - 1):+.2f} % time per launch")
