"""Round 3: where a mover unit's time goes, beside the sample kernel and alone.  Experiments build: every mover wave sums
shader-clock ticks per phase over its units (unplane_kernel, BBB_PH).  Two fills back to back, no prefetch: mover 1 runs
beside kernel 2 (stamps in buffer A), mover 2 alone (buffer B)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basebandboard_amd import _lib
_lib.select_build("experiments")
import basebandboard_amd as bbb
L = _lib.lib()
L.bbb_exp_set_awgn_debug.argtypes = [C.c_void_p]; L.bbb_exp_set_awgn_debug.restype = None
N = 1_000_000_000
A = torch.zeros(16 * 1024, dtype=torch.int64, device="cuda")
B = torch.zeros(16 * 1024, dtype=torch.int64, device="cuda")
buf = torch.empty(N, dtype=torch.int8, device="cuda")
u = bbb.LUTOPT.shipped(256); u.set_staged(True); g = bbb.CLTGRNG(u)
L.bbb_exp_set_awgn_debug(C.c_void_p(B.data_ptr()))
for s in range(40):
    g.generate(N, first_step=16 + s * N, out=buf)
torch.cuda.synchronize()
names = ["issue DMA(u+1)", "wait DMA(u)", "barrier 1", "phase 1 (+lgkm wait)", "barriers 2+3", "phase 2 (+lgkm wait)"]
def report(tag, t, rep):
    d = t.cpu().numpy()[8 * 1024:].reshape(-1, 8)
    d = d[d[:, 6] > 0]
    units = d[:, 6].astype(np.float64)
    per = d[:, :6] / units[:, None]
    tot = per.sum(axis=1)
    print(f"rep {rep} mover {tag}: {len(d)} waves, {units.mean():.1f} units each, {tot.mean():.0f} shader cycles per unit: "
          + "; ".join(f"{n} {per[:, i].mean():.0f}" for i, n in enumerate(names)), flush=True)
C2 = torch.zeros(16 * 1024, dtype=torch.int64, device="cuda")
for rep in range(3):
    A.zero_(); B.zero_(); C2.zero_(); torch.cuda.synchronize()
    base = 100 + 40 * rep
    for s in range(24):            # a streaming loop with the hint: the mover of call 12 (A) runs beside the sample kernel of call 13
        L.bbb_exp_set_awgn_debug(C.c_void_p((A if s == 12 else C2 if s == 23 else B).data_ptr()))
        g.generate(N, first_step=16 + (base + s) * N, out=buf)
        g.prefetch(N, first_step=16 + (base + s + 1) * N)
    torch.cuda.synchronize()
    report("in the streaming loop (beside the next sample kernel)", A, rep)
    report("of the last call (nothing beside it but the seeding)", C2, rep)
