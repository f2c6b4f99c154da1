"""Round 5: where the SHAPING mover's time goes beside the transmitter's noise kernel (verdict item 3: the call is mover-bound).
Experiments build: every mover wave sums shader-clock ticks per phase over its units (unplane_kernel<true>, BBB_PH; the overlay's
UnplaneGeom::dbg).  A transmitter stream of 1e9 samples per call (level 2: call 2k launches noise kernel k and mover A(k), call
2k + 1 mover B(k); both run beside kernel k + 1), the stamps of four calls kept: an A and a B in the middle of the stream, and the
last two (nothing beside them).  BBB_EXP_MOVER_FLAGS (1: mover at wave priority 3, 2: priority 1) and BBB_EXP_PLANES_FLAGS
(4 / 8 / 16: the noise kernel at priority 0 / 1 / 2 instead of 3) are read by the library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basebandboard_amd import _lib
_lib.select_build("experiments")
import basebandboard_amd as bbb
L = _lib.lib()
L.bbb_exp_set_awgn_debug.argtypes = [C.c_void_p]; L.bbb_exp_set_awgn_debug.restype = None
N = 1_000_000_000
bufs = {k: torch.zeros(16 * 1024, dtype=torch.int64, device="cuda") for k in ("rest", "A", "B", "lastA", "lastB")}
tx = bbb.TX(31, 1, 0, 16, 1, 8, device=0)
txbuf = torch.empty(N, dtype=torch.int16, device="cuda")
stx = tx.stream(N)
L.bbb_exp_set_awgn_debug(C.c_void_p(bufs["rest"].data_ptr()))
for i in range(24):
    stx.next(out=txbuf)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
a, b = ev(), ev()
a.record()
for i in range(40):
    stx.next(out=txbuf)
b.record()
torch.cuda.synchronize()
txms = a.elapsed_time(b) / 40
print(f"MOVER_FLAGS={os.environ.get('BBB_EXP_MOVER_FLAGS', '0')} PLANES_FLAGS={os.environ.get('BBB_EXP_PLANES_FLAGS', '0')}: TX stream {N / txms / 1e6:.1f} Gsample/s ({txms:.4f} ms per call)", flush=True)
names = ["issue DMA(u+1)", "wait DMA(u)", "barrier 1", "phase 1 (+lgkm wait)", "barriers 2+3", "phase 2 (+lgkm wait)"]
def report(tag, t):
    d = t.cpu().numpy()[8 * 1024:].reshape(-1, 8)
    d = d[d[:, 6] > 0]
    if not len(d):
        print(f"mover {tag}: no stamps"); return
    units = d[:, 6].astype(np.float64)
    per = d[:, :6] / units[:, None]
    tot = per.sum(axis=1)
    print(f"  mover {tag}: {len(d)} waves, {units.mean():.1f} units each, {tot.mean():.0f} shader cycles per unit: "
          + "; ".join(f"{n} {per[:, i].mean():.0f}" for i, n in enumerate(names)), flush=True)
for rep in range(2):
    for t in bufs.values(): t.zero_()
    torch.cuda.synchronize()
    for s in range(24):
        key = {12: "A", 13: "B", 22: "lastA", 23: "lastB"}.get(s, "rest")
        L.bbb_exp_set_awgn_debug(C.c_void_p(bufs[key].data_ptr()))
        stx.next(out=txbuf)
    torch.cuda.synchronize()
    report("A in the stream (beside the next noise kernel)", bufs["A"])
    report("B in the stream (beside the next noise kernel)", bufs["B"])
    report("A of the last kernel", bufs["lastA"])
    report("B of the last kernel (nothing beside it)", bufs["lastB"])
stx.close()
