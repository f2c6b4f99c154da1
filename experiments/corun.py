"""Round 3: the sample kernel beside the PLANES mover -- are the waves that share a SIMD with a mover wave slower than the
others (SIMD-local contention) or all alike (memory side)?  Two fills back to back, no prefetch: kernel 2 and mover 1 start
together.  Experiments build stamps: cycles per sample wave + its SIMD; the mover waves' SIMDs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basebandboard_amd import _lib
_lib.select_build("experiments")
import basebandboard_amd as bbb
L = _lib.lib()
L.bbb_exp_set_awgn_debug.argtypes = [C.c_void_p]; L.bbb_exp_set_awgn_debug.restype = None
N = 1_000_000_000
dbg = torch.zeros(8 * 1024, dtype=torch.int64, device="cuda")
L.bbb_exp_set_awgn_debug(C.c_void_p(dbg.data_ptr()))
buf = torch.empty(N, dtype=torch.int8, device="cuda")
u = bbb.LUTOPT.shipped(256); u.set_staged(True); g = bbb.CLTGRNG(u)
for s in range(3):
    g.generate(N, first_step=16 + s * N, out=buf); torch.cuda.synchronize()
for rep in range(3):
    dbg.zero_(); torch.cuda.synchronize()
    for s in range(2):
        g.generate(N, first_step=16 + (5 + s) * N, out=buf)
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy()
    d = raw[:4096].reshape(-1, 4)[:1018]
    cyc = (d[:, 1] - d[:, 0]).astype(np.float64); us = (d[:, 3] - d[:, 2]) / 100.0
    key = raw[5 * 1024:5 * 1024 + 1018]
    mv = raw[6 * 1024:8 * 1024].reshape(-1, 2)
    mv = mv[mv[:, 0] < 0]                       # bit 63 set = written
    mkeys = set(int(x) & 0x3fff for x in mv[:, 0])
    shared = np.array([int(k) in mkeys for k in key])
    st = (d[:, 2] - d[:, 2].min()) / 100.0
    en = (d[:, 3] - d[:, 2].min()) / 100.0
    mstart = (mv[:, 1] - d[:, 2].min()) / 100.0 if len(mv) else np.zeros(1)
    print("   sample-wave start offsets us: pct 0/25/50/75/90/99/100 =", [round(float(np.percentile(st, q)), 1) for q in (0, 25, 50, 75, 90, 99, 100)],
          " late (>50 us):", int((st > 50).sum()), " ends pct 0/50/100 =", [round(float(np.percentile(en, q)), 1) for q in (0, 50, 100)],
          " mover wave starts (relative) min/max:", round(float(mstart.min()), 1), round(float(mstart.max()), 1))
    print(f"rep {rep}: kernel span {(d[:,3].max()-d[:,2].min())/100:.1f} us; mover waves seen {len(mv)} on {len(mkeys)} SIMDs (last launch's); "
          f"sample waves sharing a SIMD with one: {shared.sum()}: cycles {cyc[shared].mean() if shared.any() else 0:.0f}, life {us[shared].mean() if shared.any() else 0:.1f} us; "
          f"others: {(~shared).sum()}: cycles {cyc[~shared].mean():.0f}, life {us[~shared].mean():.1f} us; distinct sample-wave SIMDs {len(set(int(k) for k in key))}")
