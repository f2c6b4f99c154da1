"""Round 3: per launch, cycles per wave and the clock it ran at (experiments build stamps), for a run of fills that
starts right behind a device synchronisation (hot GPU, tiny gap) and one that starts after an idle second."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basebandboard_amd import _lib
_lib.select_build("experiments")
import basebandboard_amd as bbb
L = _lib.lib()
L.bbb_exp_set_awgn_debug.argtypes = [C.c_void_p]; L.bbb_exp_set_awgn_debug.restype = None
N = 1_000_000_000
K = 60
dbg = torch.zeros(K, 8 * 1024, dtype=torch.int64, device="cuda")
buf = torch.empty(N, dtype=torch.int8, device="cuda")
u = bbb.LUTOPT.shipped(256); u.set_staged(True); g = bbb.CLTGRNG(u)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]


def run(tag, idle):
    for s in range(100):                               # hot
        g.generate(N, first_step=16 + s * N, out=buf); g.prefetch(N, first_step=16 + (s + 1) * N)
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle)
    ev[0].record()
    for s in range(K):
        L.bbb_exp_set_awgn_debug(C.c_void_p(dbg[s].data_ptr()))
        g.generate(N, first_step=16 + s * N, out=buf); g.prefetch(N, first_step=16 + (s + 1) * N)
        ev[s + 1].record()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()[:, :4 * 1024].reshape(K, -1, 4)[:, :1018]
    print(tag)
    prev_end = None
    for s in range(K):
        cyc = (d[s, :, 1] - d[s, :, 0]).astype(np.float64); us = (d[s, :, 3] - d[s, :, 2]) / 100.0
        start, end = d[s, :, 2].min() / 100.0, d[s, :, 3].max() / 100.0
        gap = start - prev_end if prev_end is not None else 0.0
        prev_end = end
        if s < 12 or s % 8 == 0:
            print(f"  step {s:2d}: event dt {ev[s].elapsed_time(ev[s + 1]):6.3f} ms  kernel span {end - start:7.1f} us  gap before {gap:6.1f} us  "
                  f"cycles {cyc.mean():9.0f}  clock {(cyc / us).mean() / 1e3:5.3f} GHz")


run("hot start (sync, then straight on)", 0)

