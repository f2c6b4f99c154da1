"""Round 3: what the guests cost the sample kernel -- cycles or clock?  Experiments build: every wave of
awgn256_kernel<false, *> leaves s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end.
Printed per mode: waves, cycles per wave (mean / min / max), wave lifetime in us, the clock they ran at, and
the span first start -> last end of the launch."""
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
dbg = torch.zeros(5 * 1024, dtype=torch.int64, device="cuda")
L.bbb_exp_set_awgn_debug(C.c_void_p(dbg.data_ptr()))
buf = torch.empty(N, dtype=torch.int8, device="cuda")


def report(tag):
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy()
    d = raw[:4 * 1024].reshape(-1, 4)[:1018]
    rend = raw[4 * 1024:4 * 1024 + 1018].astype(np.float64)
    cyc = (d[:, 1] - d[:, 0]).astype(np.float64)
    us = (d[:, 3] - d[:, 2]).astype(np.float64) / 100.0
    span = (d[:, 3].max() - d[:, 2].min()) / 100.0
    late = (d[:, 2] - d[:, 2].min()) / 100.0
    print(f"{tag:34s} cycles/wave {cyc.mean():10.0f} [{cyc.min():.0f} .. {cyc.max():.0f}]  life {us.mean():7.1f} us [{us.min():.1f} .. {us.max():.1f}]  "
          f"clock {(cyc / us).mean() / 1e3:5.3f} GHz  span {span:7.1f} us  start spread {late.max():6.1f} us (p50 {np.median(late):.1f})  round ends {rend.mean():8.0f} cycles = {100 * rend.mean() / cyc.mean():4.1f} %", flush=True)


for staged in (False, True):
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
    g = bbb.CLTGRNG(u)
    for s in range(3):
        g.generate(N, first_step=16 + s * N, out=buf); torch.cuda.synchronize()
    for rep in range(3):
        g.generate(N, first_step=16 + (3 + rep) * N, out=buf)
        report(f"staged={staged} alone #{rep}")
    # streaming: every fill announces the next (bench.py's loop); the stamps are those of the last kernel
    for rep in range(3):
        for s in range(10):
            g.generate(N, first_step=16 + (10 + s) * N, out=buf)
            g.prefetch(N, first_step=16 + (11 + s) * N)
        report(f"staged={staged} streaming, last of 10 #{rep}")
    if staged:
        for rep in range(2):
            for s in range(10):
                g.generate(N, first_step=16 + (10 + s) * N, out=buf)
            report(f"staged=True no prefetch, last of 10 #{rep}")
