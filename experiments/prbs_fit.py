"""PRBS-31 fill / clean check time vs size (events, 10 reps each): the intercept is the per-call fixed cost (launch +
bootstrap + tail), the slope the streaming rate."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
L = _lib.lib()
k = 31
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for nbits in (250_000_000, 500_000_000, 1_000_000_000, 2_000_000_000, 4_000_000_000, 10_000_000_000):
    A = torch.empty((nbits + 63) // 64, dtype=torch.int64, device="cuda")
    fill = lambda: L.bbb_prbs_fill(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), 0, sp)
    check = lambda: L.bbb_prbs_check_dev(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), C.c_void_p(cnt.data_ptr()), 0, sp)
    for _ in range(3): fill(); check()
    torch.cuda.synchronize()
    e0, e1 = ev(), ev(); e0.record()
    for _ in range(10): fill()
    e1.record(); torch.cuda.synchronize(); tf = e0.elapsed_time(e1) / 10
    check(); check(); torch.cuda.synchronize()
    e0, e1 = ev(), ev(); e0.record()
    for _ in range(10): check()
    e1.record(); torch.cuda.synchronize(); tc = e0.elapsed_time(e1) / 10
    print(f"nbits {nbits:>12}: fill {tf*1e3:8.1f} us, clean check {tc*1e3:8.1f} us", flush=True)
    rows.append((nbits / 8, tf, tc)); del A
(b0, f0, c0), (b1, f1, c1) = rows[-3], rows[-1]
for name, y0, y1 in (("fill", f0, f1), ("check", c0, c1)):
    slope = (y1 - y0) / (b1 - b0)
    print(f"{name}: {1/slope/1e9:.2f} TB/s incremental, intercept {(y0 - slope*b0)*1e3:.1f} us")
