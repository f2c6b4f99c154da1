"""PRBS-31 loopback timing (1e10 bits, same buffer): fill, check right after the fill, check of a clean buffer."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
if len(sys.argv) > 1:
    _lib.select_build("experiments")
L = _lib.lib()
k = int(os.environ.get("K", "31"))
nbits = int(float(os.environ.get("NBITS", "1e10")))
nwords = (nbits + 63) // 64
A = torch.empty(nwords, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
fill = lambda: L.bbb_prbs_fill(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), 0, sp)
check = lambda: L.bbb_prbs_check_dev(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), C.c_void_p(cnt.data_ptr()), 0, sp)
ev = lambda: torch.cuda.Event(enable_timing=True)
for _ in range(3):
    fill(); check()
torch.cuda.synchronize()
f = c = c2 = 0.0
reps = 7
for _ in range(reps):
    e0, e1, e2, e3 = ev(), ev(), ev(), ev()
    e0.record(); fill(); e1.record(); check(); e2.record(); check(); e3.record()
    torch.cuda.synchronize()
    f += e0.elapsed_time(e1); c += e1.elapsed_time(e2); c2 += e2.elapsed_time(e3)
f, c, c2 = f / reps, c / reps, c2 / reps
gb = nbits / 8 / 1e9
print(f"k={k} fill {f:.4f} ms {gb/f:.2f} TB/s | check after fill {c:.4f} ms {gb/c:.2f} TB/s | check again {c2:.4f} ms {gb/c2:.2f} TB/s | "
      f"loopback {f+c:.4f} ms = {2*gb/(f+c):.2f} TB/s = {2*gb/(f+c)/8:.3f} of 8 TB/s | errors {int(cnt.item())}")
