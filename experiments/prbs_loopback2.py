"""Round 4: PRBS-31 loopback (1e10 bits) as the loopback caller runs it: fill with the read-back hint, then the check; and the
plain pair.  AB: BBB_PRBS_SEEDS=0 in the experiments build turns the shared region seeds off."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
if os.environ.get("EXP"):
    _lib.select_build("experiments")
L = _lib.lib()
k = 31
nbits = 10_000_000_000
nwords = (nbits + 63) // 64
A = torch.empty(nwords, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fill = lambda: L.bbb_prbs_fill(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), 0, sp)
fillh = lambda: L.bbb_prbs_fill_hint(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), 1, 0, sp)
check = lambda: L.bbb_prbs_check_dev(k, 1, 0, nbits, C.c_void_p(A.data_ptr()), C.c_void_p(cnt.data_ptr()), 0, sp)
ev = lambda: torch.cuda.Event(enable_timing=True)
for _ in range(3):
    fill(); check()
torch.cuda.synchronize()
gb = nbits / 8 / 1e9
for name, ff in (("hinted", fillh), ("plain", fill)):
    f = c = 0.0
    reps = 7
    for _ in range(reps):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record(); ff(); e1.record(); check(); e2.record()
        torch.cuda.synchronize()
        f += e0.elapsed_time(e1); c += e1.elapsed_time(e2)
    f, c = f / reps, c / reps
    print(f"{name}: fill {f:.4f} ms {gb/f:.2f} TB/s | check {c:.4f} ms {gb/c:.2f} TB/s | loopback {f+c:.4f} ms = {2*gb/(f+c)/8:.3f} of 8 TB/s | errors {int(cnt.item())}", flush=True)
