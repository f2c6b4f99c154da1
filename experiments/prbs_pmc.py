"""Round 5: the PRBS-31 loopback kernels under rocprofv3 (kernel trace, or one --pmc pass): 1e10 bits, the loopback order
(hinted fill, check behind it), the plain fill, a check of the clean buffer, and torch's fill_ over the same 1.25 GB for
comparison.  Counter collection serialises kernels; each kernel here runs alone anyway."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
if os.environ.get("BBB_BUILD") == "experiments":
    _lib.select_build("experiments")
L = _lib.lib()
nbits = int(float(os.environ.get("NBITS", "1e10")))
reps = int(os.environ.get("REPS", "4"))
nwords = (nbits + 63) // 64
A = torch.empty(nwords, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fill = lambda: _lib.check(L.bbb_prbs_fill(31, 1, 0, nbits, C.c_void_p(A.data_ptr()), 0, sp), "fill")
fill_rb = lambda: _lib.check(L.bbb_prbs_fill_hint(31, 1, 0, nbits, C.c_void_p(A.data_ptr()), 1, 0, sp), "fill_hint")
check = lambda: _lib.check(L.bbb_prbs_check_dev(31, 1, 0, nbits, C.c_void_p(A.data_ptr()), C.c_void_p(cnt.data_ptr()), 0, sp), "check")
for _ in range(2):
    fill_rb(); check()
torch.cuda.synchronize()
for _ in range(reps):
    fill_rb(); check()          # the loopback as bench.py times it
torch.cuda.synchronize()
for _ in range(reps):
    fill(); check(); check()    # plain fill, check behind it, check of the clean buffer
torch.cuda.synchronize()
B = A.view(torch.int8)
for _ in range(reps):
    B.fill_(1)
torch.cuda.synchronize()
print("errors", int(cnt.item()))
