"""Design experiment (round 2): can a PRBS fill (HBM writes) and a PRBS check (HBM reads) run side by side?
Two separate 1.25 GB buffers, two streams; sequential vs concurrent wall time."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
if len(sys.argv) > 1:
    _lib.select_build("experiments")
L = _lib.lib()
nbits = 10_000_000_000
nwords = (nbits + 63) // 64
A = torch.empty(nwords, dtype=torch.int64, device="cuda")
B = torch.empty(nwords, dtype=torch.int64, device="cuda")
junk = torch.empty(1 << 28, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def fill(buf, first, n, st):
    assert L.bbb_prbs_fill(31, 1, first, n, C.c_void_p(buf.data_ptr() + first // 8), 0, C.c_void_p(st.cuda_stream)) == 0


def check(buf, first, n, st):
    assert L.bbb_prbs_check_dev(31, 1, first, n, C.c_void_p(buf.data_ptr() + first // 8), C.c_void_p(cnt.data_ptr()), 0, C.c_void_p(st.cuda_stream)) == 0


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        junk.fill_(1)                      # 2 GiB of other writes: B is not in any cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


fill(B, 0, nbits, sA); fill(A, 0, nbits, sA); torch.cuda.synchronize()
print("fill alone          %.3f ms" % timed(lambda: fill(A, 0, nbits, sA)))
print("check alone (cold)  %.3f ms" % timed(lambda: check(B, 0, nbits, sB)))
print("fill ; check (seq)  %.3f ms" % timed(lambda: (fill(A, 0, nbits, sA), check(B, 0, nbits, sA))))
print("fill || check       %.3f ms" % timed(lambda: (fill(A, 0, nbits, sA), check(B, 0, nbits, sB))))
print("check || fill       %.3f ms" % timed(lambda: (check(B, 0, nbits, sB), fill(A, 0, nbits, sA))))
print("errors", int(cnt.item()))
# pipelined true loopback on ONE buffer: check(i) beside fill(i+1)
for npieces in (2, 3, 4, 6, 8, 12):
    pb = (nbits // npieces + 8191) // 8192 * 8192
    pieces = [(f, min(pb, nbits - f)) for f in range(0, nbits, pb)]
    def loop():
        for f, n in pieces:
            fill(A, f, n, sA)
            e = torch.cuda.Event(); e.record(sA); sB.wait_event(e)
            check(A, f, n, sB)
    cnt.zero_()
    t = timed(loop)
    print(f"pipelined loopback, {len(pieces):2d} pieces of {pb/8/2**20:6.0f} MiB: {t:.3f} ms = {2*nbits/8/t/1e9:.2f} TB/s effective, errors {int(cnt.item())}", flush=True)
