"""Design experiment (round 2): PRBS-31 loopback cut into pieces that stay in the 256 MiB Infinity Cache.
fill(i) then check(i) per piece, on one stream or pipelined over two streams; wall time for 1e10 bits."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib

L = _lib.lib()
nbits = 10_000_000_000
nwords = (nbits + 63) // 64
buf = torch.empty(nwords, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def fill(first, n, st):
    rc = L.bbb_prbs_fill(31, 1, first, n, C.c_void_p(buf.data_ptr() + first // 8), 0, C.c_void_p(st.cuda_stream))
    assert rc == 0, L.bbb_last_error_detail()


def check(first, n, st):
    rc = L.bbb_prbs_check_dev(31, 1, first, n, C.c_void_p(buf.data_ptr() + first // 8), C.c_void_p(cnt.data_ptr()), 0, C.c_void_p(st.cuda_stream))
    assert rc == 0, L.bbb_last_error_detail()


def run(piece_bits, two_streams):
    pieces = [(f, min(piece_bits, nbits - f)) for f in range(0, nbits, piece_bits)]
    cnt.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not two_streams:
        for f, n in pieces:
            fill(f, n, sA)
            check(f, n, sA)
    else:
        evs = []
        for f, n in pieces:
            fill(f, n, sA)
            e = torch.cuda.Event()
            e.record(sA)
            sB.wait_event(e)
            check(f, n, sB)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, int(cnt.item())


for _ in range(2):
    run(nbits, False)
print("piece_MiB streams ms effective_TB/s errors")
for pb in (nbits, 1 << 32, 1 << 31, 1 << 30, 1 << 29, 1 << 28):
    for two in (False, True):
        best = min(run(pb, two) for _ in range(5))
        print(f"{pb/8/2**20:9.0f} {2 if two else 1} {best[0]*1e3:8.3f} {2*nbits/8/best[0]/1e12:6.2f} {best[1]}", flush=True)
