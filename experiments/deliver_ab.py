"""Round 4: the look-ahead delivery without / with the handover event on the next slot's arithmetic stream, A/B/A/B in one
process (experiments build: BBB_EXP_DELIVER_HANDOVER=1 restores rounds 2-3).  The stream object as bench.py drives it."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
bbb._lib.select_build("experiments")
libc = ctypes.CDLL(None)
N = 1_000_000_000
K = int(os.environ.get("K", "40"))
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for rnd in range(3):
    for knob in (1, 0):
        libc.setenv(b"BBB_EXP_DELIVER_HANDOVER", str(knob).encode(), 1)
        u = bbb.LUTOPT.shipped(256)
        g = bbb.CLTGRNG(u)
        with g.stream(N, first_step=16) as s:
            for _ in range(9): s.next(buf)
            u.profile(True); u.profile_read(reset=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): s.next(buf)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
            seed_ms, kern_ms, calls = u.profile_read(reset=True)
        print(f"handover={knob}: {dt*1e3:.4f} ms/step = {N/dt/1e9:.1f} Gsample/s; sample kernel by its events {kern_ms/max(calls,1):.4f} ms per launch ({calls})", flush=True)
        del u, g
