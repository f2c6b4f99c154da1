"""Detector host flow A/B (experiments build): BBB_DET_ONE_TRIP = 1 (speculative second pass queued before the host looks)
against 0 (verify, look, re-run, verify), k = 31 (always a few inconsistent chunks) and k = 23 (none)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as g
g._lib.select_build("experiments")
nbits = 10_000_000_000
for k in (31, 23):
    p = g.PRBS(k); det = g.PRBSErrorDetector(k)
    buf = p.generate(nbits)
    noise = torch.randint(0, 1000, (buf.numel(),), device=buf.device) == 0
    buf ^= noise.to(torch.int64) << 13
    del noise
    det.run_stream(buf, nbits)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t = time.perf_counter()
        st = det.run_stream(buf, nbits)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"one_trip={os.environ.get('BBB_DET_ONE_TRIP')} k={k}: median {ts[3]*1e3:.3f} ms, best {ts[0]*1e3:.3f} ms, rerun {st['chunks_rerun']}", flush=True)
    del buf
