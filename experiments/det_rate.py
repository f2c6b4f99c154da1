"""Exact detector over one 1e10-bit stream (one flipped bit per 1000 words), every PRBS order: ms and Tbit/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as g
nbits = 10_000_000_000
for k in (31, 23, 20, 15, 9, 7):
    p = g.PRBS(k); det = g.PRBSErrorDetector(k)
    buf = p.generate(nbits)
    noise = torch.randint(0, 1000, (buf.numel(),), device=buf.device) == 0
    buf ^= noise.to(torch.int64) << 13
    del noise
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        st = det.run_stream(buf, nbits)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"k={k}: {best*1e3:.3f} ms = {nbits/best/1e12:.2f} Tbit/s, errors {st['errors']}, resyncs {st['resyncs']}, chunks rerun {st['chunks_rerun']}", flush=True)
    del buf
