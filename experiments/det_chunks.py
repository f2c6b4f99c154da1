"""Round 4: bbb_prbs_detector_stream at 1e10 bits, 1e-3 injected word errors, against the chunk size (sparse form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
nbits = 10_000_000_000
gen = bbb.PRBS(31)
pbuf = gen.generate(nbits)
noise = torch.randint(0, 1000, (pbuf.numel(),), device=pbuf.device) == 0
pbuf ^= noise.to(torch.int64) << 13
del noise
det = bbb.PRBSErrorDetector(31)
for cb in (0, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ds = det.run_stream(pbuf, nbits, chunk_bits=cb)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    t = min(ts[1:])
    print(f"chunk_bits {cb:8d}: {t * 1e3:.4f} ms = {nbits / t / 1e9:.0f} Gbit/s; errors {ds['errors']}, chunks {ds['chunks']}, chunks_rerun {ds['chunks_rerun']}", flush=True)
