"""Round 3: bbb_prbs_detector_stream at 1e10 bits, 1e-3 injected errors: wall time per call (as bench.py measures it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")
nbits = 10_000_000_000
gen = bbb.PRBS(31)
pbuf = gen.generate(nbits)
noise = torch.randint(0, 1000, (pbuf.numel(),), device=pbuf.device) == 0
pbuf ^= noise.to(torch.int64) << 13
del noise
det = bbb.PRBSErrorDetector(31)
for rep in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ds = det.run_stream(pbuf, nbits)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    print(f"call {rep}: {t * 1e3:.4f} ms = {nbits / t / 1e9:.0f} Gbit/s; errors {ds['errors']}, chunks_rerun {ds['chunks_rerun']}", flush=True)
