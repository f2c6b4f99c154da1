"""Design experiment: PRBS-31 generate/check time vs size (separates the per-wave bootstrap from streaming)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
p = bbb.PRBS(31); det = bbb.PRBSErrorDetector(31)
for nbits in (1_000_000_000, 2_500_000_000, 5_000_000_000, 10_000_000_000, 20_000_000_000, 40_000_000_000):
    buf = torch.empty((nbits + 63)//64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        p.generate(nbits, out=buf); det.count_errors(buf, nbits)
    torch.cuda.synchronize()
    del buf
