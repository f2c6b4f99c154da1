import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
u = bbb.LUTOPT.shipped(512); g = bbb.CLTGRNG(u)
n = 1 << 28
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.generate(n, first_step=18 + i * n); torch.cuda.synchronize()
    print(f"n512 {n} samples: {(time.perf_counter()-t0)*1e3:.3f} ms")
