"""Does the TX rate depend on what ran before it in the process (bench.py's extras measure it after the AWGN loop)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
ntx = 1 << 29
def txrate(tag):
    tx = bbb.TX(31, 1, 0, 16, 1, 8)
    buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
    for i in range(2): tx.generate(ntx, first_sample=i * ntx, out=buf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2, 8): tx.generate(ntx, first_sample=i * ntx, out=buf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"{tag}: TX {dt*1e3:.4f} ms/call = {ntx/dt/1e9:.1f} Gsample/s")
txrate("fresh process")
N = 1_000_000_000
u = bbb.LUTOPT.shipped(256); u.set_staged(True); g = bbb.CLTGRNG(u)
b = torch.empty(N, dtype=torch.int8, device="cuda")
for s in range(12):
    g.generate(N, first_step=16 + s * N, out=b); g.prefetch(N, first_step=16 + (s + 1) * N)
torch.cuda.synchronize()
txrate("after 12 staged AWGN fills (handle alive)")
del u, g
torch.cuda.synchronize()
txrate("after deleting that handle")
big = torch.empty(3_000_000_000, dtype=torch.int8, device="cuda"); big.fill_(1); torch.cuda.synchronize()
txrate("with 3 GB more allocated")
