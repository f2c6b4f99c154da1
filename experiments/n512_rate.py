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

# the stream with prefetch hints: the seeding of fill s+1 beside the kernel of fill s
buf = torch.empty(n, dtype=torch.int16, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8):
        g.generate(n, first_step=18 + (4 + 8 * rep + i) * n, out=buf)
        g.prefetch(n, first_step=18 + (5 + 8 * rep + i) * n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"n512 stream with prefetch: {dt*1e3:.3f} ms per fill = {n/dt/1e9:.1f} Gsample/s")
