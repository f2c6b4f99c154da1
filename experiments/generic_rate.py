"""Rate of the table-driven kernel for the other shipped matrices (design check)."""
import time, torch, basebandboard_amd as g
for n, ns in ((16, 50_000_000), (64, 50_000_000), (128, 50_000_000), (512, 50_000_000), (256, 1_000_000_000)):
    u = g.LUTOPT.shipped(n)
    c = g.CLTGRNG(u)
    x = c.generate(ns, first_step=16)
    torch.cuda.synchronize(); t = time.perf_counter()
    x = c.generate(ns, first_step=16 + ns)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(n, x.dtype, "%.2f Gsample/s" % (ns / dt / 1e9), flush=True)
