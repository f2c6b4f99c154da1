"""Design check: a non-shipped k = 256 matrix through LUTOPT.specialise (compile time, rate, parity)."""
import time, tempfile, numpy as np, torch, basebandboard_amd as bbb
import oracle as O
packed = bbb.recurrences.n256
perm = packed[1:] + packed[:1]
u = bbb.LUTOPT.from_packed(perm, init=12345)
g = bbb.CLTGRNG(u)
n = 200_000_000
a = g.generate(n, first_step=3)
torch.cuda.synchronize(); t = time.perf_counter(); g.generate(n, first_step=3 + n); torch.cuda.synchronize(); print("table-driven %.2f Gsample/s" % (n / (time.perf_counter() - t) / 1e9), flush=True)
t = time.perf_counter(); u.specialise(build_dir=tempfile.mkdtemp()); print("specialise %.1f s" % (time.perf_counter() - t), flush=True)
b = g.generate(n, first_step=3)
torch.cuda.synchronize(); t = time.perf_counter(); g.generate(n, first_step=3 + n); torch.cuda.synchronize(); print("own kernel %.2f Gsample/s" % (n / (time.perf_counter() - t) / 1e9), flush=True)
print("equal", torch.equal(a, b), np.array_equal(b[:100000].cpu().numpy(), O.Lutopt(packed=perm).awgn(12345, 3, 100000)))
