import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
u = bbb.LUTOPT.shipped(256)
n = 1 << 26
buf = torch.empty(n * 8, dtype=torch.int32, device="cuda")
for i in range(3):
    u.generate_words(n, first_step=i * n, out=buf)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, 6):
    u.generate_words(n, first_step=i * n, out=buf)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"word stream n256: {n} states in {dt*1e3:.3f} ms = {n/dt/1e9:.1f} G states/s = {32*n/dt/1e12:.2f} TB/s written")
