"""n512 (packed kernel, int16 out): steady-state rate of the stream with prefetch hints, after a clock ramp."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
u = bbb.LUTOPT.shipped(512); g = bbb.CLTGRNG(u)
n = 1 << 28
buf = torch.empty(n, dtype=torch.int16, device="cuda")
def loop(k, s0):
    for i in range(s0, s0 + k):
        g.generate(n, first_step=18 + i * n, out=buf)
        g.prefetch(n, first_step=18 + (i + 1) * n)
loop(60, 0); torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); loop(60, 60 * (rep + 1)); b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 60
    print(f"n512 stream with prefetch: {dt:.3f} ms per fill of 2^28 = {n / dt / 1e6:.1f} Gsample/s", flush=True)
