"""11-point sweep of 1e9 bits (one noise pass), every sweep at another stream position so that it seeds for itself."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import channel
u = bbb.LUTOPT.shipped(256)
nv = 8
mk = lambda fb: [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=fb) for db in range(11)]
run = channel.gpu_runner(u)
channel.sweep_seeds(mk(0), run, world=1); channel.sweep_seeds(mk(1 << 20), run, world=1)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(10):
        channel.sweep_seeds(mk((i + 2 + 10 * rep) << 20), run, world=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"sweep with its own seeding: {dt*1e3:.4f} ms = {11e9/dt/1e12:.2f} Tbit/s", flush=True)
