"""Round 4: one 11-point sweep of 1e9 bits per point, ALONE (synchronised before and after), several times: what an isolated
call costs -- with a fresh runner per call (bench.py's isolated_call) and with one runner kept."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import channel
u = bbb.LUTOPT.shipped(256)
nv = 8
mk = lambda fb: [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=fb) for db in range(11)]
channel.sweep_seeds(mk(1 << 20), channel.gpu_runner(u), world=1)
torch.cuda.synchronize()
kept = channel.gpu_runner(u)
for i in range(8):
    runner = kept if i >= 4 else channel.gpu_runner(u)
    ts = mk((i + 2) << 21)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    channel.sweep_seeds(ts, runner, world=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"isolated sweep {i} ({'kept' if i >= 4 else 'fresh'} runner): {dt*1e3:.4f} ms = {11e9/dt/1e12:.2f} Tbit/s", flush=True)
    time.sleep(0.05 if i % 2 else 0.0)
