"""Round 4: host time INSIDE bbb_ber_trials_dev for the first calls of a process (11 trials of 1e9 bits, each call alone)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import channel
u = bbb.LUTOPT.shipped(256)
nv = 8
mk = lambda fb: [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=fb) for db in range(11)]
c = torch.zeros((8, 11, 2), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for i in range(6):
    ts = mk((i + 1) << 21)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    channel.run_trials_into(u, ts, c[i])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"call {i}: host inside the call {1e6*(t1-t0):.0f} us, until the GPU is done {1e6*(t2-t0):.0f} us", flush=True)
