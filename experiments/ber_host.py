"""Round 4: host time of one 11-point sweep call (the call returns before the GPU is done) and the isolated call's wall time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import channel
u = bbb.LUTOPT.shipped(256)
nv = 8
mk = lambda fb: [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=fb) for db in range(11)]
c = torch.zeros((11, 2), dtype=torch.int64, device="cuda")
bbb.run_trials_into(u, mk(0), c); bbb.run_trials_into(u, mk(1 << 20), c)
torch.cuda.synchronize()
for rep in range(5):
    ts = mk((rep + 2) << 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bbb.run_trials_into(u, ts, c)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"isolated call: host returns after {1e3*(t1-t0):.3f} ms, GPU done after {1e3*(t2-t0):.3f} ms", flush=True)
