"""Round 5: four isolated 11-point sweeps at four stream positions (each derives its own start states), for kernel traces and counter
passes over seed_head_kernel / seed_tail_planes_kernel / ber256_fused_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, basebandboard_amd as g
from basebandboard_amd import channel
if os.environ.get("EXP"): g._lib.select_build("experiments")
u = g.LUTOPT.shipped(256)
nv = 8
for i in range(4):
    trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=i << 21) for db in range(11)]
    out = g.run_trials(u, trials)
    torch.cuda.synchronize()
print(out[:2])
