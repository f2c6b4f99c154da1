"""Design experiment: kernel timeline of one 11-point BER sweep call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, basebandboard_amd as g
from basebandboard_amd import channel
u = g.LUTOPT.shipped(256)
nv = 8
trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv) for db in range(11)]
for _ in range(3):
    out = g.run_trials(u, trials)
torch.cuda.synchronize()
print(out[:2])
