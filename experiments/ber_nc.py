"""Design experiment: duration of the fused BER kernel against the number of settings in the group (one noise pass each).
Run under rocprofv3 --kernel-trace --stats and read the per-instance averages."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, basebandboard_amd as g
from basebandboard_amd import channel
u = g.LUTOPT.shipped(256)
nv = 8
for n in (1, 2, 4, 6, 7, 8, 10, 11, 12):
    trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db % 11, nv) + db // 11, noise_var=nv) for db in range(n)]
    for _ in range(4):
        out = g.run_trials(u, trials)
torch.cuda.synchronize()
