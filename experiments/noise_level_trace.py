"""Round 5: the noise stream at a given look-ahead level for a kernel trace (usage: noise_level_trace.py <level>): 64 reads of 1e9 samples."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
level = int(sys.argv[1])
buf = torch.empty(N, dtype=torch.int8, device="cuda:0")
u = bbb.LUTOPT.shipped(256, init=1, device=0)
u.set_staged(True, look_ahead=level)
g = bbb.CLTGRNG(u)
st = g.stream(N, first_step=16)
for _ in range(64):
    st.next(out=buf)
torch.cuda.synchronize()
st.close()
