import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
if os.environ.get('BBB_EXP'): _l.select_build('experiments')
import basebandboard_amd as bbb
N = 1_000_000_000
u = bbb.LUTOPT.shipped(256); u.set_staged(True)
g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device="cuda")
first = lambda s: 16 + s * N
for s in range(8):
    g.generate(N, first_step=first(s), out=buf)
    g.prefetch(N, first_step=first(s + 1))
torch.cuda.synchronize()
