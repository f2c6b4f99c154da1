"""Design experiment: kernel breakdown of bbb_tx_fill_i16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
n = 1 << 29
tx = bbb.TX(31, 1, 0, 16, 1, 8)
buf = torch.empty(n, dtype=torch.int16, device="cuda")
for i in range(4):
    tx.generate(n, first_sample=i * n, out=buf)
torch.cuda.synchronize()
