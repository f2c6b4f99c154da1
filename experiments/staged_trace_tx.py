import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
ntx = int(os.environ.get("NTX_LOG2", "29")) and (1 << int(os.environ.get("NTX_LOG2", "29")))
tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(True)
buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
for i in range(8):
    tx.generate(ntx, first_sample=i * ntx, out=buf)
torch.cuda.synchronize()
