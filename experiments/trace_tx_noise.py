import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
what = sys.argv[1] if len(sys.argv) > 1 else "tx"
if what == "tx":
    tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(True)
    buf = torch.empty(N, dtype=torch.int16, device="cuda")
    for i in range(12):
        tx.generate(N, first_sample=i * N, out=buf)
else:
    u = bbb.LUTOPT.shipped(256); g = bbb.CLTGRNG(u)
    buf = torch.empty(N, dtype=torch.int8, device="cuda")
    with g.stream(N, first_step=16) as st:
        for i in range(12):
            st.next(out=buf)
torch.cuda.synchronize()
