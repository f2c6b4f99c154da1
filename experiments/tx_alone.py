"""The TX sample kernel ALONE (a device sync after every call, no prefetch hint): with and without the data bits
(bit_en), staged and not, beside the noise-only kernel of the same size.  Run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
n = 1 << 29
for staged in (False, True):
    for bit_en in (1, 0):
        tx = bbb.TX(31, bit_en, 0, 16, 1, 8)
        tx.urng.set_staged(staged)
        buf = torch.empty(n, dtype=torch.int16, device="cuda")
        for i in range(6):
            tx.generate(n, first_sample=i * n, out=buf, stream_on=False)
            torch.cuda.synchronize()
        del tx
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged)
    g = bbb.CLTGRNG(u)
    b8 = torch.empty(n, dtype=torch.int8, device="cuda")
    for i in range(6):
        g.generate(n, first_step=16 + i * n, out=b8)
        torch.cuda.synchronize()
