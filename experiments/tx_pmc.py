"""Round 5: the transmitter stream's kernels under counter passes (SQ_INSTS_VALU per launch): how much vector issue the noise kernel, the
shaping movers, the start-state derivation and the data-bit generator ask for per 2e9 samples -- the sum the stream is bound by
(DESIGN.md 3.6).  Twelve calls of 1e9 samples."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
tx = bbb.TX(31, 1, 0, 16, 1, 8, device=0)
txbuf = torch.empty(N, dtype=torch.int16, device="cuda:0")
stx = tx.stream(N)
for i in range(12):
    stx.next(out=txbuf)
torch.cuda.synchronize()
stx.close()
print("ok")
