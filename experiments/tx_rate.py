"""TX waveform rate: bbb_tx_fill_i16 on 2^29 samples per call (as bench.py's extra)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
ntx = 1 << 29
tx = bbb.TX(31, 1, 0, 16, 1, 8)
buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
tx.generate(ntx, out=buf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5):
    tx.generate(ntx, first_sample=(i + 1) * ntx, out=buf)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"tx_waveform {ntx} samples: {ms:.4f} ms per call = {ntx/ms/1e6:.1f} Gsample/s = {2*ntx/ms/1e6:.0f} GB/s of int16 written")
