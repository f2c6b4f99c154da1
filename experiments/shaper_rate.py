"""PRBSShaper.x alone (bbb_shaper_fill_i16, noise off): 2^30 samples per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
n = 1 << 30
tx = bbb.TX(31, 1, 0, 16, 0, 8)          # bit_en 1, noise_en 0
buf = torch.empty(n, dtype=torch.int16, device="cuda")
for i in range(2):
    tx.generate(n, first_sample=i * n, out=buf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(2, 8):
    tx.generate(n, first_sample=i * n, out=buf)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 6
print(f"shaper only (PRBS fill + waveform kernel): {ms:.4f} ms per 2^30 samples = {n/ms/1e6:.1f} Gsample/s = {2*n/ms/1e9:.2f} TB/s written")
