"""Staged TX (noise kernel + shaping mover), 1e9 samples per call: rate per mover width (BBB_TXMOVER_THREADS, experiments build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib
if os.environ.get("BBB_EXP"):
    _lib.select_build("experiments")
import basebandboard_amd as bbb
ntx = 1_000_000_000
tx = bbb.TX(31, 1, 0, 16, 1, 8)
tx.urng.set_staged(True)
buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
for i in range(4):
    tx.generate(ntx, first_sample=i * ntx, out=buf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(4, 14):
    tx.generate(ntx, first_sample=i * ntx, out=buf)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"mover threads {os.environ.get('BBB_TXMOVER_THREADS', 'default')}: {ms:.4f} ms per call = {ntx/ms/1e6:.1f} Gsample/s", flush=True)
