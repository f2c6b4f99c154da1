"""Round 3: same-box A/B of two builds of the library (AB_LIB) on the TX waveform, 1e9 samples per call, levels 1, 2, 4."""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
if os.environ.get("AB_LIB"):
    _l.LIB_PATH = pathlib.Path(os.environ["AB_LIB"]).resolve()
import basebandboard_amd as bbb
N = 1_000_000_000
ev = lambda: torch.cuda.Event(enable_timing=True)
tb = torch.empty(N, dtype=torch.int16, device="cuda")
for la in (0, 2, 4):
    tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(True, look_ahead=la if la >= 2 else False)
    for i in range(40): tx.generate(N, first_sample=i * N, out=tb)
    torch.cuda.synchronize()
    a, b = ev(), ev(); a.record()
    for i in range(40, 100): tx.generate(N, first_sample=i * N, out=tb)
    b.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('AB_LIB', 'product')}: TX level {max(la, 1)}: {a.elapsed_time(b) / 60:.4f} ms/call = {60e3 / a.elapsed_time(b):.1f} Gsample/s", flush=True)
    del tx
