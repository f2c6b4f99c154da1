"""Round 3: TX waveform rate (noise kernel + shaping planes mover), 1e9 samples per call, and the noise stream beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
ev = lambda: torch.cuda.Event(enable_timing=True)
def tx_rate(ntx, staged, reps=40, la=0):
    tx = bbb.TX(31, 1, 0, 16, 1, 8)
    tx.urng.set_staged(staged, look_ahead=la if la >= 2 else False)
    buf = torch.empty(ntx, dtype=torch.int16, device="cuda")
    for i in range(30):
        tx.generate(ntx, first_sample=i * ntx, out=buf)
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for i in range(30, 30 + reps):
        tx.generate(ntx, first_sample=i * ntx, out=buf)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for ntx, staged in ((1_000_000_000, True), (1 << 29, True), (1 << 29, False), (1_000_000_000, True)):
    ms = tx_rate(ntx, staged)
    print(f"TX {ntx} samples per call, staged={staged}: {ms:.4f} ms per call = {ntx / ms / 1e6:.1f} Gsample/s", flush=True)
for la in (2, 4, 2):
    ms = tx_rate(1_000_000_000, True, la=la)
    print(f"TX 1e9 samples per call, staged, look-ahead {la}: {ms:.4f} ms per call = {1e3 / ms:.1f} Gsample/s", flush=True)
