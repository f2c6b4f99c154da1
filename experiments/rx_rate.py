"""Receiver slicer rate (bbb_rx_slice): 2^30 int16 samples, stride 8 / 4 / 1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
n = 1 << 30
x = torch.randint(-2000, 2000, (n,), dtype=torch.int16, device="cuda")
for stride in (8, 4, 2, 1):
    rx = bbb.RX(31, 8, 0)
    rx.slice(x, stride=stride)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rx.slice(x, stride=stride)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"stride {stride}: {ms:.4f} ms per 2^30 samples = {2*n/ms/1e9:.2f} TB/s of samples read, {n/stride/ms/1e6:.1f} Gbit/s decided")
