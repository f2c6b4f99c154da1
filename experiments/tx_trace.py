"""Round 4: the transmitter stream in steady state for a kernel trace (rocprofv3 --kernel-trace -- python3 experiments/tx_trace.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")
N = 1_000_000_000
buf16 = torch.empty(N, dtype=torch.int16, device="cuda")
x = bbb.TX(31, 1, 0, 16, 1, 8)
if os.environ.get("LEVEL"): x.urng.set_staged(True, look_ahead=int(os.environ["LEVEL"]))
with x.stream(N, first_sample=0) as st:
    for _ in range(30): st.next(buf16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st.next(buf16)
    torch.cuda.synchronize(); dtx = (time.perf_counter() - t0) / 20
print(f"TX {dtx*1e3:.4f} ms per call = {N/dtx/1e9:.1f} Gsample/s")
