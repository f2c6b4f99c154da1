"""Round 4: the mover on the caller's stream (BBB_EXP_MOVER_OWN_STREAM=1, experiments build) against the mover on its own
internal stream tied to the caller's by two events.  ONE mode per process (the environment decides), one handle per form, as
bench.py has it: noise stream over K steps and over 200, TX stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
bbb._lib.select_build("experiments")
N = 1_000_000_000
K = int(os.environ.get("K", "20"))
buf = torch.empty(N, dtype=torch.int8, device="cuda")
buf16 = torch.empty(N, dtype=torch.int16, device="cuda")
u = bbb.LUTOPT.shipped(256)
g = bbb.CLTGRNG(u)
res = []
with g.stream(N, first_step=16) as s:
    for _ in range(64): s.next(buf)
    for k in (K, 200, K):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): s.next(buf)
        torch.cuda.synchronize(); res.append(N / ((time.perf_counter() - t0) / k) / 1e9)
x = bbb.TX(31, 1, 0, 16, 1, 8)
with x.stream(N, first_sample=0) as st:
    for _ in range(20): st.next(buf16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): st.next(buf16)
    torch.cuda.synchronize(); dtx = (time.perf_counter() - t0) / 40
print(f"mover_own_stream={os.environ.get('BBB_EXP_MOVER_OWN_STREAM', '0')}: noise K={K}: {res[0]:.1f}, K=200: {res[1]:.1f}, K={K}: {res[2]:.1f} Gsample/s; TX {N/dtx/1e9:.1f} Gsample/s", flush=True)
