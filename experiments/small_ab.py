"""Round 4: the noise stream at two reads per kernel with the 434-register form of the sample kernel (guests one at a time)
against the 344-register form (BBB_EXP_NOISE_SMALL=1: both guests at once), one process per mode (experiments build)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
bbb._lib.select_build("experiments")
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda")
u = bbb.LUTOPT.shipped(256)
g = bbb.CLTGRNG(u)
with g.stream(N, first_step=16) as s:
    for _ in range(64): s.next(buf)
    res = []
    for k in (20, 200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): s.next(buf)
        torch.cuda.synchronize(); res.append(N / ((time.perf_counter() - t0) / k) / 1e9)
print(f"small={os.environ.get('BBB_EXP_NOISE_SMALL', '0')}: K=20 {res[0]:.1f}, K=200 {res[1]:.1f} Gsample/s", flush=True)
