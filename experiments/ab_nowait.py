"""Round 3: mover and seeding beside the sample kernel AT THE SAME TIME (experiments build, BBB_SEED_NO_WAIT=1: the seeding
does not wait for the slot's mover) against one after the other: noise at one read per kernel, TX at levels 1 and 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basebandboard_amd import _lib as _l
_l.select_build("experiments")
import basebandboard_amd as bbb
N = 1_000_000_000
ev = lambda: torch.cuda.Event(enable_timing=True)
tag = "no wait" if os.environ.get("BBB_SEED_NO_WAIT") else "seeding waits for the mover"
buf = torch.empty(N, dtype=torch.int8, device="cuda")
for LA in (0, 2):
    u = bbb.LUTOPT.shipped(256); u.set_staged(True, look_ahead=LA if LA >= 2 else False); g = bbb.CLTGRNG(u)
    def loop(k, s0):
        for s in range(s0, s0 + k):
            g.generate(N, first_step=16 + s * N, out=buf)
            g.prefetch(N, first_step=16 + (s + 1) * N)
    loop(100, 0); torch.cuda.synchronize()
    a, b = ev(), ev(); a.record(); loop(200, 100); b.record(); torch.cuda.synchronize()
    print(f"{tag}: noise level {max(LA, 1)}: {a.elapsed_time(b) / 200:.4f} ms/step", flush=True)
    del g, u
del buf
for la in (0, 2):
    tx = bbb.TX(31, 1, 0, 16, 1, 8); tx.urng.set_staged(True, look_ahead=la if la >= 2 else False)
    tb = torch.empty(N, dtype=torch.int16, device="cuda")
    for i in range(30): tx.generate(N, first_sample=i * N, out=tb)
    torch.cuda.synchronize()
    a, b = ev(), ev(); a.record()
    for i in range(30, 70): tx.generate(N, first_sample=i * N, out=tb)
    b.record(); torch.cuda.synchronize()
    print(f"{tag}: TX level {max(la, 1)}: {a.elapsed_time(b) / 40:.4f} ms/call = {40e3 / a.elapsed_time(b):.1f} Gsample/s", flush=True)
    del tx, tb
