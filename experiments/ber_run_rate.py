"""11-point sweep continued over many calls (bbb_ber_run_*): 1e9 bits per point and call, one seeding per m calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import channel
u = bbb.LUTOPT.shipped(256)
nv = 8
trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv) for db in range(11)]
for m in (1, 2, 4, 8, 16):
    with channel.ContinuedTrials(u, trials, m) as run:
        for _ in range(m):
            run.next(read=False)
        torch.cuda.synchronize()
        ncall = 4 * m if m >= 4 else 16
        t0 = time.perf_counter()
        for _ in range(ncall - 1):
            run.next(read=False)
        tot = run.next()
        dt = (time.perf_counter() - t0) / ncall
        print(f"m = {m:2d}: {dt*1e3:.4f} ms per call of 11 x 1e9 bits = {11e9/dt/1e12:.2f} Tbit/s (seeding every {m} calls, inside); totals {tot[0]} ... {tot[-1]}", flush=True)
