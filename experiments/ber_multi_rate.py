"""Round 5: the 11-point sweep through the C ABI's own entries, each call ALONE (synchronised before and after):
bbb_ber_trials (synchronous), bbb_ber_sweep_multi over this one device, and BASELINE configs[4]'s 88 trials (11 points x 8
seeds, the seeds as stretches 2^48 apart of the one cycle) on ONE device through bbb_ber_sweep_multi(BBB_SHARD_GROUPS):
the N = 1 figure an 8-GPU run of the same call is divided by."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
if os.environ.get("EXP"): bbb._lib.select_build("experiments")
from basebandboard_amd import channel, _lib
u = bbb.LUTOPT.shipped(256)
nv = 8
mk = lambda fb, seed=0: [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db, nv), noise_var=nv, first_bit=fb, warmup=16 + (seed << 48)) for db in range(11)]
for i in range(3):
    channel.run_trials(u, mk((i + 1) << 20))
    channel.sweep_multi([u], mk((i + 5) << 20))
torch.cuda.synchronize()
for name, f in (("bbb_ber_trials", lambda ts: channel.run_trials(u, ts)), ("bbb_ber_sweep_multi x1", lambda ts: channel.sweep_multi([u], ts))):
    dts = []
    for i in range(8):
        ts = mk((i + 10) << 21)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f(ts)
        dts.append(time.perf_counter() - t0)
    print(f"{name}: " + " ".join(f"{x*1e3:.4f}" for x in dts) + f" ms; median {sorted(dts)[4]*1e3:.4f} ms = {11e9/sorted(dts)[4]/1e12:.2f} Tbit/s", flush=True)
# 88 trials on one device
for rep in range(4):
    ts = [t for s in range(8) for t in mk((rep + 40) << 21, s)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = channel.sweep_multi([u], ts, mode=_lib.SHARD_GROUPS)
    dt = time.perf_counter() - t0
    print(f"88 trials (8 seeds x 11 points) on one device: {dt*1e3:.4f} ms = {88e9/dt/1e12:.2f} Tbit/s; per sweep {dt/8*1e3:.4f} ms", flush=True)
