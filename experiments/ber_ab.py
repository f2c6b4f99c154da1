"""Design experiment: the fused BER kernel in variant builds (AB_LIB = a libbbb_hip_vx*.so built with -DBBB_BER_X=n: 1 no PRBS
ring, 2 no comparators, 3 neither).  Kernel time by hipEvents around bbb_ber_trials_dev with the generator's start states
cached (same stream position every call), PRBS seeding included (~35 us)."""
import sys, os, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basebandboard_amd import _lib as _l
if os.environ.get("AB_LIB"):
    _l.LIB_PATH = pathlib.Path(os.environ["AB_LIB"]).resolve()
import torch, basebandboard_amd as g
from basebandboard_amd import channel
u = g.LUTOPT.shipped(256)
nv = 8
for n in (1, 11):
    trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db % 11, nv), noise_var=nv) for db in range(n)]
    c = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    for _ in range(3):
        g.run_trials_into(u, trials, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.run_trials_into(u, trials, c)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('AB_LIB', 'product')}: ncfg {n}: {e0.elapsed_time(e1) / 10:.4f} ms per call", flush=True)
