"""Design experiment (experiments build): per-wave shader-clock stamps of the fused BER kernel -- prologue, loop, epilogue."""
import sys, os, pathlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basebandboard_amd import _lib as _l
_l.LIB_PATH = pathlib.Path(os.environ.get("AB_LIB", "basebandboard_amd/libbbb_hip_exp.so")).resolve()
import numpy as np, torch, basebandboard_amd as g
from basebandboard_amd import channel
u = g.LUTOPT.shipped(256)
nv = 8
L = _l.lib()
for n in (1, 11):
    trials = [channel.Trial(nbits=1_000_000_000, amp=channel.amp_for_ebn0(db % 11, nv), noise_var=nv) for db in range(n)]
    c = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    for _ in range(3):
        g.run_trials_into(u, trials, c)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 8192)()
    assert L.bbb_exp_ber_debug_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:1017]
    pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    print(f"ncfg {n}: prologue {np.median(pro)} cycles (max {pro.max()}), loop {np.median(loop)} = {np.median(loop)/480:.1f} per step (min {loop.min()/480:.1f} max {loop.max()/480:.1f}), epilogue {np.median(epi)} (max {epi.max()}); loop real time {np.median(a[:,6]-a[:,5])/100:.1f} us -> clock {np.median(loop)/(np.median(a[:,6]-a[:,5])/100)/1e3:.3f} GHz; kernel real time first start to last end {(a[:,7].max()-a[:,4].min())/100:.1f} us", flush=True)
