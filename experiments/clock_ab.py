"""Round 5: cycles per wave and the shader clock of the sample kernel inside the noise stream -- for one-off variant libraries whose
sample kernel stamps s_memtime / s_memrealtime at both ends into a __device__ array (experiments/build_variant.py, the `st*` variants;
bbb_var_read_stamps).  The stamps outside the loop leave the kernel's loop as the product's.  usage: clock_ab.py <lib.so>"""
import sys, os, time, pathlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
L = _lib.lib()
L.bbb_var_read_stamps.argtypes = [C.c_void_p]; L.bbb_var_read_stamps.restype = C.c_int
N = 1_000_000_000
u = bbb.LUTOPT.shipped(256, init=1, device=0)
g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device="cuda:0")
st = g.stream(N, first_step=16)
for _ in range(70):
    st.next(out=buf)
out = []
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        st.next(out=buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = np.zeros(4 * 1024, dtype=np.uint64)
    assert L.bbb_var_read_stamps(h.ctypes.data) == 0
    h = h.reshape(1024, 4)[:1018].astype(np.float64)
    cyc = h[:, 1] - h[:, 0]; real = (h[:, 3] - h[:, 2]) / 100e6
    out.append(f"{N * 100 / dt / 1e9:.1f} Gsample/s; last sample kernel: {cyc.mean() / 1e6:.4f} M cycles per wave, {real.mean() * 1e3:.4f} ms per wave, {np.mean(cyc / real) / 1e9:.4f} GHz")
print(os.path.basename(sys.argv[1]) + ": " + " | ".join(out), flush=True)
st.close()
