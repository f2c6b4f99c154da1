"""Round 5: the noise stream and the transmitter stream (the stream objects, as bench.py times them) -- for A/B runs of the mover
(BBB_UNPLANE_BLOCKS_PER_CU and friends, experiments build when the first argument is 'exp')."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
from basebandboard_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "exp":
    _lib.select_build("experiments")
elif len(sys.argv) > 1 and sys.argv[1].endswith(".so"):      # a one-off variant library (experiments/build_variant.py)
    import pathlib
    _lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
N = 1_000_000_000
dev = "cuda:0"
ev = lambda: torch.cuda.Event(enable_timing=True)
u = bbb.LUTOPT.shipped(256, init=1, device=0)
g = bbb.CLTGRNG(u)
buf = torch.empty(N, dtype=torch.int8, device=dev)
st = g.stream(N, first_step=16)
for _ in range(70):
    st.next(out=buf)
res = []
for K in (20, 200):
    st.seek(st.tell())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        st.next(out=buf)
    torch.cuda.synchronize()
    res.append(N * K / (time.perf_counter() - t0) / 1e9)
st.close()
tx = bbb.TX(31, 1, 0, 16, 1, 8, device=0)
txbuf = torch.empty(N, dtype=torch.int16, device=dev)
stx = tx.stream(N)
for i in range(24):
    stx.next(out=txbuf)
torch.cuda.synchronize()
a, b = ev(), ev()
a.record()
for i in range(40):
    stx.next(out=txbuf)
b.record()
torch.cuda.synchronize()
stx.close()
txms = a.elapsed_time(b) / 40
print(f"noise stream {res[0]:.1f} Gsample/s over 20 steps, {res[1]:.1f} over 200; TX stream {N / txms / 1e6:.1f} Gsample/s ({txms:.4f} ms per call)", flush=True)
