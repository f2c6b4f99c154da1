"""Round 5: the noise stream at look-ahead levels 2 (the stream object's choice), 3, 4 and 8, as bench.py times it (1e9 int8 samples per read,
200 reads after a warm-up); product build, one process, each level twice."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
buf = torch.empty(N, dtype=torch.int8, device="cuda:0")
for rep in range(2):
    for level in (2, 3, 4, 8):
        u = bbb.LUTOPT.shipped(256, init=1, device=0)
        u.set_staged(True, look_ahead=level)
        g = bbb.CLTGRNG(u)
        st = g.stream(N, first_step=16)
        for _ in range(48):
            st.next(out=buf)
        out = []
        for K in (24, 192):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                st.next(out=buf)
            torch.cuda.synchronize()
            out.append(N * K / (time.perf_counter() - t0) / 1e9)
        st.close()
        print(f"level {level}: noise stream {out[0]:.1f} Gsample/s over 24 reads, {out[1]:.1f} over 192", flush=True)
        del st, g, u
