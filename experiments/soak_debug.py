"""Debug: replay the random-mix soak of tests/test_gpu_staged.py (seed 3) and report the first wrong result, in variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import basebandboard_amd as gpu
BIG = 1 << 24

def run(variant, seed=3, nops=24):
    rng = np.random.default_rng(seed)
    x = gpu.TX(31, 1, 0, 16, 1, 8); u = x.urng; g = gpu.CLTGRNG(u)
    y = gpu.TX(31, 1, 0, 16, 1, 8); d = gpu.CLTGRNG(y.urng)
    side = torch.cuda.Stream()
    bufs = [torch.empty(BIG + 8192, dtype=torch.int8, device="cuda") for _ in range(2)]
    pos, checks = 16, []
    u.set_staged(True, look_ahead=2)
    for it in range(nops):
        op = rng.integers(0, 10)
        n = BIG + 16 * int(rng.integers(0, 512))
        if op < 5:
            out = bufs[it & 1][:n]
            use_side = rng.integers(0, 4) == 0
            if variant == "noside": use_side_eff = False
            else: use_side_eff = use_side
            ctx = torch.cuda.stream(side) if use_side_eff else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                g.generate(n, first_step=pos, out=out)
                snap = out.clone()
            if rng.integers(0, 3) and variant != "noprefetch":
                g.prefetch(n, first_step=pos + n)
            checks.append((it, "awgn", n, pos, snap))
            pos += n
        elif op == 5:
            n2 = n + int(rng.integers(1, 16)); p2 = int(rng.integers(0, 1 << 40))
            checks.append((it, "awgn", n2, p2, g.generate(n2, first_step=p2)))
        elif op == 6:
            if variant == "notx":
                pass
            else:
                checks.append((it, "tx", n, pos, x.generate(n, first_sample=pos)))
        elif op == 7:
            t = gpu.Trial(nbits=200_000 + it, amp=90, noise_var=8, first_bit=it)
            gpu.run_trials(u, [t])
        elif op == 8:
            u.set_staged(True, look_ahead=int(rng.integers(1, 4)) if rng.integers(0, 2) else False)
        else:
            torch.cuda.synchronize()
        if variant == "syncall":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    bad = []
    for it, kind, n, p, got in checks:
        ref = d.generate(n, first_step=p) if kind == "awgn" else y.generate(n, first_sample=p)
        if not torch.equal(got, ref):
            diff = (got != ref).nonzero().flatten()
            bad.append((it, kind, n, p, int(diff.numel()), int(diff[0]), int(diff[-1])))
    print(variant, "bad:", bad, flush=True)

for v in ("asis", "noside", "notx", "noprefetch", "syncall"):
    run(v)
