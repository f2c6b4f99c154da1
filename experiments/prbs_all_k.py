"""PRBS fill / check rate for every order (design check)."""
import torch, basebandboard_amd as g
nbits = 10_000_000_000
for k in (7, 9, 11, 15, 20, 23, 31):
    p = g.PRBS(k); det = g.PRBSErrorDetector(k)
    buf = p.generate(nbits)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record(); p.generate(nbits, out=buf); ev[1].record()
    n1 = det.count_errors(buf, nbits); ev[2].record()
    n2 = det.count_errors(buf, nbits); ev[3].record()
    torch.cuda.synchronize()
    gb = nbits / 8e9
    print(k, "fill %.0f GB/s" % (gb / ev[0].elapsed_time(ev[1]) * 1e3), "check %.0f" % (gb / ev[1].elapsed_time(ev[2]) * 1e3), "again %.0f" % (gb / ev[2].elapsed_time(ev[3]) * 1e3), n1 + n2, flush=True)
