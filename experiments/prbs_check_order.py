"""Is the checker slower right after the generator wrote the buffer? (design experiment)"""
import torch, basebandboard_amd as g
for nbits in (5_000_000_000, 10_000_000_000, 40_000_000_000):
    p = g.PRBS(31); det = g.PRBSErrorDetector(31)
    buf = p.generate(nbits)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    torch.cuda.synchronize()
    for rep in range(2):
        ev[0].record(); p.generate(nbits, out=buf); ev[1].record()
        n1 = det.count_errors(buf, nbits); ev[2].record()
        n2 = det.count_errors(buf, nbits); ev[3].record()
        n3 = det.count_errors(buf, nbits); ev[4].record()
        torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
        gb = nbits / 8e9
        print(nbits, "gen %.1f GB/s" % (gb / t[0] * 1e3), "check after gen %.1f" % (gb / t[1] * 1e3), "2nd %.1f" % (gb / t[2] * 1e3), "3rd %.1f" % (gb / t[3] * 1e3), n1 + n2 + n3, flush=True)
