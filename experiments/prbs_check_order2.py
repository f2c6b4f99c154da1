"""Does idle time between fill and check matter? (design experiment)"""
import time, torch, basebandboard_amd as g
nbits = 10_000_000_000
p = g.PRBS(31); det = g.PRBSErrorDetector(31)
buf = p.generate(nbits)
other = torch.empty(400_000_000, dtype=torch.int64, device=buf.device)
gb = nbits / 8e9
for label, between in [("nothing", lambda: None), ("sync+5ms sleep", lambda: (torch.cuda.synchronize(), time.sleep(0.005))),
                       ("read 3.2 GB of other data", lambda: other.sum()), ("write 3.2 GB of other data", lambda: other.fill_(1))]:
    for rep in range(2):
        p.generate(nbits, out=buf)
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); n = det.count_errors(buf, nbits); e1.record(); torch.cuda.synchronize()
        print(label, "check %.1f GB/s" % (gb / e0.elapsed_time(e1) * 1e3), n, flush=True)
