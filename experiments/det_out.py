import time, torch, basebandboard_amd as g
nbits = 10_000_000_000
p = g.PRBS(31); det = g.PRBSErrorDetector(31)
buf = p.generate(nbits)
noise = torch.randint(0, 1000, (buf.numel(),), device=buf.device) == 0
buf ^= noise.to(torch.int64) << 13
del noise
for kw in ({}, {"want_err": True}, {"want_err": True, "want_reload": True}):
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        st = det.run_stream(buf, nbits, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(kw, round(dt * 1e3, 2), "ms", round(nbits / dt / 1e9), "Gbit/s", flush=True)
