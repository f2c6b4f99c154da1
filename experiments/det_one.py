import time, torch, basebandboard_amd as g
nbits = 10_000_000_000
p = g.PRBS(31); det = g.PRBSErrorDetector(31)
buf = p.generate(nbits)
noise = torch.randint(0, 1000, (buf.numel(),), device=buf.device) == 0
buf ^= noise.to(torch.int64) << 13
del noise
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    st = det.run_stream(buf, nbits)
    torch.cuda.synchronize(); print(round((time.perf_counter() - t) * 1e3, 2), "ms", st["chunks_rerun"], flush=True)
