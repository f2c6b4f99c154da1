"""Timing of bbb_prbs_detector_stream over chunk / warm-up sizes (design experiment)."""
import time, torch, basebandboard_amd as g
nbits = 10_000_000_000
p = g.PRBS(31); det = g.PRBSErrorDetector(31)
buf = p.generate(nbits)
noise = torch.randint(0, 1000, (buf.numel(),), device=buf.device) == 0
buf ^= noise.to(torch.int64) << 13
del noise
for chunk, warm in [(4096, 1024), (4096, 512), (8192, 1024), (8192, 512), (16384, 1024), (16384, 512), (32768, 1024), (2048, 512)]:
    det.run_stream(buf, nbits, chunk_bits=chunk, warm_bits=warm)
    torch.cuda.synchronize(); t = time.perf_counter()
    st = det.run_stream(buf, nbits, chunk_bits=chunk, warm_bits=warm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(chunk, warm, round(dt * 1e3, 2), "ms", round(nbits / dt / 1e9, 1), "Gbit/s rerun", st["chunks_rerun"], "errors", st["errors"], flush=True)
