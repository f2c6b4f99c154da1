"""Round 5: bbb_prbs_detector_stream at 1e10 bits, 1e-3 injected errors per word, against the chunk size (the fused kernel takes
chunks of 128 .. 512 words; a wave owns 64 chunks): smaller chunks = more waves than are resident = the dispatcher staggers the
waves' streaming and serial phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
nbits = 10_000_000_000
gen = bbb.PRBS(31)
pbuf = gen.generate(nbits)
noise = torch.randint(0, 1000, (pbuf.numel(),), device=pbuf.device) == 0
pbuf ^= noise.to(torch.int64) << 13
del noise
det = bbb.PRBSErrorDetector(31)
ref = None
for rnd in range(3):
    for cb in (0, 32768, 40960, 57344):
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ds = det.run_stream(pbuf, nbits, chunk_bits=cb)
            ts.append(time.perf_counter() - t)
        key = (ds["errors"], ds["errors_raw"], ds["reload_clocks"], ds["resyncs"])
        ref = ref or key
        ts = sorted(ts[1:])
        print(f"chunk_bits {cb:6d}: median {ts[2] * 1e3:.4f} ms = {nbits / ts[2] / 1e9:.0f} Gbit/s (min {ts[0] * 1e3:.4f}); chunks {ds['chunks']}, rerun {ds['chunks_rerun']}, totals equal {key == ref}", flush=True)
