"""Round 5: the transmitter stream at look-ahead levels 2 (the stream object's choice), 3, 4 and 6: one noise kernel per m calls -- fewer
start-state derivations beside the kernel per sample.  Product build; each level in this one process, twice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb
N = 1_000_000_000
ev = lambda: torch.cuda.Event(enable_timing=True)
txbuf = torch.empty(N, dtype=torch.int16, device="cuda:0")
for rep in range(2):
    for level in (2, 3, 4, 6):
        tx = bbb.TX(31, 1, 0, 16, 1, 8, device=0)
        tx.urng.set_staged(True, look_ahead=level)
        stx = tx.stream(N)
        for i in range(24):
            stx.next(out=txbuf)
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for i in range(48):
            stx.next(out=txbuf)
        b.record()
        torch.cuda.synchronize()
        stx.close()
        ms = a.elapsed_time(b) / 48
        print(f"level {level}: TX stream {N / ms / 1e6:.1f} Gsample/s ({ms:.4f} ms per call)", flush=True)
        del stx, tx
