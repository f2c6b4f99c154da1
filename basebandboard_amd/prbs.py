"""PRBS bit sequences -- host-side mirror of gateware/bbb/prbs.py.

`PRBS(k)` and `PRBSErrorDetector(k)` keep the reference's constructor arguments and its
`ValueError("k=... invalid for PRBS")` (prbs.py:29-30, 55-56).  Streams are packed 64 bits per
word, LSB first, in int64 CUDA tensors (torch has no general uint64; only the bit pattern matters).
"""
import ctypes as C

import torch

from . import _lib

# k -> the second feedback tap of x^k + x^tap + 1 (the first is k itself); table of prbs.py:14.
TAPS = {7: 6, 9: 5, 11: 9, 15: 14, 20: 3, 23: 18, 31: 28}


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check_k(k):
    if k not in TAPS.keys():
        raise ValueError("k={} invalid for PRBS".format(k))


class PRBS:
    """PRBS-k source (mirror of prbs.py:17-35); `generate` gives the bits its output `x` takes on
    successive clocks, packed."""

    def __init__(self, k, init=1, device=0):
        _check_k(k)
        if not 0 < init < (1 << k):
            raise ValueError("PRBS state must be in [1, 2^k)")
        self.k, self.init, self.device = k, int(init), int(device)

    def state_at(self, nbits):
        s = C.c_uint64()
        _lib.check(_lib.lib().bbb_prbs_state_at(self.k, self.init, int(nbits), C.byref(s)), "bbb_prbs_state_at")
        return s.value

    def generate(self, nbits, first_bit=0, out=None, will_read_back=False):
        """will_read_back: hint that a checker follows right behind (bbb_prbs_fill_hint): same bits, the fill leaves no
        dirty lines in the memory-side cache for the reader to write back."""
        nwords = (int(nbits) + 63) // 64
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty(nwords, dtype=torch.int64, device=dev)
        if out.dtype != torch.int64 or out.numel() < nwords or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous int64 CUDA tensor with >= ceil(nbits/64) words")
        _lib.check(_lib.lib().bbb_prbs_fill_hint(self.k, self.init, int(first_bit), int(nbits), C.c_void_p(out.data_ptr()),
                                                 1 if will_read_back else 0, self.device, _stream_ptr(self.device)), "bbb_prbs_fill")
        return out[:nwords]


class PRBSErrorDetector:
    """Error detector for a received PRBS-k stream (mirror of prbs.py:38-99): exact
    self-synchronising FSM (`run`) or phase-known bulk comparison (`count_errors`)."""

    def __init__(self, k, device=0):
        _check_k(k)
        self.k, self.device = k, int(device)

    def run(self, bits):
        """Cycle-exact detector on a uint8 CUDA tensor [nstreams, n] of 0/1 input bits, one
        independent detector per row.  Returns (err, reload) with the values the reference's
        testbench reads after each clock (prbs.py:146-150)."""
        if bits.dtype != torch.uint8 or bits.dim() != 2 or not bits.is_cuda:
            raise ValueError("bits must be a uint8 CUDA tensor [nstreams, n]")
        bits = bits.contiguous()
        err = torch.empty_like(bits)
        reload = torch.empty_like(bits)
        _lib.check(_lib.lib().bbb_prbs_detector_run(self.k, C.c_void_p(bits.data_ptr()), bits.shape[0], bits.shape[1],
                                                    C.c_void_p(err.data_ptr()), C.c_void_p(reload.data_ptr()),
                                                    self.device, _stream_ptr(self.device)), "bbb_prbs_detector_run")
        return err, reload

    def run_stream(self, packed, nbits, want_err=False, want_reload=False, chunk_bits=0, warm_bits=0):
        """The same machine over ONE long packed stream (bit t at word t//64, LSB first), executed in
        parallel chunks with state hand-off; exact (bbb_prbs_detector_stream).  Returns a dict of
        totals (errors = err while reload == 0, errors_raw, reload_clocks, resyncs, ...) and, when
        asked for, the packed `err` / `reload` streams as int64 CUDA tensors."""
        if packed.dtype != torch.int64 or not packed.is_cuda or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous int64 CUDA tensor")
        if packed.numel() * 64 < nbits:
            raise ValueError("packed holds fewer than nbits bits")
        nw = (int(nbits) + 63) // 64
        err = torch.empty(nw, dtype=torch.int64, device=packed.device) if want_err else None
        rl = torch.empty(nw, dtype=torch.int64, device=packed.device) if want_reload else None
        st = _lib.DetectorStats()
        _lib.check(_lib.lib().bbb_prbs_detector_stream(
            self.k, C.c_void_p(packed.data_ptr()), int(nbits), C.c_void_p(err.data_ptr() if want_err else None),
            C.c_void_p(rl.data_ptr() if want_reload else None), C.byref(st), int(chunk_bits), int(warm_bits),
            self.device, _stream_ptr(self.device)), "bbb_prbs_detector_stream")
        out = {n: int(getattr(st, n)) for n, _ in _lib.DetectorStats._fields_}
        if want_err:
            out["err"] = err
        if want_reload:
            out["reload"] = rl
        return out

    def count_errors(self, packed, nbits, first_bit=0, init=1):
        """Phase-known bulk check: how many of `nbits` packed bits differ from PRBSk started at
        `init` -- the sum of `err` over a stream for which `reload` stays 0 (prbs.py:79)."""
        if packed.dtype != torch.int64 or not packed.is_cuda or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous int64 CUDA tensor")
        if packed.numel() * 64 < nbits:
            raise ValueError("packed holds fewer than nbits bits")
        n = C.c_uint64()
        _lib.check(_lib.lib().bbb_prbs_check(self.k, int(init), int(first_bit), int(nbits), C.c_void_p(packed.data_ptr()),
                                             C.byref(n), self.device, _stream_ptr(self.device)), "bbb_prbs_check")
        return n.value
