"""Baseband receiver front end -- host-side mirror of gateware/bbb/rx.py.

`RX(prbs_k, samples_per_bit, sample_delay)` keeps the reference's parameters (rx.py:15-21): the
incoming samples are thresholded to single bits (`~sample[-1]`, rx.py:29), delayed by
`sample_delay` samples (BitDelayLine, rx.py:32-33) and handed to a PRBSErrorDetector once per bit.
Here the sample stream is a tensor in HBM: `slice` returns the decided bits packed 64 per word,
`count_errors` feeds them to the phase-known checker, `detect` to the exact detector FSM.
(The reference clocks its detector from bit 1 of a log2(samples_per_bit)-bit counter, i.e. every
4 samples whatever samples_per_bit is -- rx.py:35-39; `stride` defaults to samples_per_bit, pass 4
to reproduce that.)
"""
import ctypes as C

import torch

from . import _lib
from .prbs import PRBSErrorDetector, TAPS


class RX:
    def __init__(self, prbs_k, samples_per_bit, sample_delay, device=0):
        if prbs_k not in TAPS.keys():
            raise ValueError("k={} invalid for PRBS".format(prbs_k))
        if samples_per_bit < 1 or samples_per_bit & (samples_per_bit - 1):
            raise ValueError("samples_per_bit must be a power of 2")            # rx.py:18
        if not 0 <= sample_delay <= samples_per_bit:
            raise ValueError("sample_delay may not exceed the delay line length")  # delayline.py:54-55
        self.prbs_k, self.samples_per_bit, self.sample_delay, self.device = prbs_k, int(samples_per_bit), int(sample_delay), int(device)
        self.prbsdet = PRBSErrorDetector(prbs_k, device=device)

    def slice(self, samples, first_sample=0, stride=None, strict=False):
        """Decided bits of an int16 CUDA tensor: bit j = samples[first_sample + sample_delay + j*stride] >= 0
        (`strict`: > 0, the capture script software/memdump/decode.py:15).  Returns (packed int64 tensor, nbits)."""
        if samples.dtype != torch.int16 or not samples.is_cuda or not samples.is_contiguous():
            raise ValueError("samples must be a contiguous int16 CUDA tensor")
        stride = self.samples_per_bit if stride is None else int(stride)
        phase = int(first_sample) + self.sample_delay
        n = samples.numel()
        nbits = (n - phase + stride - 1) // stride if phase < n else 0
        out = torch.empty((nbits + 63) // 64, dtype=torch.int64, device=samples.device)
        nb = C.c_uint64()
        dev = samples.device.index or 0
        _lib.check(_lib.lib().bbb_rx_slice(C.c_void_p(samples.data_ptr()), n, stride, phase, int(bool(strict)),
                                           C.c_void_p(out.data_ptr()), C.byref(nb), dev,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "bbb_rx_slice")
        assert nb.value == nbits
        return out, nbits

    @staticmethod
    def decode_capture(raw, stride=4, device=0):
        """The reference's capture decoder (software/memdump/decode.py:11-18): `raw` holds little-endian
        int16 ADC samples (it reads 8192 of them from the serial port), every `stride`-th sample is
        compared against 0 (`dat > 0`) and the bits are returned as a uint8 numpy array."""
        import numpy as np
        x = np.frombuffer(bytes(raw), dtype="<i2").astype(np.int16)
        t = torch.from_numpy(x.copy()).to(torch.device("cuda", device))
        rx = RX(7, 1, 0, device=device)
        bits, nbits = rx.slice(t, stride=stride, strict=True)
        return np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:nbits]

    def count_errors(self, samples, first_sample=0, first_bit=0, prbs_init=1, stride=None):
        """Slice, then count the positions that differ from PRBS-k (started `first_bit` bits after `prbs_init`)."""
        bits, nbits = self.slice(samples, first_sample, stride)
        return self.prbsdet.count_errors(bits, nbits, first_bit=first_bit, init=prbs_init), nbits

    def detect(self, samples, first_sample=0, stride=None, want_err=False, want_reload=False):
        """Slice at this receiver's `sample_delay`, then the exact self-synchronising detector over the
        whole stream (rx.py:41-46 at scale): totals as PRBSErrorDetector.run_stream returns them."""
        bits, nbits = self.slice(samples, first_sample, stride)
        return self.prbsdet.run_stream(bits, nbits, want_err=want_err, want_reload=want_reload)

    def phase_search(self, samples, stride=None, strict=False):
        """Every setting of the reference's `sample_delay` knob (0 .. samples_per_bit - 1; rx.py:19): the
        detector's totals per phase and the phase with the fewest errors."""
        if samples.dtype != torch.int16 or not samples.is_cuda or not samples.is_contiguous():
            raise ValueError("samples must be a contiguous int16 CUDA tensor")
        stride = self.samples_per_bit if stride is None else int(stride)
        nph = self.samples_per_bit
        st = (_lib.DetectorStats * nph)()
        dev = samples.device.index or 0
        _lib.check(_lib.lib().bbb_rx_phase_search(C.c_void_p(samples.data_ptr()), samples.numel(), stride, nph,
                                                  int(bool(strict)), self.prbs_k, st, dev,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "bbb_rx_phase_search")
        out = [{n: int(getattr(s, n)) for n, _ in _lib.DetectorStats._fields_} for s in st]
        best = min(range(nph), key=lambda p: (out[p]["errors"] + out[p]["reload_clocks"], p))
        return out, best
