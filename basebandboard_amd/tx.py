"""Transmitter output stream -- host-side mirror of gateware/bbb/tx.py.

`TX(prbs_k, bit_en, src_sel, shape_sel, noise_en, noise_var)` takes the reference's arguments
(tx.py:39-52).  Its output is the shaped data bits (PRBS-k or the pulse source) plus the CLT noise
stream scaled by `noise_var`, each gated by its enable, as 12-bit signed samples at 8 per data bit.
`generate` fills a tensor with that stream on the GPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .bitshaper import PRBSShaper, Pulser, _cfg
from .prbs import PRBS
from .rng import CLTGRNG, LUTOPT


class TX:
    def __init__(self, prbs_k, bit_en, src_sel, shape_sel, noise_en, noise_var, device=0, init=1, prbs_init=1):
        self.betas = np.linspace(0, 1, 32).tolist()                      # tx.py:54
        self.prbs = PRBS(prbs_k, init=prbs_init, device=device)           # tx.py:55 (raises ValueError on a bad k)
        self.prbs_shaper = PRBSShaper.from_rcf(self.prbs, shape_sel, self.betas)
        self.pulse = Pulser()
        self.pulse_shaper = PRBSShaper.from_rcf(self.pulse, shape_sel, self.betas)
        self.urng = LUTOPT.shipped(256, init=init, device=device)         # tx.py:70: the n256 recurrence
        self.grng = CLTGRNG(self.urng)
        if not 0 <= int(noise_var) <= 15:
            raise ValueError("noise_var is a 4-bit unsigned value")      # tx.py:52
        self.bit_en, self.src_sel, self.noise_en, self.noise_var = bool(bit_en), int(src_sel), bool(noise_en), int(noise_var)
        self.device = int(device)

    def generate(self, nsamples, first_sample=0, warmup=16, out=None, stream_on=True):
        """Samples first_sample .. first_sample + nsamples - 1 of `x` (int16, 12-bit signed).
        stream_on: announce that the next call continues where this one ends (bbb_awgn_prefetch), so that the
        noise generator's start states for it are derived beside this call's kernels; a call that does not
        continue there simply ignores the hint."""
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty(int(nsamples), dtype=torch.int16, device=dev)
        if out.dtype != torch.int16 or out.numel() < nsamples or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous int16 tensor on {dev} with >= nsamples elements")
        shaper = self.pulse_shaper if self.src_sel else self.prbs_shaper            # tx.py:65
        cfg = _cfg(shaper.coefficients[shaper.setsel], shaper.prbs, self.bit_en, self.noise_en, self.noise_var, warmup)
        self.urng._bind_stream()
        _lib.check(_lib.lib().bbb_tx_fill_i16(self.urng._h, C.byref(cfg), C.c_void_p(out.data_ptr()), int(nsamples),
                                              int(first_sample)), "bbb_tx_fill_i16")
        if stream_on and self.noise_en:
            _lib.check(_lib.lib().bbb_awgn_prefetch(self.urng._h, int(nsamples), int(warmup) + int(first_sample) + int(nsamples)),
                       "bbb_awgn_prefetch")
        return out[:nsamples]

    def stream(self, nsamples_per_call, first_sample=0, warmup=16):
        """TX.x read sequentially (bbb_tx_stream_*): `with tx.stream(n) as s: s.next(out=buf)`."""
        return WaveformStream(self, nsamples_per_call, first_sample, warmup)


class WaveformStream:
    """bbb_tx_stream_*: the transmitter's waveform read sequentially in equal calls; the library turns the two-kernel
    form on and announces every next call.  The TX object's settings are copied when the stream is opened.  Context
    manager; closing restores the generator handle's mode."""

    def __init__(self, tx, nsamples_per_call, first_sample=0, warmup=16):
        self.tx, self.n = tx, int(nsamples_per_call)
        shaper = tx.pulse_shaper if tx.src_sel else tx.prbs_shaper            # tx.py:65
        cfg = _cfg(shaper.coefficients[shaper.setsel], shaper.prbs, tx.bit_en, tx.noise_en, tx.noise_var, warmup)
        s = C.c_void_p()
        tx.urng._bind_stream()
        _lib.check(_lib.lib().bbb_tx_stream_open(tx.urng._h, C.byref(cfg), self.n, int(first_sample), C.byref(s)), "bbb_tx_stream_open")
        self._s = s

    def _out(self, n, out):
        dev = torch.device("cuda", self.tx.device)
        if out is None:
            out = torch.empty(n, dtype=torch.int16, device=dev)
        if out.dtype != torch.int16 or out.numel() < n or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous int16 tensor on {dev} with >= {n} elements")
        self.tx.urng._bind_stream()
        return out

    def next(self, out=None):
        out = self._out(self.n, out)
        _lib.check(_lib.lib().bbb_tx_stream_next(self._s, C.c_void_p(out.data_ptr())), "bbb_tx_stream_next")
        return out[:self.n]

    def read(self, nsamples, out=None):
        n = int(nsamples)
        out = self._out(n, out)
        _lib.check(_lib.lib().bbb_tx_stream_read(self._s, C.c_void_p(out.data_ptr()), n), "bbb_tx_stream_read")
        return out[:n]

    def seek(self, first_sample):
        _lib.check(_lib.lib().bbb_tx_stream_seek(self._s, int(first_sample)), "bbb_tx_stream_seek")

    def tell(self):
        v = C.c_uint64()
        _lib.check(_lib.lib().bbb_tx_stream_tell(self._s, C.byref(v)), "bbb_tx_stream_tell")
        return v.value

    def close(self):
        s, self._s = getattr(self, "_s", None), None
        if s:
            _lib.check(_lib.lib().bbb_tx_stream_close(s), "bbb_tx_stream_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
