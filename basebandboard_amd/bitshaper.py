"""Shape a bitstream into oversampled pulses -- host-side mirror of gateware/bbb/bitshaper.py.

`PRBSShaper(prbs, setsel, coefficients)` keeps the reference's constructor (bitshaper.py:25) and its
`from_rcf` class method (:88-109); `generate` returns the samples the reference's `x` shows on
successive clocks (8 per data bit, 12-bit signed in int16), computed on the GPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .prbs import PRBS


class Pulser:
    """Alternative data source of TX: a single 1 every 256 bit periods (mirror of tx.py:20-30)."""
    k = 0
    init = 1


def _cfg(coeffs, source, bit_en=1, noise_en=0, noise_var=0, warmup=0):
    c = _lib.TxCfg()
    if len(coeffs) != 64:
        raise ValueError("a coefficient set has 64 entries (8 per bit duration, bitshaper.py:19-21)")
    for i, v in enumerate(coeffs):
        if not -256 < int(v) <= 255:
            raise ValueError("coefficients must be integers in (-256, 255)")
        c.coeffs[i] = int(v)
    if isinstance(source, PRBS):
        c.source, c.prbs_k, c.prbs_state = 0, source.k, source.init
    elif isinstance(source, Pulser) or source is Pulser:
        c.source, c.prbs_k, c.prbs_state = 1, 0, 1
    else:
        raise ValueError("source must be a PRBS or a Pulser")
    c.bit_en, c.noise_en, c.noise_var, c.warmup = int(bool(bit_en)), int(bool(noise_en)), int(noise_var), int(warmup)
    return c


def rcf_coefficients(beta):
    """The 64 fixed-point taps of a raised-cosine pulse with roll-off beta, T = 8 samples per bit, peak
    254 -- the table PRBSShaper.from_rcf stores (bitshaper.py:97-107): sinc(t/T) cos(pi beta t/T) /
    (1 - (2 beta t/T)^2), the two singular taps replaced by their limit, truncated towards zero."""
    T = 8
    t = np.arange(-32, 32)
    singular = np.where(np.abs(t) == T / (2 * beta)) if beta != 0.0 else None
    if singular is not None:
        t[singular] = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        c = 1 / T * np.sinc(t / T) * np.cos(np.pi * beta * t / T) / (1 - (2 * beta * t / T) ** 2)
    if singular is not None:
        c[singular] = np.pi / (4 * T) * np.sinc(1 / (2 * beta))
    return (c * T * 254).astype(np.int64).tolist()


class PRBSShaper:
    """Pulse shaper at 8 samples per data bit: every bit launches a 64-tap pulse (sign by the bit
    value) and the output is the superposition of the 8 pulses in flight (mirror of
    bitshaper.py:12-86).  `prbs`: the bit source (PRBS or Pulser); `setsel`: index into
    `coefficients`, a list of tap sets of 64 integers each, all within (-256, 255)."""

    def __init__(self, prbs, setsel, coefficients):
        if not 1 <= len(coefficients) <= 33:
            raise ValueError("between 1 and 33 coefficient sets")
        self.prbs, self.setsel = prbs, int(setsel)
        self.coefficients = [list(map(int, c)) for c in coefficients]
        if not 0 <= self.setsel < len(self.coefficients):
            raise ValueError("setsel out of range")

    @classmethod
    def from_rcf(cls, prbs, setsel, betas):
        """One raised-cosine tap set per roll-off in `betas`; when that leaves room (< 32 sets) a
        4-sample rectangular pulse is added as the last set (bitshaper.py:88-109)."""
        cc = [rcf_coefficients(b) for b in betas]
        if len(cc) < 32:
            cc.append([0] * 30 + [254] * 4 + [0] * 30)
        return cls(prbs, setsel, cc)

    def generate(self, nsamples, first_sample=0, out=None):
        device = getattr(self.prbs, "device", 0)
        dev = torch.device("cuda", device)
        if out is None:
            out = torch.empty(int(nsamples), dtype=torch.int16, device=dev)
        if out.dtype != torch.int16 or out.numel() < nsamples or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous int16 tensor on {dev} with >= nsamples elements")
        cfg = _cfg(self.coefficients[self.setsel], self.prbs)
        _lib.check(_lib.lib().bbb_shaper_fill_i16(C.byref(cfg), C.c_void_p(out.data_ptr()), int(nsamples), int(first_sample),
                                                  device, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   "bbb_shaper_fill_i16")
        return out[:nsamples]
