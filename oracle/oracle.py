"""ctypes front end of oracle/libbbb_oracle.so (the C restatement in bbb_oracle.c) plus a
tiny independent pure-Python restatement used to cross-check the C one.

TEST INFRASTRUCTURE ONLY -- see bbb_oracle.h for the rules and the parity status.
"""
import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SO = _HERE / "libbbb_oracle.so"
MAX_K = 512
WORDS = MAX_K // 64

PRBS_TAPS = {7: 6, 9: 5, 11: 9, 15: 14, 20: 3, 23: 18, 31: 28}   # gateware/bbb/prbs.py:14


def build(force=False, native=False, out=None):
    """Compile the oracle with gcc.  `native=True` builds a -march=native copy at `out`
    (used by bench.py on the GPU box's host for the timed CPU baseline)."""
    src = _HERE / "bbb_oracle.c"
    target = pathlib.Path(out) if out else _SO
    if (not force and target.exists()
            and target.stat().st_mtime >= max(src.stat().st_mtime, (_HERE / "bbb_oracle.h").stat().st_mtime)):
        return target
    flags = ["-O3", "-fPIC", "-std=c11", "-shared"]
    flags += ["-march=native"] if native else ["-mpopcnt", "-msse4.2"]
    subprocess.check_call(["gcc", *flags, "-o", str(target), str(src)])
    return target


class _Lutopt(C.Structure):
    _fields_ = [("k", C.c_int), ("ntaps", C.c_int * MAX_K), ("taps", (C.c_uint16 * 8) * MAX_K)]


class _Trial(C.Structure):
    _fields_ = [("prbs_k", C.c_int), ("prbs_state", C.c_uint64), ("amp", C.c_int),
                ("noise_var", C.c_int), ("warmup", C.c_uint64), ("first_bit", C.c_uint64),
                ("nbits", C.c_uint64)]


_lib = None


def lib(path=None):
    global _lib
    if path is not None:
        return _bind(C.CDLL(str(path)))
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(str(_SO)))
    return _lib


def _bind(l):
    u64p, u8p, i8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int8)
    LP = C.POINTER(_Lutopt)
    l.bbo_lutopt_load.argtypes = [LP, C.c_char_p]
    l.bbo_lutopt_from_packed.argtypes = [LP, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    l.bbo_lutopt_step.argtypes = [LP, u64p, u64p]
    l.bbo_lutopt_step.restype = None
    l.bbo_lutopt_run.argtypes = [LP, u64p, C.c_uint64, u64p]
    l.bbo_lutopt_run.restype = None
    l.bbo_clt_tree.argtypes = [u64p, C.c_int]
    l.bbo_lutopt_states.argtypes = [LP, u64p, C.c_uint64, C.c_uint64, u64p]
    l.bbo_lutopt_states.restype = None
    l.bbo_lutopt_words_u32.argtypes = [LP, u64p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
    l.bbo_clt_tree_bulk.argtypes = [u64p, C.c_int, C.c_uint64, C.POINTER(C.c_int16)]
    l.bbo_clt_tree_bulk.restype = None
    l.bbo_clt_popcount.argtypes = [u64p, C.c_int]
    l.bbo_clt_wrap.argtypes = [C.c_int, C.c_int]
    for f in (l.bbo_awgn_stream_i8, l.bbo_awgn_stream_i8_fast256):
        f.argtypes = [LP, u64p, C.c_uint64, C.c_uint64, i8p]
        f.restype = None
    l.bbo_prbs_tap.argtypes = [C.c_int]
    l.bbo_prbs_bits.argtypes = [C.c_int, u64p, C.c_uint64, u8p]
    l.bbo_prbs_packed.argtypes = [C.c_int, u64p, C.c_uint64, u64p]
    l.bbo_prbs_packed_fast.argtypes = [C.c_int, u64p, C.c_uint64, u64p]
    l.bbo_prbs_check_packed.argtypes = [C.c_int, u64p, C.c_uint64, u64p, u64p]
    l.bbo_prbs_detector_run.argtypes = [C.c_int, u8p, C.c_uint64, u8p, u8p]
    l.bbo_prbs_detector_packed.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    l.bbo_txrx_decide.argtypes = [C.c_int] * 4
    l.bbo_ber_trial.argtypes = [LP, u64p, C.POINTER(_Trial), u64p, u64p]
    l.bbo_rnghunt_recur.argtypes = [C.c_int, C.c_int, u64p, u8p, C.c_int, u8p]
    i16p = C.POINTER(C.c_int16)
    l.bbo_shaper_i16.argtypes = [i16p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, i16p]
    l.bbo_tx_i16.argtypes = [LP, u64p, i16p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64,
                             C.c_uint64, C.c_uint64, i16p]
    return l


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def int_to_words(v, k):
    """HDL integer (bit i = x[i], rng.py:135) -> ceil(k/64) little-endian u64 words (padded to WORDS)."""
    out = np.zeros(WORDS, dtype=np.uint64)
    for w in range((k + 63) // 64):
        out[w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def words_to_int(w, k):
    v = 0
    for i in range((k + 63) // 64):
        v |= int(w[i]) << (64 * i)
    return v & ((1 << k) - 1)


class Lutopt:
    """The reference's LUTOPT recurrence (gateware/bbb/rng.py:14-55), CPU oracle."""

    def __init__(self, path=None, packed=None, _lib=None):
        self._l = _lib or lib()
        self._m = _Lutopt()
        if path is not None and not str(path).endswith(".taps"):
            if self._l.bbo_lutopt_load(C.byref(self._m), str(path).encode()):      # reference 0/1 text format
                raise ValueError(f"cannot load matrix {path}")
        else:
            if path is not None:
                packed = [[int(x) for x in l.split()] for l in open(path) if l.strip()]
            flat = np.array([c for row in packed for c in row], dtype=np.uint16)
            off = np.zeros(len(packed) + 1, dtype=np.uint32)
            off[1:] = np.cumsum([len(r) for r in packed])
            if self._l.bbo_lutopt_from_packed(C.byref(self._m), len(packed),
                                              flat.ctypes.data_as(C.POINTER(C.c_uint16)),
                                              off.ctypes.data_as(C.POINTER(C.c_uint32))):
                raise ValueError("bad packed matrix")
        self.k = self._m.k

    @property
    def packed(self):
        return [[self._m.taps[r][j] for j in range(self._m.ntaps[r])] for r in range(self.k)]

    def step_int(self, x):
        a = int_to_words(x, self.k)
        b = np.zeros(WORDS, dtype=np.uint64)
        self._l.bbo_lutopt_step(C.byref(self._m), _u64(a), _u64(b))
        return words_to_int(b, self.k)

    def run_int(self, init, nsteps):
        a = int_to_words(init, self.k)
        b = np.zeros(WORDS, dtype=np.uint64)
        self._l.bbo_lutopt_run(C.byref(self._m), _u64(a), nsteps, _u64(b))
        return words_to_int(b, self.k)

    def states(self, init, first_step, nstates):
        """uint64[nstates, ceil(k/64)]: A^(first_step+1+i) init."""
        nw = (self.k + 63) // 64
        out = np.zeros((nstates, nw), dtype=np.uint64)
        self._l.bbo_lutopt_states(C.byref(self._m), _u64(int_to_words(init, self.k)), first_step, nstates, _u64(out))
        return out

    def words_u32(self, init, first_step, nstates, msb_first=False):
        """uint32[nstates * k/32]; msb_first = the dieharder dump of software/rnghunt/util/verify.py:46-52."""
        out = np.zeros(nstates * (self.k // 32), dtype=np.uint32)
        if self._l.bbo_lutopt_words_u32(C.byref(self._m), _u64(int_to_words(init, self.k)), first_step, nstates,
                                        int(msb_first), out.ctypes.data_as(C.POINTER(C.c_uint32))):
            raise ValueError("k must be a multiple of 32")
        return out

    def clt_tree_bulk(self, words):
        """Un-truncated tree values (int16) of uint64[nstates, k/64] caller-supplied words."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros(words.shape[0], dtype=np.int16)
        self._l.bbo_clt_tree_bulk(_u64(words), self.k, words.shape[0], out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out

    def clt_tree(self, x):
        return self._l.bbo_clt_tree(_u64(int_to_words(x, self.k)), self.k)

    def clt_popcount(self, x):
        return self._l.bbo_clt_popcount(_u64(int_to_words(x, self.k)), self.k)

    def clt_wrap(self, v):
        return self._l.bbo_clt_wrap(v, self.k)

    def awgn(self, init, first_step, nsamples, fast=False, out=None):
        if out is None:
            out = np.empty(nsamples, dtype=np.int8)
        assert out.dtype == np.int8 and out.size >= nsamples and out.flags.c_contiguous
        f = self._l.bbo_awgn_stream_i8_fast256 if fast else self._l.bbo_awgn_stream_i8
        f(C.byref(self._m), _u64(int_to_words(init, self.k)), first_step, nsamples,
          out.ctypes.data_as(C.POINTER(C.c_int8)))
        return out

    def ber_trial(self, init, prbs_k, prbs_state, amp, noise_var, warmup, first_bit, nbits):
        t = _Trial(prbs_k, prbs_state, amp, noise_var, warmup, first_bit, nbits)
        nb, ne = C.c_uint64(), C.c_uint64()
        if self._l.bbo_ber_trial(C.byref(self._m), _u64(int_to_words(init, self.k)), C.byref(t),
                                 C.byref(nb), C.byref(ne)):
            raise ValueError("k={} invalid for PRBS".format(prbs_k))
        return nb.value, ne.value


def prbs_bits(k, nbits, state=1, _lib=None):
    """(bits as uint8 array, state after) -- gateware/bbb/prbs.py:32-35."""
    l = _lib or lib()
    s = C.c_uint64(state)
    out = np.empty(nbits, dtype=np.uint8)
    if l.bbo_prbs_bits(k, C.byref(s), nbits, out.ctypes.data_as(C.POINTER(C.c_uint8))):
        raise ValueError("k={} invalid for PRBS".format(k))
    return out, s.value


def prbs_packed(k, nbits, state=1, fast=False, _lib=None):
    l = _lib or lib()
    s = C.c_uint64(state)
    out = np.zeros((nbits + 63) // 64, dtype=np.uint64)
    f = l.bbo_prbs_packed_fast if fast else l.bbo_prbs_packed
    if f(k, C.byref(s), nbits, _u64(out)):
        raise ValueError("k={} invalid for PRBS".format(k))
    return out, s.value


def prbs_check_packed(k, words, nbits, state=1, _lib=None):
    l = _lib or lib()
    s = C.c_uint64(state)
    ne = C.c_uint64()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if l.bbo_prbs_check_packed(k, C.byref(s), nbits, _u64(words), C.byref(ne)):
        raise ValueError("k={} invalid for PRBS".format(k))
    return ne.value


def prbs_detector_run(k, bits, _lib=None):
    """(err, reload) uint8 arrays -- gateware/bbb/prbs.py:61-99, cycle exact."""
    l = _lib or lib()
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    err = np.empty(len(bits), dtype=np.uint8)
    rl = np.empty(len(bits), dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    if l.bbo_prbs_detector_run(k, p(bits), len(bits), p(err), p(rl)):
        raise ValueError("k={} invalid for PRBS".format(k))
    return err, rl


def prbs_detector_packed(k, words, nbits, _lib=None):
    """(err_words, reload_words, stats) of the detector on a packed stream; stats = dict(errors, errors_raw,
    reload_clocks, resyncs)."""
    l = _lib or lib()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    nw = (nbits + 63) // 64
    assert len(words) >= nw
    err = np.zeros(nw, dtype=np.uint64)
    rl = np.zeros(nw, dtype=np.uint64)
    st = (C.c_uint64 * 4)()
    if l.bbo_prbs_detector_packed(k, _u64(words), nbits, _u64(err), _u64(rl), st):
        raise ValueError("k={} invalid for PRBS".format(k))
    return err, rl, dict(errors=st[0], errors_raw=st[1], reload_clocks=st[2], resyncs=st[3])


def shaper(coeffs, k, nsamples, first_sample=0, prbs_state=1, source=0):
    """PRBSShaper output samples (gateware/bbb/bitshaper.py:25-86) as int16."""
    c = np.ascontiguousarray(coeffs, dtype=np.int16)
    assert c.size == 64
    out = np.empty(nsamples, dtype=np.int16)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int16))  # noqa: E731
    if lib().bbo_shaper_i16(p(c), source, k, prbs_state, first_sample, nsamples, p(out)):
        raise ValueError("k={} invalid for PRBS".format(k))
    return out


def tx(lut, init, coeffs, k, nsamples, first_sample=0, prbs_state=1, source=0, bit_en=1, noise_en=1, noise_var=8,
       warmup=16):
    """TX.x samples (gateware/bbb/tx.py:60-81) as int16; `lut` is an oracle Lutopt."""
    c = np.ascontiguousarray(coeffs, dtype=np.int16)
    out = np.empty(nsamples, dtype=np.int16)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int16))  # noqa: E731
    if lib().bbo_tx_i16(C.byref(lut._m), _u64(int_to_words(init, lut.k)), p(c), source, k, prbs_state, bit_en, noise_en,
                        noise_var, warmup, first_sample, nsamples, p(out)):
        raise ValueError("k={} invalid for PRBS".format(k))
    return out


def rx_slice(samples, stride, phase, strict=False):
    """Decided bits (uint8 array) of an int16 sample array -- rx.py:29 / decode.py:15-16."""
    s = np.ascontiguousarray(samples, dtype=np.int16)
    out = np.empty(len(s) // max(1, stride) + 2, dtype=np.uint8)
    f = lib().bbo_rx_slice
    f.restype = C.c_uint64
    f.argtypes = [C.POINTER(C.c_int16), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint8)]
    n = f(s.ctypes.data_as(C.POINTER(C.c_int16)), len(s), stride, phase, int(strict), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:n]


def txrx_decide(g, bit, amp, noise_var):
    return lib().bbo_txrx_decide(int(g), int(bit), int(amp), int(noise_var))


def rnghunt_recur(nrows, ncols, col_words, x_bits, n):
    cw = np.array(col_words, dtype=np.uint64)
    xb = np.array(x_bits, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    if lib().bbo_rnghunt_recur(nrows, ncols, _u64(cw), p(xb), n, p(out)):
        raise ValueError("bad shape")
    return out


# ---- tiny independent pure-Python restatement (small cases only) ------------------

def py_lutopt_step(packed, x):
    """rng.py:38-40 with Python ints."""
    y = 0
    for r, taps in enumerate(packed):
        b = 0
        for c in taps:
            b ^= (x >> c) & 1
        y |= b << r
    return y


def py_clt_tree(x, n):
    """clt-grng-evaluate.py:10-15 on the LSB-first bits of x."""
    v = [(x >> i) & 1 for i in range(n)]
    while len(v) > 1:
        v = [v[2 * j] - v[2 * j + 1] for j in range(len(v) // 2)]
    return v[0]


def py_prbs(k, nbits, lfsr=1):
    """prbs.py:112-113."""
    out = []
    for _ in range(nbits):
        bit = ((lfsr >> (k - 1)) ^ (lfsr >> PRBS_TAPS[k] - 1)) & 1
        lfsr = ((lfsr << 1) | bit) & ((1 << k) - 1)
        out.append(bit)
    return out, lfsr


def data_path(n):
    """Matrix files live with the product package (pure data, shared)."""
    return _HERE.parent / "basebandboard_amd" / "data" / f"lutopt_{n}.taps"
