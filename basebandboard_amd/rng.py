"""Uniform and Gaussian random number generators -- host-side mirror of gateware/bbb/rng.py.

`LUTOPT` and `CLTGRNG` keep the reference's constructor surface (rng.py:21-55, 70-78).  Where the
reference's HDL emits one value per clock, these objects emit whole streams into HBM: sample i of
`CLTGRNG.generate` is the value the reference's `grng.x` shows for LUTOPT state number
first_step + i + 1 (its log2(n)-clock pipeline delay, rng.py:67-68, removed).  All computation
happens in libbbb_hip.so on the GPU; torch only owns the device memory.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, recurrences


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _int_to_words(v, k):
    n = (k + 63) // 64
    return (C.c_uint64 * n)(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)])


class LUTOPT:
    """Wide uniform word source: a GF(2) linear recurrence x' = A x with a sparse A (mirror of rng.py:14-40).

    `a` is the k x k 0/1 recurrence matrix, `init` the initial state as an integer whose bit i is
    state element i (rng.py:30,135).  `x` after n clocks is `state_at(n)`.
    """

    def __init__(self, a, init=1, device=0):
        a = np.asarray(a)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("recurrence matrix must be square")
        self._setup([np.nonzero(row)[0].tolist() for row in a], init, device)   # rng.py:38-39

    @classmethod
    def from_packed(cls, packed, init=1, device=0):
        """From a list of k lists holding the positions of the 1 entries of each row (rng.py:42-55)."""
        self = cls.__new__(cls)
        self._setup([list(r) for r in packed], init, device)
        return self

    @classmethod
    def from_matrix_file(cls, path, init=1, device=0):
        """From a software/rnghunt/matrices/N style text file, parsed by the library."""
        l = _lib.lib()
        k = C.c_int()
        taps = C.POINTER(C.c_uint16)()
        off = C.POINTER(C.c_uint32)()
        _lib.check(l.bbb_lutopt_load_matrix_file(str(path).encode(), C.byref(k), C.byref(taps), C.byref(off)),
                   "bbb_lutopt_load_matrix_file")
        try:
            packed = [[taps[j] for j in range(off[r], off[r + 1])] for r in range(k.value)]
        finally:
            l.bbb_free(taps)
            l.bbb_free(off)
        return cls.from_packed(packed, init, device)

    @classmethod
    def shipped(cls, n=256, init=1, device=0):
        """The maximum-period matrix the reference uses for width n (rng_recurrences.py)."""
        return cls.from_packed(recurrences.load_packed(recurrences.matrix_path(n)), init, device)

    def _setup(self, packed, init, device):
        l = _lib.lib()
        self.k = len(packed)
        self.packed = packed
        self.init = int(init)
        self.device = int(device)
        flat = (C.c_uint16 * max(1, sum(len(r) for r in packed)))(*[c for r in packed for c in r])
        offs = [0]
        for r in packed:
            offs.append(offs[-1] + len(r))
        off = (C.c_uint32 * len(offs))(*offs)
        if self.init < 0 or self.init >> self.k:
            raise ValueError("init does not fit in k bits")
        h = C.c_void_p()
        _lib.check(l.bbb_lutopt_create(C.byref(h), self.k, flat, off, _int_to_words(self.init, self.k), self.device),
                   "bbb_lutopt_create")
        self._h = h

    @property
    def a(self):
        a = np.zeros((self.k, self.k), dtype=np.uint8)
        for r, taps in enumerate(self.packed):
            a[r, taps] = 1
        return a

    @property
    def specialised(self):
        return bool(_lib.lib().bbb_lutopt_is_specialised(self._h))

    def specialise(self, build_dir=None, hipcc="/opt/rocm/bin/hipcc"):
        """Build and attach a sample kernel for THIS matrix (for matrices that are not among the shipped ones,
        e.g. a result of gf2.search): basebandboard_amd/gen_lutopt_kernel.py emits the straight-line network, hipcc
        compiles csrc/custom_fill_template.hip around it for gfx950, bbb_lutopt_attach_custom_library attaches the
        result: the sample kernel and, for k = 256, the fused BER trial kernels.
        Cached under `build_dir` (default ~/.cache/basebandboard_amd) by the hash of the tap lists.  Needs hipcc
        on this machine; power-of-two k <= 256.  Returns the path of the library."""
        import hashlib
        import pathlib
        import subprocess
        import sys
        if self.k > 256 or self.k & (self.k - 1):
            raise ValueError("custom kernels exist for power-of-two k <= 256")
        root = pathlib.Path(__file__).resolve().parent
        # the cache key covers the taps AND everything the built library shares with libbbb_hip.so (template, BER kernels,
        # launch header with BBB_CUSTOM_ABI, generator): a library cached by another checkout is never picked up
        hsh = hashlib.sha1(repr([list(r) for r in self.packed]).encode())
        for dep in ("csrc/custom_fill_template.hip", "csrc/ber_kernels_impl.hpp", "csrc/awgn_launch.hpp", "csrc/bitslice_util.hpp",
                    "csrc/bbb_common.hpp", "csrc/custom_abi.hpp", "gen_lutopt_kernel.py"):
            hsh.update((root / dep).read_bytes())
        key = hsh.hexdigest()[:16]
        bdir = pathlib.Path(build_dir) if build_dir else pathlib.Path.home() / ".cache" / "basebandboard_amd"
        work = bdir / f"k{self.k}_{key}"
        so = work / f"libbbb_custom_{key}.so"
        if not so.exists():
            work.mkdir(parents=True, exist_ok=True)
            taps = work / f"lutopt_{self.k}.taps"
            taps.write_text("".join(" ".join(map(str, r)) + "\n" for r in self.packed))
            subprocess.check_call([sys.executable, str(root / "gen_lutopt_kernel.py"), str(taps), str(work / "custom_gen.inc")])
            tmp = work / (so.name + ".tmp")
            subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", f"-DBBB_N={self.k}",
                                   f"-DBBB_LOG={self.k.bit_length() - 1}", f"-I{work}", f"-I{root / 'csrc'}", "-Wno-unused-function",
                                   str(root / "csrc" / "custom_fill_template.hip"), "-o", str(tmp)])
            tmp.replace(so)
        _lib.check(_lib.lib().bbb_lutopt_attach_custom_library(self._h, str(so).encode()), "bbb_lutopt_attach_custom_library")
        self._custom = so
        return so

    def set_staged(self, enable=True, look_ahead=False):
        """Two-kernel form of the large k = 256 fills of this handle (bbb_lutopt_set_staged): same samples; the output
        is written as full lines by a second kernel that overlaps the next fill's arithmetic.  look_ahead = m (True: 2):
        a fill's sample kernel also produces the next m - 1 fills' samples (for consumers that read the stream
        sequentially in equal fills); 2 <= m <= 8."""
        level = 0 if not enable else (1 if not look_ahead else (2 if look_ahead is True else int(look_ahead)))
        _lib.check(_lib.lib().bbb_lutopt_set_staged(self._h, level), "bbb_lutopt_set_staged")

    def state_at(self, nsteps):
        """Integer value of `x` after nsteps clocks from reset."""
        out = (C.c_uint64 * ((self.k + 63) // 64))()
        _lib.check(_lib.lib().bbb_lutopt_state_at(self._h, nsteps, out), "bbb_lutopt_state_at")
        return sum(int(w) << (64 * i) for i, w in enumerate(out))

    def generate_words(self, nstates, first_step=0, msb_first=False, out=None):
        """`x` in bulk: the states after first_step + 1 ... first_step + nstates clocks as k/32 consecutive 32-bit
        words each (int32 CUDA tensor of nstates * k/32 elements holding the bit patterns).  msb_first=True gives
        the words of the dieharder dump the reference writes (software/rnghunt/util/verify.py:37-52)."""
        if self.k % 32:
            raise ValueError("the word stream needs k to be a multiple of 32")
        dev = torch.device("cuda", self.device)
        n = int(nstates) * (self.k // 32)
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=dev)
        if out.dtype != torch.int32 or out.numel() < n or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous int32 tensor on {dev} with >= nstates * k/32 elements")
        self._bind_stream()
        _lib.check(_lib.lib().bbb_lutopt_fill_words(self._h, C.c_void_p(out.data_ptr()), int(nstates), int(first_step),
                                                    int(bool(msb_first))), "bbb_lutopt_fill_words")
        return out[:n]

    def profile(self, enable=True):
        """Time the generator kernels on the device (hipEvents on the launch stream)."""
        _lib.check(_lib.lib().bbb_lutopt_profile(self._h, int(enable)), "bbb_lutopt_profile")

    def profile_read(self, reset=True):
        """(seed_ms, kernel_ms, calls) accumulated since the last reset; waits for the events."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint64()
        _lib.check(_lib.lib().bbb_lutopt_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n), int(reset)),
                   "bbb_lutopt_profile_read")
        return a.value, b.value, n.value

    def profile_read_mover(self, reset=True):
        """(mover_ms, movers): the second kernel of the two-kernel form, timed on its own stream."""
        a, n = C.c_double(), C.c_uint64()
        _lib.check(_lib.lib().bbb_lutopt_profile_read_mover(self._h, C.byref(a), C.byref(n), int(reset)),
                   "bbb_lutopt_profile_read_mover")
        return a.value, n.value

    def _bind_stream(self):
        _lib.check(_lib.lib().bbb_lutopt_set_stream(self._h, _stream_ptr(self.device)), "bbb_lutopt_set_stream")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().bbb_lutopt_destroy(h)
            except Exception:
                pass
            self._h = None


class CLTGRNG:
    """Approximately Gaussian integers: a +-1 weighted sum of the n bits of a uniform word, formed by
    a log2(n)-level subtractor tree (mirror of rng.py:58-108).  Zero mean, variance n/4, signed
    log2(n)-bit output (rng.py:63-66,78)."""

    def __init__(self, urng):
        n = urng.k
        if n & (n - 1):
            raise ValueError("urng width must be a power of two")     # rng.py:72-76
        self.urng = urng
        self.n = n
        self.dtype = torch.int8 if n <= 256 else torch.int16

    @property
    def variance(self):
        return 2.0 ** (int(np.log2(self.n)) - 2)

    def generate(self, nsamples, first_step=0, out=None):
        """nsamples consecutive samples, starting with the one for LUTOPT state first_step + 1."""
        u = self.urng
        dev = torch.device("cuda", u.device)
        if out is None:
            out = torch.empty(int(nsamples), dtype=self.dtype, device=dev)
        if out.dtype != self.dtype or out.numel() < nsamples or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous {self.dtype} tensor on {dev} with >= nsamples elements")
        u._bind_stream()
        fn = _lib.lib().bbb_awgn_fill_i8 if self.dtype == torch.int8 else _lib.lib().bbb_awgn_fill_i16
        _lib.check(fn(u._h, C.c_void_p(out.data_ptr()), int(nsamples), int(first_step)), "bbb_awgn_fill")
        return out[:nsamples]

    def prefetch(self, nsamples, first_step=0):
        """Announce the next `generate(nsamples, first_step)`: its start states are derived now, on a
        side stream, beside whatever the GPU is running.  Purely a performance hint."""
        _lib.check(_lib.lib().bbb_awgn_prefetch(self.urng._h, int(nsamples), int(first_step)), "bbb_awgn_prefetch")

    def stream(self, nsamples_per_call, first_step=0):
        """The generator as what it is in the reference: ONE sequential stream, drained in order (rng.py:70-108 emits
        one value per clock).  Returns a `SampleStream` (bbb_awgn_stream_*): `next()` delivers the following
        nsamples_per_call samples; staging, the two-kernel form and the announcement of every next read are the
        library's business."""
        return SampleStream(self, nsamples_per_call, first_step)

    @staticmethod
    def tree(states, n):
        """Adder-tree value (un-truncated) of caller-supplied uniform words: `states` is an int64 CUDA
        tensor [nstates, n/64] holding the bit patterns (software/clt-grng/clt-grng-evaluate.py:8-16)."""
        if states.dtype != torch.int64 or states.dim() != 2 or states.shape[1] * 64 != n or not states.is_cuda:
            raise ValueError("states must be an int64 CUDA tensor of shape [nstates, n/64]")
        states = states.contiguous()
        out = torch.empty(states.shape[0], dtype=torch.int16, device=states.device)
        dev = states.device.index or 0
        _lib.check(_lib.lib().bbb_clt_tree_i16(n, C.c_void_p(states.data_ptr()), states.shape[0],
                                               C.c_void_p(out.data_ptr()), dev, _stream_ptr(dev)), "bbb_clt_tree_i16")
        return out


class SampleStream:
    """bbb_awgn_stream_*: the CLTGRNG sample stream read sequentially.  Context manager; closing restores the
    generator handle's mode."""

    def __init__(self, grng, nsamples_per_call, first_step=0):
        self.grng = grng
        self.n = int(nsamples_per_call)
        u = grng.urng
        s = C.c_void_p()
        u._bind_stream()
        _lib.check(_lib.lib().bbb_awgn_stream_open(u._h, self.n, int(first_step), 1 if grng.dtype == torch.int8 else 2, C.byref(s)),
                   "bbb_awgn_stream_open")
        self._s = s

    def _out(self, n, out):
        u = self.grng.urng
        dev = torch.device("cuda", u.device)
        if out is None:
            out = torch.empty(n, dtype=self.grng.dtype, device=dev)
        if out.dtype != self.grng.dtype or out.numel() < n or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous {self.grng.dtype} tensor on {dev} with >= {n} elements")
        u._bind_stream()
        return out

    def next(self, out=None):
        """The next nsamples_per_call samples (asynchronous on the current torch stream)."""
        out = self._out(self.n, out)
        _lib.check(_lib.lib().bbb_awgn_stream_next(self._s, C.c_void_p(out.data_ptr())), "bbb_awgn_stream_next")
        return out[:self.n]

    def read(self, nsamples, out=None):
        """The next `nsamples` samples, any length; the stream continues behind them."""
        n = int(nsamples)
        out = self._out(n, out)
        _lib.check(_lib.lib().bbb_awgn_stream_read(self._s, C.c_void_p(out.data_ptr()), n), "bbb_awgn_stream_read")
        return out[:n]

    def seek(self, first_step):
        _lib.check(_lib.lib().bbb_awgn_stream_seek(self._s, int(first_step)), "bbb_awgn_stream_seek")

    def tell(self):
        v = C.c_uint64()
        _lib.check(_lib.lib().bbb_awgn_stream_tell(self._s, C.byref(v)), "bbb_awgn_stream_tell")
        return v.value

    def close(self):
        s, self._s = getattr(self, "_s", None), None
        if s:
            _lib.check(_lib.lib().bbb_awgn_stream_close(s), "bbb_awgn_stream_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
