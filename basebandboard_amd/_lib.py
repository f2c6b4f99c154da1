"""ctypes binding of libbbb_hip.so (C ABI: include/bbb.h).

The library is the product: if it is missing or cannot be loaded this module raises --
there is no Python/NumPy/CPU fallback for any of the compute entry points.
"""
import ctypes as C
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libbbb_hip.so"

SHARD_TRIALS, SHARD_SEEDS, SHARD_BITS, SHARD_GROUPS = 0, 1, 2, 3
BBB_OK, BBB_EINVAL, BBB_ENOMEM, BBB_EHIP, BBB_EIO, BBB_ENODEV, BBB_EUNSUP = 0, -1, -2, -3, -4, -5, -6

# every symbol include/bbb.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "bbb_abi_version", "bbb_strerror", "bbb_last_error_detail", "bbb_device_count", "bbb_free",
    "bbb_lutopt_load_matrix_file", "bbb_lutopt_create", "bbb_lutopt_destroy", "bbb_lutopt_set_stream",
    "bbb_lutopt_is_specialised", "bbb_lutopt_set_staged", "bbb_lutopt_set_custom_fill", "bbb_lutopt_set_custom_ber", "bbb_lutopt_attach_custom_library", "bbb_lutopt_profile", "bbb_lutopt_profile_read", "bbb_lutopt_profile_read_mover", "bbb_lutopt_state_at", "bbb_lutopt_fill_words", "bbb_awgn_fill_i8", "bbb_awgn_fill_i16", "bbb_awgn_prefetch", "bbb_awgn_stream_open", "bbb_awgn_stream_next", "bbb_awgn_stream_read", "bbb_awgn_stream_seek", "bbb_awgn_stream_tell", "bbb_awgn_stream_close",
    "bbb_clt_tree_i16", "bbb_prbs_fill", "bbb_prbs_fill_hint", "bbb_prbs_check", "bbb_prbs_check_dev", "bbb_prbs_state_at",
    "bbb_prbs_detector_run", "bbb_prbs_detector_stream", "bbb_ber_trials", "bbb_ber_trials_dev", "bbb_ber_run_open", "bbb_ber_run_next", "bbb_ber_run_next_dev", "bbb_ber_run_tell", "bbb_ber_run_close", "bbb_ber_sweep_multi", "bbb_sweep_shard", "bbb_multi_last_info", "bbb_multi_release", "bbb_shaper_fill_i16", "bbb_tx_fill_i16", "bbb_tx_stream_open", "bbb_tx_stream_next", "bbb_tx_stream_read", "bbb_tx_stream_seek", "bbb_tx_stream_tell", "bbb_tx_stream_close", "bbb_rx_slice", "bbb_rx_phase_search", "bbb_gf2_berlekamp_massey", "bbb_gf2_recur",
    "bbb_gf2_dot", "bbb_gf2_poly_is_primitive", "bbb_gf2_poly_modexp", "bbb_lutopt_charpoly", "bbb_lutopt_is_full_period",
    "bbb_lutopt_save_matrix_file", "bbb_lutopt_search_candidate", "bbb_lutopt_search",
]


class BbbError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__(f"{what}: {detail}" if detail else what)
        self.code = code


class TrialCfg(C.Structure):
    """bbb_trial_cfg"""
    _fields_ = [("prbs_k", C.c_int32), ("amp", C.c_int32), ("noise_var", C.c_int32), ("reserved", C.c_int32),
                ("prbs_state", C.c_uint64), ("warmup", C.c_uint64), ("first_bit", C.c_uint64),
                ("nbits", C.c_uint64)]


class TxCfg(C.Structure):
    """bbb_tx_cfg"""
    _fields_ = [("coeffs", C.c_int16 * 64), ("source", C.c_int32), ("prbs_k", C.c_int32), ("prbs_state", C.c_uint64),
                ("bit_en", C.c_int32), ("noise_en", C.c_int32), ("noise_var", C.c_int32), ("reserved", C.c_int32),
                ("warmup", C.c_uint64)]


class DetectorStats(C.Structure):
    """bbb_detector_stats"""
    _fields_ = [(n, C.c_uint64) for n in ("bits", "errors", "errors_raw", "reload_clocks", "resyncs", "chunks",
                                          "chunks_rerun", "serial_fallback")]


class SearchStats(C.Structure):
    """bbb_search_stats"""
    _fields_ = [(n, C.c_uint64) for n in ("tested", "full_degree", "order_divides", "primitive", "kernel_ns")]


class MultiInfo(C.Structure):
    """bbb_multi_info"""
    _fields_ = [("n_devices", C.c_int32), ("n_ranks_seen", C.c_int32), ("rccl_reused", C.c_int32), ("reserved", C.c_int32),
                ("rccl_path", C.c_char * 256)]


class Ber(C.Structure):
    """bbb_ber"""
    _fields_ = [("bits", C.c_uint64), ("errors", C.c_uint64)]


_lib = None


def select_build(name):
    """Choose which build of the library this process loads; must be called before the first use.
    "product" (default) = libbbb_hip.so.  "experiments" = libbbb_hip_exp.so, the same sources compiled with
    -DBBB_EXPERIMENTS, in which BBB_* environment variables select kernel variants for A/B timing; the product
    build has no such switches."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = _HERE / {"product": "libbbb_hip.so", "experiments": "libbbb_hip_exp.so"}[name]


def lib():
    """Load libbbb_hip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  basebandboard_amd has no CPU fallback.")
    l = C.CDLL(str(LIB_PATH))
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    u64p = C.POINTER(C.c_uint64)
    l.bbb_abi_version.restype = i32
    l.bbb_strerror.restype = C.c_char_p
    l.bbb_strerror.argtypes = [i32]
    l.bbb_last_error_detail.restype = C.c_char_p
    l.bbb_device_count.argtypes = [C.POINTER(i32)]
    l.bbb_free.argtypes = [vp]
    l.bbb_free.restype = None
    l.bbb_lutopt_load_matrix_file.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(C.POINTER(C.c_uint16)),
                                              C.POINTER(C.POINTER(C.c_uint32))]
    l.bbb_lutopt_create.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), u64p, i32]
    l.bbb_lutopt_destroy.argtypes = [vp]
    l.bbb_lutopt_set_stream.argtypes = [vp, vp]
    l.bbb_lutopt_is_specialised.argtypes = [vp]
    l.bbb_lutopt_set_staged.argtypes = [vp, i32]
    l.bbb_lutopt_state_at.argtypes = [vp, u64, u64p]
    l.bbb_lutopt_profile.argtypes = [vp, i32]
    l.bbb_lutopt_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p, i32]
    l.bbb_lutopt_profile_read_mover.argtypes = [vp, C.POINTER(C.c_double), u64p, i32]
    l.bbb_lutopt_fill_words.argtypes = [vp, vp, u64, u64, i32]
    l.bbb_awgn_fill_i8.argtypes = [vp, vp, u64, u64]
    l.bbb_awgn_fill_i16.argtypes = [vp, vp, u64, u64]
    l.bbb_awgn_prefetch.argtypes = [vp, u64, u64]
    l.bbb_awgn_stream_open.argtypes = [vp, u64, u64, i32, C.POINTER(vp)]
    l.bbb_awgn_stream_next.argtypes = [vp, vp]
    l.bbb_awgn_stream_read.argtypes = [vp, vp, u64]
    l.bbb_awgn_stream_seek.argtypes = [vp, u64]
    l.bbb_awgn_stream_tell.argtypes = [vp, u64p]
    l.bbb_awgn_stream_close.argtypes = [vp]
    l.bbb_clt_tree_i16.argtypes = [i32, vp, u64, vp, i32, vp]
    l.bbb_prbs_fill.argtypes = [i32, u64, u64, u64, vp, i32, vp]
    l.bbb_prbs_fill_hint.argtypes = [i32, u64, u64, u64, vp, C.c_uint, i32, vp]
    l.bbb_prbs_check.argtypes = [i32, u64, u64, u64, vp, u64p, i32, vp]
    l.bbb_prbs_check_dev.argtypes = [i32, u64, u64, u64, vp, vp, i32, vp]
    l.bbb_prbs_state_at.argtypes = [i32, u64, u64, u64p]
    l.bbb_prbs_detector_run.argtypes = [i32, vp, u64, u64, vp, vp, i32, vp]
    l.bbb_rx_phase_search.argtypes = [vp, u64, u64, u64, i32, i32, C.POINTER(DetectorStats), i32, vp]
    l.bbb_prbs_detector_stream.argtypes = [i32, vp, u64, vp, vp, C.POINTER(DetectorStats), u64, u64, i32, vp]
    l.bbb_ber_trials.argtypes = [vp, C.POINTER(TrialCfg), i32, C.POINTER(Ber)]
    l.bbb_ber_trials_dev.argtypes = [vp, C.POINTER(TrialCfg), i32, vp]
    l.bbb_multi_last_info.argtypes = [C.POINTER(MultiInfo)]
    l.bbb_ber_run_open.argtypes = [vp, C.POINTER(TrialCfg), i32, C.c_uint32, C.POINTER(vp)]
    l.bbb_ber_run_next.argtypes = [vp, C.POINTER(Ber)]
    l.bbb_ber_run_next_dev.argtypes = [vp, vp]
    l.bbb_ber_run_tell.argtypes = [vp, u64p, u64p]
    l.bbb_ber_run_close.argtypes = [vp]
    l.bbb_ber_sweep_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(TrialCfg), i32, i32, C.POINTER(Ber)]
    l.bbb_sweep_shard.argtypes = [C.POINTER(TrialCfg), i32, i32, i32, i32, C.POINTER(TrialCfg)]
    l.bbb_multi_release.argtypes = []
    l.bbb_shaper_fill_i16.argtypes = [C.POINTER(TxCfg), vp, u64, u64, i32, vp]
    l.bbb_tx_fill_i16.argtypes = [vp, C.POINTER(TxCfg), vp, u64, u64]
    l.bbb_tx_stream_open.argtypes = [vp, C.POINTER(TxCfg), u64, u64, C.POINTER(vp)]
    l.bbb_tx_stream_next.argtypes = [vp, vp]
    l.bbb_tx_stream_read.argtypes = [vp, vp, u64]
    l.bbb_tx_stream_seek.argtypes = [vp, u64]
    l.bbb_tx_stream_tell.argtypes = [vp, u64p]
    l.bbb_tx_stream_close.argtypes = [vp]
    l.bbb_rx_slice.argtypes = [vp, u64, u64, u64, i32, vp, u64p, i32, vp]
    u8p = C.POINTER(C.c_uint8)
    l.bbb_gf2_berlekamp_massey.argtypes = [u8p, u64, u8p, C.POINTER(C.c_int64)]
    l.bbb_gf2_recur.argtypes = [i32, i32, u64p, u8p, i32, u8p]
    l.bbb_lutopt_set_custom_fill.argtypes = [vp, vp]
    l.bbb_lutopt_set_custom_ber.argtypes = [vp, vp]
    l.bbb_lutopt_attach_custom_library.argtypes = [vp, C.c_char_p]
    l.bbb_gf2_dot.argtypes = [i32, i32, u64p, u8p, u8p]
    u16p, u32p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)
    l.bbb_gf2_poly_is_primitive.argtypes = [u8p, i32, C.POINTER(C.c_int)]
    l.bbb_gf2_poly_modexp.argtypes = [u8p, i32, u64p, i32, u8p]
    l.bbb_lutopt_charpoly.argtypes = [i32, u16p, u32p, u8p, C.POINTER(C.c_int)]
    l.bbb_lutopt_is_full_period.argtypes = [i32, u16p, u32p, C.POINTER(C.c_int)]
    l.bbb_lutopt_save_matrix_file.argtypes = [C.c_char_p, i32, u16p, u32p]
    l.bbb_lutopt_search_candidate.argtypes = [i32, u64, u64, u16p, u32p]
    l.bbb_lutopt_search.argtypes = [i32, u64, u64, u64, u64p, u16p, u32p, C.POINTER(SearchStats), i32, vp]
    for name in SYMBOLS:
        getattr(l, name)          # AttributeError here = header and library out of step
    _lib = l
    return l


def check(rc, what):
    """Map a BBB_E* code to the exception the reference interface raises for it."""
    if rc == BBB_OK:
        return
    l = lib()
    detail = l.bbb_last_error_detail().decode(errors="replace")
    name = l.bbb_strerror(rc).decode()
    if rc == BBB_EINVAL:
        raise ValueError(detail or name)          # e.g. "k=8 invalid for PRBS" (prbs.py:29-30)
    raise BbbError(rc, f"{what} failed ({name})", detail)
