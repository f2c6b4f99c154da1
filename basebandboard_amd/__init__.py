"""basebandboard_amd -- MI355X (gfx950) implementation of basebandboard's AWGN / PRBS Monte-Carlo
path: LUTOPT uniform generator, CLT Gaussian generator, PRBS generator / error detector, the
fused BPSK bit-error trial, and the pulse shaper / transmitter output stream.  Compute lives in libbbb_hip.so (C ABI: include/bbb.h); these modules
mirror the reference's Python interface (gateware/bbb/rng.py, prbs.py, bitshaper.py, tx.py, rx.py).
"""
from .prbs import PRBS, PRBSErrorDetector, TAPS          # noqa: F401
from .rng import CLTGRNG, LUTOPT, SampleStream          # noqa: F401
from .channel import Trial, run_trials, run_trials_into, sweep, gpu_runner, shard, ContinuedTrials, prepare  # noqa: F401
from .bitshaper import PRBSShaper, Pulser                # noqa: F401
from .tx import TX, WaveformStream                       # noqa: F401
from .rx import RX                                       # noqa: F401
from . import gf2, recurrences                           # noqa: F401
