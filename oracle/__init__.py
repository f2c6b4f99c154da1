"""CPU oracle for the basebandboard AWGN / PRBS path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package `basebandboard_amd` never does.
"""
from .oracle import *  # noqa: F401,F403
