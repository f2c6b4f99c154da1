#!/usr/bin/env python3
"""Command-line entry of the network generator, which lives inside the package
(basebandboard_amd/gen_lutopt_kernel.py) so that LUTOPT.specialise() needs nothing outside it.
Loaded by file path: generating a kernel must not import torch.

Usage: gen_lutopt_kernel.py <matrix.txt> <out.inc>
"""
import importlib.util
import pathlib
import sys

_p = pathlib.Path(__file__).resolve().parent.parent / "basebandboard_amd" / "gen_lutopt_kernel.py"
_spec = importlib.util.spec_from_file_location("bbb_gen_lutopt_kernel", _p)
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})

if __name__ == "__main__":
    _m.main()
