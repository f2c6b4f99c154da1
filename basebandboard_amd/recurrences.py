"""Maximum-period LUTOPT recurrence matrices shipped with the package.

The data files hold the reference's found matrices (software/rnghunt/matrices/N; identical to
gateware/bbb/rng_recurrences.py n16..n256) as packed tap lists, one row per line.  Files in the
reference's own 0/1 text format (written by software/rnghunt/src/bin/rnghunt.rs:51-53) load too:
`load_packed` accepts both, and the C ABI has `bbb_lutopt_load_matrix_file` for them.
`nN` below are the packed tap lists, the form gateware/bbb/rng_recurrences.py exposes.
"""
import pathlib

_DATA = pathlib.Path(__file__).resolve().parent / "data"
SIZES = (16, 32, 64, 128, 192, 256, 512)


def matrix_path(n):
    p = _DATA / f"lutopt_{n}.taps"
    if not p.exists():
        raise ValueError(f"no shipped recurrence for n={n} (have {SIZES})")
    return p


def load_packed(path):
    """Per-row tap lists from either format: packed taps (one line per row, space separated column
    indices -- how the shipped matrices are stored) or the reference's 0/1 text matrix
    (software/rnghunt/matrices/N: line r, char c = A[r][c]; util/pack.py:6-18)."""
    rows = [l.strip() for l in open(path) if l.strip()]
    n = len(rows)
    if all(set(r) <= {"0", "1"} and len(r) == n for r in rows) and n > 1:
        return [[c for c, ch in enumerate(r) if ch == "1"] for r in rows]
    packed = [[int(x) for x in r.split()] for r in rows]
    if any(not t or min(t) < 0 or max(t) >= n for t in packed):
        raise ValueError(f"{path}: not a recurrence matrix")
    return packed


def __getattr__(name):
    if name.startswith("n") and name[1:].isdigit() and int(name[1:]) in SIZES:
        return load_packed(matrix_path(int(name[1:])))
    raise AttributeError(name)
