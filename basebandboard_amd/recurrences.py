"""Maximum-period LUTOPT recurrence matrices shipped with the package.

The data files are the reference's found matrices (software/rnghunt/matrices/N; identical to
gateware/bbb/rng_recurrences.py n16..n256), kept in the same text format so that files written
by the reference's search tool (software/rnghunt/src/bin/rnghunt.rs:51-53) load unchanged.
`nN` below are the packed tap lists, the form gateware/bbb/rng_recurrences.py exposes.
"""
import pathlib

_DATA = pathlib.Path(__file__).resolve().parent / "data"
SIZES = (16, 32, 64, 128, 192, 256, 512)


def matrix_path(n):
    p = _DATA / f"lutopt_{n}.txt"
    if not p.exists():
        raise ValueError(f"no shipped recurrence for n={n} (have {SIZES})")
    return p


def load_packed(path):
    """Text matrix (line r, char c = A[r][c]) -> list of per-row tap lists (util/pack.py:18-23)."""
    rows = [l.strip() for l in open(path) if l.strip()]
    n = len(rows)
    if any(len(r) != n or set(r) - {"0", "1"} for r in rows):
        raise ValueError(f"{path}: not a square 0/1 matrix")
    return [[c for c, ch in enumerate(r) if ch == "1"] for r in rows]


def __getattr__(name):
    if name.startswith("n") and name[1:].isdigit() and int(name[1:]) in SIZES:
        return load_packed(matrix_path(int(name[1:])))
    raise AttributeError(name)
