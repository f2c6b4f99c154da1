#!/usr/bin/env python3
"""Import the LUTOPT recurrence matrices (pure data) from the reference checkout.

Reads  /root/reference/software/rnghunt/matrices/N   (N lines of N chars '0'/'1',
       line r char c = A[r][c]; format written by software/rnghunt/src/bin/rnghunt.rs:51-53
       and read by software/rnghunt/util/pack.py:6-18)
Writes basebandboard_amd/data/lutopt_N.taps          (packed tap lists: line r = the column indices
       of the ones of row r, i.e. the `packed` form of gateware/bbb/rng.py:42-55, as plain numbers)
and cross-checks every N <= 256 against the packed tap lists
gateware/bbb/rng_recurrences.py (nN), which is what gateware/bbb/tx.py:15,70 feeds
to LUTOPT.from_packed.  Only runs in the build container (the reference does not
travel to the GPU box); the output files are committed.
"""
import importlib.util
import pathlib
import sys

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent.parent / "basebandboard_amd" / "data"


def main():
    spec = importlib.util.spec_from_file_location(
        "rng_recurrences", REF / "gateware/bbb/rng_recurrences.py")
    rec = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rec)
    OUT.mkdir(parents=True, exist_ok=True)
    for n in (16, 32, 64, 128, 192, 256, 512):
        rows = [l.strip() for l in open(REF / f"software/rnghunt/matrices/{n}") if l.strip()]
        assert len(rows) == n and all(len(r) == n and set(r) <= {"0", "1"} for r in rows), n
        packed = [[c for c, ch in enumerate(r) if ch == "1"] for r in rows]
        twin = getattr(rec, f"n{n}", None)
        if twin is not None:
            assert [sorted(t) for t in twin] == packed, f"matrices/{n} != rng_recurrences.n{n}"
            status = "== rng_recurrences.n%d" % n
        else:
            status = "(no packed twin in rng_recurrences.py)"
        wr = sorted(set(len(p) for p in packed))
        (OUT / f"lutopt_{n}.taps").write_text("\n".join(" ".join(map(str, t)) for t in packed) + "\n")
        print(f"n={n}: row weights {wr} {status}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
