#!/usr/bin/env python3
"""Fixtures produced by RUNNING the reference's own Python in the build container.

Unlike tools/make_golden.py (which re-evaluates the model lines embedded in the reference's HDL
tests, because those files need `migen`), every vector written here is the output of reference code
executed unchanged from where it lies under /root/reference:

  ref_recur.json     software/rnghunt/util/binarymatrix.py:30-35  recur(a, x, n), imported as a module:
                     `x = np.mod(np.dot(a, x), 2); out.append(x[0][0])`.  Called for every matrix the
                     reference ships (software/rnghunt/matrices/{16..512}, loaded the way
                     util/verify.py:7-14 loads them) from init 1 and from a second seed, 4096 steps
                     (bit 0 of every state), and again on the index-swapped copies P A P^T, P x for
                     every state bit j, 64 steps, so that recur's "bit 0" reads out bit j: the full
                     state of the first 64 steps is thereby pinned by reference code, not only bit 0.
  ref_clt.npz        software/clt-grng/clt-grng-evaluate.py executed unchanged (compile + exec of the
                     file's text in a fresh namespace, MPLBACKEND=Agg, np.random.seed(S) first).  It
                     runs its 100 000-sample tree (:8-16) and prints mean / variance (:30-31); the
                     `normed` keyword of :34 no longer exists in matplotlib 3.10, so the script ends
                     there with an AttributeError which is caught -- `samples` is complete by then.
                     The 100 000 x 256 input bits are re-drawn with the same seed and the same
                     `np.random.randint(2, size=n)` calls and stored (packed) beside `samples`.
  ref_pack.json      software/rnghunt/util/pack.py run as __main__ on each matrices/N: its printed
                     packed tap lists (the format of gateware/bbb/rng_recurrences.py), plus the lists
                     of rng_recurrences.py itself (imports without dependencies).
  ref_words.npz      software/rnghunt/util/verify.py run as __main__ on matrices/256 and matrices/192
                     with a seeded np.random: its dieharder dump `outnums` (:37-52) = the uniform word
                     stream LUTOPT.x in bulk (200 000 states as 32-bit words, x[32j] the MSB of word
                     j).  The script also shells out to an external `./ppsearch` that is not in the
                     repository; that one call is answered by a no-op (it does not touch the data).
                     Stored: the start state verify.py drew, the first 2048 and last 512 states'
                     words, and the sha256 of the whole word stream.
  ref_lfsr.json      software/rnghunt/util/lfsr.py run as __main__: the two LFSR strings its Rust
                     Berlekamp-Massey tests use.

Only the fixtures travel to the GPU box; the reference's source stays in /root/reference.
Run in the build container:   python3 tools/make_golden_ref.py
"""
import contextlib
import hashlib
import importlib.util
import io
import json
import os
import pathlib
import runpy
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

REF = pathlib.Path("/root/reference")
ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"
NS = (16, 32, 64, 128, 192, 256, 512)
SEED = 20261004


def load_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_matrix(n):
    """matrices/N exactly as util/verify.py:7-14 and util/pack.py:6-14 read it."""
    lines = [l.strip() for l in open(REF / "software/rnghunt/matrices" / str(n))]
    a = np.empty((n, n), dtype=np.uint8)
    for r in range(n):
        a[r] = [int(c) for c in lines[r]]
    return a


def int_to_col(v, n):
    x = np.zeros((n, 1), dtype=np.uint8)
    for i in range(n):
        x[i] = (v >> i) & 1
    return x


def second_seed(n):
    return int("0123456789abcdef" * 8, 16) & ((1 << n) - 1) | 1


def make_recur(bm):
    out = {"source": "software/rnghunt/util/binarymatrix.py:30-35 recur(a, x, n), imported and called",
           "matrices": "software/rnghunt/matrices/N", "nsteps_bit0": 4096, "nsteps_full": 64}
    for n in NS:
        a = load_matrix(n)
        ent = {}
        for label, init in (("init1", 1), ("seed2", second_seed(n))):
            x0 = int_to_col(init, n)
            bit0 = bm.recur(a, x0, 4096)
            # full states: swap index 0 <-> j  (P = P^T = P^-1), y = P x obeys y' = (P A P) y
            cols = []
            for j in range(n):
                p = np.arange(n)
                p[0], p[j] = j, 0
                ap = a[p][:, p]
                xp = x0[p]
                cols.append(bm.recur(ap, xp, 64))
            states = []
            for t in range(64):
                v = 0
                for j in range(n):
                    v |= int(cols[j][t]) << j
                states.append(hex(v))
            assert [int(s, 16) & 1 for s in states] == [int(b) for b in bit0[:64]]
            ent[label] = {"init": hex(init), "bit0": "".join(str(int(b)) for b in bit0), "states_hex": states}
        out[str(n)] = ent
        print("recur", n, "done", flush=True)
    json.dump(out, open(OUT / "ref_recur.json", "w"), indent=0)


def make_clt():
    path = REF / "software/clt-grng/clt-grng-evaluate.py"
    code = compile(path.read_text(), str(path), "exec")
    ns = {"__name__": "__main__"}
    np.random.seed(SEED)
    stdout = io.StringIO()
    ended = "ran to the end"
    with contextlib.redirect_stdout(stdout):
        try:
            exec(code, ns)
        except AttributeError as e:          # matplotlib >= 3.1: hist(..., normed=True), line 34
            ended = "AttributeError at the first plt.hist: " + str(e)[:80]
    samples = np.array(ns["samples"])
    n, nsamp = ns["n"], ns["nsamp"]
    # the very same draws again
    np.random.seed(SEED)
    bits = np.empty((nsamp, n), dtype=np.uint8)
    for i in range(nsamp):
        bits[i] = np.random.randint(2, size=n)
    # x[i] of the script = bit i of the word (LSB first), 4 little-endian u64 per sample
    packed = np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(nsamp, n // 64)
    np.savez_compressed(OUT / "ref_clt.npz", samples=samples.astype(np.int16), states=packed,
                        seed=np.int64(SEED), n=np.int64(n))
    meta = {"source": "software/clt-grng/clt-grng-evaluate.py executed unchanged, np.random.seed(%d) first" % SEED,
            "ended": ended, "printed": stdout.getvalue().strip().splitlines(),
            "mean": float(samples.mean()), "var": float(samples.var())}
    json.dump(meta, open(OUT / "ref_clt_meta.json", "w"), indent=0)
    print("clt", meta["printed"], ended, flush=True)


def run_script(path, argv, cwd=None, patch_subprocess=False):
    old_argv, old_cwd = sys.argv, os.getcwd()
    stdout = io.StringIO()
    import subprocess
    real_run = subprocess.run
    try:
        sys.argv = [str(path)] + [str(a) for a in argv]
        if cwd:
            os.chdir(cwd)
        if patch_subprocess:
            subprocess.run = lambda *a, **k: None       # `./ppsearch` is not part of the repository
        with contextlib.redirect_stdout(stdout):
            runpy.run_path(str(path), run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
        subprocess.run = real_run
    return stdout.getvalue()


def make_pack():
    out = {"source": "software/rnghunt/util/pack.py run on software/rnghunt/matrices/N; "
                     "gateware/bbb/rng_recurrences.py imported"}
    rr = load_module(REF / "gateware/bbb/rng_recurrences.py", "ref_rng_recurrences")
    for n in NS:
        text = run_script(REF / "software/rnghunt/util/pack.py", [REF / "software/rnghunt/matrices" / str(n)])
        body = text[text.index("Packed:") + len("Packed:"):]
        packed = eval(body.strip().replace(",\n]", "]"))
        ent = {"pack_py": packed}
        if hasattr(rr, "n%d" % n):
            ent["rng_recurrences"] = getattr(rr, "n%d" % n)
        out[str(n)] = ent
    json.dump(out, open(OUT / "ref_pack.json", "w"))
    print("pack done", flush=True)


def make_words():
    arrays = {}
    meta = {"source": "software/rnghunt/util/verify.py run as __main__ (np.random.seed first; the external "
                      "./ppsearch call answered by a no-op): its `outnums` dieharder dump, :37-52"}
    for n in (192, 256):
        with tempfile.TemporaryDirectory() as tmp:
            np.random.seed(SEED + n)
            run_script(REF / "software/rnghunt/util/verify.py", [REF / "software/rnghunt/matrices" / str(n)],
                       cwd=tmp, patch_subprocess=True)
            lines = open(os.path.join(tmp, "outnums")).read().splitlines()
        header, body = lines[:6], lines[6:]
        words = np.array([int(l) for l in body], dtype=np.uint32)
        wps = n // 32
        assert len(words) == 200000 * wps and header[4].startswith("count: %d" % len(words))
        # the start state: verify.py draws b (:19), uses it for the BM sequence, then draws b again (:34)
        np.random.seed(SEED + n)
        np.random.randint(2, size=(n, 1))
        b = np.random.randint(2, size=(n, 1)).astype(np.uint8)
        arrays["init_bits_%d" % n] = b[:, 0].copy()
        arrays["head_%d" % n] = words[:2048 * wps]
        arrays["tail_%d" % n] = words[-512 * wps:]
        meta[str(n)] = {"nstates": 200000, "words_per_state": wps,
                        "sha256_le_u32": hashlib.sha256(words.astype("<u4").tobytes()).hexdigest()}
        print("words", n, meta[str(n)], flush=True)
    np.savez_compressed(OUT / "ref_words.npz", **arrays)
    json.dump(meta, open(OUT / "ref_words_meta.json", "w"), indent=0)


def make_lfsr():
    text = run_script(REF / "software/rnghunt/util/lfsr.py", [])
    json.dump({"source": "software/rnghunt/util/lfsr.py run as __main__", "lines": text.split()},
              open(OUT / "ref_lfsr.json", "w"))


def main():
    assert REF.exists(), "run this in the build container (needs /root/reference)"
    OUT.mkdir(parents=True, exist_ok=True)
    which = set(sys.argv[1:]) or {"recur", "clt", "pack", "words", "lfsr"}
    if "lfsr" in which:
        make_lfsr()
    if "pack" in which:
        make_pack()
    if "clt" in which:
        make_clt()
    if "recur" in which:
        make_recur(load_module(REF / "software/rnghunt/util/binarymatrix.py", "ref_binarymatrix"))
    if "words" in which:
        make_words()


if __name__ == "__main__":
    main()
