#!/usr/bin/env python3
"""Generate the golden known-answer fixtures under tests/golden/.

The reference's HDL modules cannot be imported here (they need `migen`, which is
not installed), so the vectors are produced by re-evaluating, with plain
Python/numpy, the *software models embedded in the reference's own tests* -- the
exact expressions the reference uses as ITS oracle for the HDL:

  LUTOPT      gateware/bbb/rng.py:134-135   x = mod(dot(a, x), 2); int from x[::-1]
  CLT tree    gateware/bbb/rng.py:173-181   (same tree as software/clt-grng/clt-grng-evaluate.py:10-15)
  PRBS        gateware/bbb/prbs.py:112-113  two-line integer LFSR model, TAPS prbs.py:14
  shaper      gateware/bbb/bitshaper.py:97-109 (coefficients), :143-155 (lfilter model of the test)
  rnghunt     software/rnghunt/src/binary_matrix.rs:183-192 (test_recur KAT, literal)
              software/rnghunt/src/berlekamp_massey.rs:40,45 (PRBS-9 / PRBS-11 strings, literal)

Matrices come from basebandboard_amd/data/lutopt_N.taps (imported from the
reference by tools/import_matrices.py).  Nothing here is imported from the build's
own oracle or product code, so the fixtures are independent of both.
Run in the build container:  python3 tools/make_golden.py
"""
import json
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
DATA = ROOT / "basebandboard_amd" / "data"
OUT = ROOT / "tests" / "golden"

TAPS = {7: 6, 9: 5, 11: 9, 15: 14, 20: 3, 23: 18, 31: 28}   # prbs.py:14


def load_matrix(n):
    rows = [[int(x) for x in l.split()] for l in open(DATA / f"lutopt_{n}.taps") if l.strip()]
    a = np.zeros((len(rows), len(rows)), dtype=np.uint8)
    for r, taps in enumerate(rows):
        a[r, taps] = 1                      # rng.py:51-54 from_packed
    return a


def lutopt_states(a, init, nsteps):
    """rng.py:126-135 model: column vector x, x[0] = LSB of the integer."""
    k = a.shape[0]
    x = np.zeros((k, 1), dtype=np.uint8)
    for i in range(k):
        x[i] = (init >> i) & 1
    a64 = a.astype(np.int64)
    out = []
    for _ in range(nsteps):
        x = np.mod(np.dot(a64, x), 2)
        out.append(int(''.join(str(int(xi)) for xi in x[::-1].flatten()), 2))
    return out


def clt_tree(x_int, n):
    """rng.py:173-181: LSB-first bit list, log2(n) levels of y[p//2] = x[p] - x[p+1]."""
    logn = int(np.log2(n))
    x = np.array([int(c) for c in bin(x_int)[2:].rjust(n, "0")[::-1]])
    for level in range(logn):
        level_n = 2**(logn - level)
        y = np.zeros(level_n//2, dtype=np.int16)
        for pair in range(0, level_n, 2):
            y[pair//2] = x[pair] - x[pair+1]
        x = y
    return int(x[0])


def wrap_signed(v, bits):
    """CLTGRNG.x is Signal((logn, True)) (rng.py:78): the tree value truncated to logn bits, signed."""
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def prbs_bits(k, nbits, lfsr=1):
    bits = []
    for _ in range(nbits):
        bit = ((lfsr >> (k-1)) ^ (lfsr >> TAPS[k]-1)) & 1      # prbs.py:112
        lfsr = ((lfsr << 1) | bit) & ((1 << k)-1)              # prbs.py:113
        bits.append(bit)
    return bits, lfsr


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    # ---- LUTOPT + CLT -------------------------------------------------------
    lut = {}
    for n, nsteps in ((16, 256), (32, 256), (64, 256), (128, 256), (256, 4096)):
        a = load_matrix(n)
        logn = int(np.log2(n))
        states = lutopt_states(a, 1, nsteps)
        tree = [clt_tree(s, n) for s in states]
        lut[str(n)] = {
            "init": "0x1",
            "states_hex": [hex(s) for s in states[:64]],
            "state_last_hex": hex(states[-1]),
            "clt_tree": tree,                                   # un-truncated tree value
            "clt_out": [wrap_signed(t, logn) for t in tree],    # logn-bit signed output
        }
    # a second seed for n256: init = 0xDEADBEEF... pattern (bit i of the integer = x[i])
    a = load_matrix(256)
    init2 = int("0123456789abcdef" * 4, 16)
    st = lutopt_states(a, init2, 512)
    lut["256_seed2"] = {
        "init": hex(init2),
        "states_hex": [hex(s) for s in st[:8]],
        "state_last_hex": hex(st[-1]),
        "clt_out": [wrap_signed(clt_tree(s, 256), 8) for s in st],
    }
    # extreme input: bits set exactly where the tree weight is +1 -> +128 -> wraps to -128
    mplus = sum(1 << i for i in range(256) if bin(i).count("1") % 2 == 0)
    lut["256_extreme"] = {"x_hex": hex(mplus), "clt_tree": clt_tree(mplus, 256),
                          "clt_out": wrap_signed(clt_tree(mplus, 256), 8)}
    json.dump(lut, open(OUT / "lutopt_clt.json", "w"), indent=0)

    # ---- PRBS ---------------------------------------------------------------
    pr = {}
    for k in TAPS:
        bits, s = prbs_bits(k, 4096)
        pr[str(k)] = {"init": 1, "bits": "".join(map(str, bits)), "state_after": s}
        b2, s2 = prbs_bits(k, 1024, lfsr=(0x5A5A5A5A & ((1 << k) - 1)) | 1)
        pr[str(k) + "_seed2"] = {"init": (0x5A5A5A5A & ((1 << k) - 1)) | 1,
                                 "bits": "".join(map(str, b2)), "state_after": s2}
    # literal strings held by the reference's Rust tests (cross-pin of PRBS9 / PRBS11)
    pr["rnghunt_bm_prbs9"] = "0000100011000010011"       # berlekamp_massey.rs:40
    pr["rnghunt_bm_prbs11"] = "00000000101000000100010"  # berlekamp_massey.rs:45
    json.dump(pr, open(OUT / "prbs.json", "w"), indent=0)

    # ---- rnghunt GF(2) KATs (literals from the Rust unit tests) -------------
    gf2 = {
        "test_recur": {   # binary_matrix.rs:183-192: column-major words, MSbit = row 0
            "nrows": 8, "ncols": 8,
            "col_words_hex": ["0x7400000000000000", "0x5800000000000000", "0xC500000000000000",
                              "0xD000000000000000", "0xD500000000000000", "0xE600000000000000",
                              "0xF100000000000000", "0x4700000000000000"],
            "x_bits": [1, 0, 1, 0, 1, 0, 1, 0], "n": 24,
            "out_bits": [1, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0],
        }
    }
    # binary_matrix.rs:133-180 (test_dot): a 64 x 128 matrix, a 128-bit vector and their product; the numbers
    # are read out of the reference's test at generation time (data, not code)
    import re
    ref = pathlib.Path("/root/reference/software/rnghunt/src/binary_matrix.rs")
    if ref.exists():
        src = ref.read_text()
        body = src[src.index("fn test_dot()"):src.index("fn test_recur()")]
        words = re.findall(r"0x[0-9A-Fa-f]{16}", body)
        vecs = re.findall(r"&?\[\s*((?:[01],\s*)+[01])\s*\]", body)
        assert len(words) == 128 and len(vecs) == 2
        gf2["test_dot"] = {"nrows": 64, "ncols": 128, "col_words_hex": words,
                           "x_bits": [int(c) for c in re.findall(r"[01]", vecs[0])],
                           "out_bits": [int(c) for c in re.findall(r"[01]", vecs[1])]}
        assert len(gf2["test_dot"]["x_bits"]) == 128 and len(gf2["test_dot"]["out_bits"]) == 64
    else:                                   # no reference checkout here: keep the committed vector
        gf2["test_dot"] = json.load(open(OUT / "gf2.json"))["test_dot"]
    json.dump(gf2, open(OUT / "gf2.json", "w"), indent=0)
    # ---- pulse shaper (gateware/bbb/bitshaper.py) ----------------------------------------
    # coefficient sets exactly as PRBSShaper.from_rcf computes them (bitshaper.py:97-109) for the
    # 32 roll-offs TX uses (tx.py:54: np.linspace(0, 1, 32)), and the expected waveform from the
    # model in the reference's own test (bitshaper.py:143-155): +-1 impulses at the midpoint of each
    # 8-sample bit period, scipy.signal.lfilter with the pulse, 13 samples of pipeline delay:
    #     shaped[73:] == lfilter(c, [1], y)[60:-13]
    import scipy.signal
    T = 8

    def rcf(beta):
        t = np.arange(-32, 32)
        if beta != 0.0:
            replace = np.where(np.abs(t) == T/(2*beta))
            t[replace] = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            c = 1/T * np.sinc(t/T) * np.cos(np.pi * beta * t/T)/(1-(2*beta*t/T)**2)
        if beta != 0.0:
            c[replace] = np.pi/(4*T) * np.sinc(1/(2*beta))
        return (c * T * 254).astype(int).tolist()          # bitshaper.py:107 (np.int == int64)

    def model(c, k, nsamp):
        bits, _ = prbs_bits(k, nsamp // 8)
        y = np.zeros(nsamp)
        y[4::8] = 2*np.array(bits) - 1                      # bitshaper.py:151-153
        f = scipy.signal.lfilter(np.array(c), [1], y)       # bitshaper.py:154
        return [int(v) for v in f[60:nsamp-13]]             # == shaped[73:nsamp]   (bitshaper.py:155)

    betas = np.linspace(0, 1, 32).tolist()                  # tx.py:54
    sets = [rcf(b) for b in betas]
    rect = [0]*30 + [254]*4 + [0]*30                        # bitshaper.py:108
    sh = {"betas": betas, "rcf_coeffs": sets, "rect": rect,
          "test_prbs_shaper": {"k": 9, "beta": 0.5, "coeffs": rcf(0.5), "nsamples": 320,
                               "shaped_from_73": model(rcf(0.5), 9, 320)},
          "prbs31_set10": {"k": 31, "set": 10, "nsamples": 4096, "shaped_from_73": model(sets[10], 31, 4096)},
          "prbs7_set31": {"k": 7, "set": 31, "nsamples": 2048, "shaped_from_73": model(sets[31], 7, 2048)},
          "prbs15_rect": {"k": 15, "nsamples": 1024, "shaped_from_73": model(rect, 15, 1024)}}
    json.dump(sh, open(OUT / "shaper.json", "w"))
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
