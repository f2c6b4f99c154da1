#!/usr/bin/env python3
"""Prime factorisations of 2^n - 1 for the orders this package tests for full period.

A degree-n polynomial p over GF(2) is primitive iff x^(2^n - 1) = 1 (mod p) and x^((2^n - 1)/q) != 1
for every prime q dividing 2^n - 1 (software/rnghunt/src/binary_polynomial.rs:166-199 states the same
test; its table software/rnghunt/src/factors_data.rs lists the cofactors (2^n - 1)/q).  The primes are
mathematical constants: this script DERIVES them (cyclotomic splitting 2^n - 1 = prod_{d | n} Phi_d(2),
trial division, Pollard rho; the two large Fermat-number splits F7, F8 are entered as known values) and
proves each line (product equals 2^n - 1, every factor passes Miller-Rabin).  When the reference
checkout is present the cofactors are also compared with its table.

Output: basebandboard_amd/data/mersenne_factors.txt, one line per n:  n: q1 q2 ...   (distinct primes)
"""
import math
import pathlib
import random
import re
import sys

ORDERS = (4, 7, 8, 9, 11, 15, 16, 20, 23, 31, 32, 33, 64, 128, 192, 200, 256, 512)

# published splits of the Fermat numbers F7 = 2^128 + 1 and F8 = 2^256 + 1 (verified below)
KNOWN = (59649589127497217, 5704689200685129054721,
         1238926361552897, 93461639715357977769163558199606896584051237541638188580280321)


def is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small + (41, 43, 47, 53, 59, 61, 67, 71):
        if a % n == 0:
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def rho(n, rnd):
    if n % 2 == 0:
        return 2
    while True:
        c, y, m = rnd.randrange(1, n), rnd.randrange(1, n), 256
        g = r = q = 1
        while g == 1:
            x = y
            for _ in range(r):
                y = (y * y + c) % n
            k = 0
            while k < r and g == 1:
                ys = y
                for _ in range(min(m, r - k)):
                    y = (y * y + c) % n
                    q = q * abs(x - y) % n
                g = math.gcd(q, n)
                k += m
            r *= 2
        if g == n:
            g = 1
            while g == 1:
                ys = (ys * ys + c) % n
                g = math.gcd(abs(x - ys), n)
        if g != n:
            return g


def factor(n, rnd, out):
    if n == 1:
        return
    if is_prime(n):
        out.add(n)
        return
    for k in KNOWN:
        if n % k == 0 and n != k:
            factor(k, rnd, out)
            factor(n // k, rnd, out)
            return
    for p in range(3, 100000, 2):
        if n % p == 0:
            out.add(p)
            while n % p == 0:
                n //= p
            factor(n, rnd, out)
            return
    if n.bit_length() > 100:
        raise SystemExit(f"no split known for {n}")
    d = rho(n, rnd)
    factor(d, rnd, out)
    factor(n // d, rnd, out)


def cyclotomic_at_2(d):
    """Phi_d(2) by dividing 2^d - 1 by every Phi_e(2), e | d, e < d."""
    v = 2 ** d - 1
    for e in range(1, d):
        if d % e == 0:
            v //= cyclotomic_at_2(e)
    return v


def primes_of(n):
    rnd = random.Random(n)
    out = set()
    for d in range(1, n + 1):
        if n % d == 0:
            factor(cyclotomic_at_2(d), rnd, out)
    r = 2 ** n - 1
    rest = r
    for q in out:
        assert is_prime(q) and r % q == 0
        while rest % q == 0:
            rest //= q
    assert rest == 1, n
    return sorted(out)


def reference_cofactors(path):
    """{n: set of cofactors} parsed from the reference's table (development-time cross-check only)."""
    text = pathlib.Path(path).read_text()
    body = text[text.index("= [") + 3:]
    table, n = {}, 0
    for blk in re.finditer(r"&\[\s*((?:&\[[^\]]*\],?\s*)+)\]", body):
        n += 1
        vals = set()
        for words in re.findall(r"&\[([^\]]*)\]", blk.group(1)):
            v = 0
            for w in re.findall(r"0x[0-9a-fA-F]+", words):
                v = (v << 64) | int(w, 16)
            vals.add(v)
        table[n] = vals
    return table


def emit_inc(data_path, out_path):
    """C++ table for csrc/: per order the exponents r = 2^n - 1 and r / q (little-endian 64-bit words)."""
    rows = []
    for line in pathlib.Path(data_path).read_text().splitlines():
        if not line or line.startswith("#"):
            continue
        n, qs = line.split(":")
        rows.append((int(n), [int(q) for q in qs.split()]))
    words, entries = [], []
    for n, qs in rows:
        r = 2 ** n - 1
        assert all(r % q == 0 for q in qs)
        W = (n + 63) // 64
        entries.append((n, len(qs) + 1, len(words)))
        for e in [r] + [r // q for q in qs]:
            words += [(e >> (64 * i)) & (2 ** 64 - 1) for i in range(W)]
    out = ["// GENERATED by tools/make_factors.py --inc from data/mersenne_factors.txt -- do not edit.",
           "// exponents of the primitivity test: 2^n - 1 first, then (2^n - 1)/q for every prime q | 2^n - 1;",
           "// each is ceil(n/64) little-endian words starting at kMersenneWords[offset]",
           "struct MersenneEntry { int n, nexp, offset; };",
           f"static const MersenneEntry kMersenne[{len(entries)}] = {{" + ", ".join(f"{{{n}, {k}, {o}}}" for n, k, o in entries) + "};",
           f"static const uint64_t kMersenneWords[{len(words)}] = {{"]
    for i in range(0, len(words), 4):
        out.append("  " + ", ".join(f"0x{w:016x}ull" for w in words[i:i + 4]) + ",")
    out.append("};")
    pathlib.Path(out_path).write_text("\n".join(out) + "\n")


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--inc":
        emit_inc(sys.argv[2], sys.argv[3])
        return
    root = pathlib.Path(__file__).resolve().parent.parent
    lines = []
    ref = pathlib.Path("/root/reference/software/rnghunt/src/factors_data.rs")
    table = reference_cofactors(ref) if ref.exists() else None
    for n in ORDERS:
        qs = primes_of(n)
        lines.append(f"{n}: " + " ".join(map(str, qs)))
        if table is not None:
            r = 2 ** n - 1
            mine = {r} | {r // q for q in qs}
            assert mine == table[n], f"n={n}: cofactors differ from the reference's table"
    out = root / "basebandboard_amd" / "data" / "mersenne_factors.txt"
    out.write_text("# distinct primes dividing 2^n - 1 (tools/make_factors.py)\n" + "\n".join(lines) + "\n")
    print(f"{len(lines)} orders -> {out}" + (" (cofactors equal the reference's table)" if table is not None else ""))


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
