#!/usr/bin/env python3
"""Emit the straight-line bit-sliced LUTOPT step + CLT vertical counter for one matrix.

Input : a recurrence matrix in the reference's text format (software/rnghunt/matrices/N,
        shipped as packed tap lists in basebandboard_amd/data/lutopt_N.taps).
Output: a C++ include for the HIP kernels with

  lutoptN_step(a, b, cnt)   b = A*a over GF(2) on 32 generators per lane
                            (gateware/bbb/rng.py:38-40: row r = XOR of its taps, all rows
                            from the OLD state), and cnt[0..7] = the bit planes of the int8
                            CLTGRNG sample of the NEW state (rng.py:96-108).

Bit-sliced arithmetic.  Register p holds state bit p of 32 independent generators.
The adder tree of rng.py:96-105 equals sum_i (-1)^popcount(i) x[i]; with
y_i = x_i (popcount(i) even) or 1 - x_i (odd), T = sum_i y_i lies in [0, n] and the
log2(n)-bit signed output is (T - n/2) mod n: for n = 256 the low 8 bits of T with bit 7
flipped.  T is formed by a carry-save adder network whose cells are single gfx950
V_BITOP3_B32 instructions (sum = 3-input XOR, carry = majority), with the input
inversions folded into the truth tables.

Usage: gen_lutopt_kernel.py <matrix.txt> <out.inc>
"""
import sys
import zlib


def load(path):
    """Packed tap lists (one row per line) or the reference's 0/1 text matrix."""
    rows = [l.strip() for l in open(path) if l.strip()]
    n = len(rows)
    if n > 1 and all(set(r) <= {"0", "1"} and len(r) == n for r in rows):
        return n, [[c for c, ch in enumerate(r) if ch == "1"] for r in rows]
    return n, [[int(x) for x in r.split()] for r in rows]


def tt3(f, inv):
    """8-bit truth table of f(a^inv0, b^inv1, c^inv2) in BITOP3 order (a=0xF0, b=0xCC, c=0xAA)."""
    t = 0
    for idx in range(8):
        a, b, c = (idx >> 2) & 1, (idx >> 1) & 1, idx & 1
        if f(a ^ inv[0], b ^ inv[1], c ^ inv[2]):
            t |= 1 << idx
    return t


def row_order(n, taps):
    """Greedy order of row evaluation that retires old planes early (register pressure)."""
    uses = [0] * n
    for r in range(n):
        for c in taps[r]:
            uses[c] += 1
    remaining = set(range(n))
    left = uses[:]
    order = []
    while remaining:
        best, best_key = None, None
        for r in remaining:
            kills = sum(1 for c in taps[r] if left[c] == 1)
            near = sum(1.0 / left[c] for c in taps[r])
            key = (kills, near, -r)
            if best_key is None or key > best_key:
                best, best_key = r, key
        order.append(best)
        remaining.remove(best)
        for c in taps[best]:
            left[c] -= 1
    return order


def plan_parking(n, taps, order, npark, lookahead):
    """Choose `npark` groups of 4 consecutively computed rows whose planes are parked in LDS between
    their birth (step s) and shortly before their first use (step s+1).

    Returns (groups, reload_at, store_at): groups[g] = 4 plane indices; reload_at[pos] / store_at[pos]
    = group ids whose ds_read_b128 is emitted before / whose ds_write_b128 is emitted after the row
    at schedule position pos.  A group is only eligible if its reload precedes its store within a
    step (one LDS slot per group, no double buffering)."""
    pos = {r: i for i, r in enumerate(order)}
    first_use = [n] * n
    for r in range(n):
        for c in taps[r]:
            first_use[c] = min(first_use[c], pos[r])
    cands = []
    for j in range(n // 4):
        members = order[4 * j: 4 * j + 4]
        f = min(first_use[p] for p in members)
        reload_pos = max(0, f - lookahead)
        store_pos = 4 * j + 3
        if reload_pos > store_pos:
            continue
        # register-time saved: from the store to the end of the step, plus from the start of the
        # next step to the reload, for each of the four planes
        saved = 4 * ((n - store_pos) + reload_pos)
        cands.append((saved, j, members, reload_pos, store_pos))
    cands.sort(reverse=True)
    chosen = cands[:npark]
    groups, reload_at, store_at = [], {}, {}
    for g, (_, j, members, rp, sp) in enumerate(chosen):
        groups.append(members)
        reload_at.setdefault(rp, []).append(g)
        store_at.setdefault(sp, []).append(g)
    return groups, reload_at, store_at


def generate(n, taps, npark=0, lookahead=6):
    logn = n.bit_length() - 1
    assert 1 << logn == n and n >= 16
    out = []
    emit = out.append
    nops = 0
    tmp_id = [0]

    def tmp():
        tmp_id[0] += 1
        return f"t{tmp_id[0]}"

    # ---- vertical counter state: per weight level a list of (expr, inverted) -------------
    nlev = logn          # output planes 0..logn-1 ; level logn (the carry-out) is dropped
    levels = [[] for _ in range(nlev + 2)]

    def push(level, item):
        nonlocal nops
        if level >= nlev:
            return       # carries out of the top output bit are never needed
        levels[level].append(item)
        while len(levels[level]) >= 3:
            (a, ia), (b, ib), (c, ic) = levels[level][:3]
            del levels[level][:3]
            s = tmp()
            emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: x ^ y ^ z, (ia, ib, ic)):02x});")
            nops += 1
            if level + 1 < nlev:
                cy = tmp()
                emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: (x & y) | (x & z) | (y & z), (ia, ib, ic)):02x});")
                nops += 1
                push(level + 1, (cy, 0))
            levels[level].append((s, 0))

    emit(f"// GENERATED by tools/gen_lutopt_kernel.py from lutopt_{n}.taps -- do not edit.")
    flat = ",".join(",".join(map(str, t)) for t in taps)
    if not npark:
        emit(f"#define LUTOPT{n}_TAPS_CRC 0x{zlib.crc32(flat.encode()) & 0xffffffff:08x}u")
    order = row_order(n, taps)
    groups, reload_at, store_at = plan_parking(n, taps, order, npark, lookahead) if npark else ([], {}, {})
    fn = f"lutopt{n}p" if npark else f"lutopt{n}"
    if npark:
        emit(f"#define LUTOPT{n}_NPARK {len(groups)}")
        emit(f"typedef uint32_t lutopt{n}_v4 __attribute__((ext_vector_type(4)));")
        emit(f"// park[g*64 + lane] holds planes {{p0,p1,p2,p3}} of group g between a step and the next")
        emit(f"static __device__ __forceinline__ void {fn}_park_init(const uint32_t (&a)[{n}], lutopt{n}_v4 *park, unsigned lane)")
        emit("{")
        for g, m in enumerate(groups):
            emit(f"  park[{g} * 64 + lane] = (lutopt{n}_v4){{a[{m[0]}], a[{m[1]}], a[{m[2]}], a[{m[3]}]}};")
        emit("}")
        emit(f"static __device__ __forceinline__ void {fn}_step(uint32_t (&a)[{n}], uint32_t (&b)[{n}], uint32_t (&cnt)[{logn}], lutopt{n}_v4 *park, unsigned lane)")
    else:
        emit(f"static __device__ __forceinline__ void lutopt{n}_step(const uint32_t (&a)[{n}], uint32_t (&b)[{n}], uint32_t (&cnt)[{logn}])")
    emit("{")
    for position, r in enumerate(order):
        for g in reload_at.get(position, []):
            m = groups[g]
            emit(f"  {{ const lutopt{n}_v4 pk = park[{g} * 64 + lane]; a[{m[0]}] = pk.x; a[{m[1]}] = pk.y; a[{m[2]}] = pk.z; a[{m[3]}] = pk.w; }}")
        t = taps[r]
        assert 1 <= len(t) <= 8
        # XOR the taps three at a time (V_BITOP3 0x96), then pairs
        terms = [f"a[{c}]" for c in t]
        while len(terms) > 1:
            if len(terms) >= 3:
                x = terms[:3]
                terms = terms[3:]
                if not terms:
                    emit(f"  b[{r}] = __builtin_amdgcn_bitop3_b32({x[0]}, {x[1]}, {x[2]}, 0x96);")
                    terms = [f"b[{r}]"]
                    nops += 1
                    break
                v = tmp()
                emit(f"  const uint32_t {v} = __builtin_amdgcn_bitop3_b32({x[0]}, {x[1]}, {x[2]}, 0x96);")
                nops += 1
                terms.insert(0, v)
            else:
                emit(f"  b[{r}] = {terms[0]} ^ {terms[1]};")
                nops += 1
                terms = [f"b[{r}]"]
        if terms[0] != f"b[{r}]":
            emit(f"  b[{r}] = {terms[0]};")
        inv = bin(r).count("1") & 1     # weight -1 positions enter the counter complemented
        push(0, (f"b[{r}]", inv))
        for g in store_at.get(position, []):
            m = groups[g]
            emit(f"  park[{g} * 64 + lane] = (lutopt{n}_v4){{b[{m[0]}], b[{m[1]}], b[{m[2]}], b[{m[3]}]}};")
    # ---- finish the counter: ripple the leftovers up ------------------------------------
    for lev in range(nlev):
        while len(levels[lev]) > 1:
            assert len(levels[lev]) == 2
            (a, ia), (b, ib) = levels[lev]
            levels[lev] = []
            s = tmp()
            emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x ^ y, (ia, ib, ib)):02x});")
            nops += 1
            if lev + 1 < nlev:
                cy = tmp()
                emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x & y, (ia, ib, ib)):02x});")
                nops += 1
                push(lev + 1, (cy, 0))
            levels[lev].append((s, 0))
        (x, ix), = levels[lev]
        assert ix == 0
        if lev == nlev - 1:
            emit(f"  cnt[{lev}] = ~{x};   // (T - n/2) mod n: flip the top output bit")
            nops += 1
        else:
            emit(f"  cnt[{lev}] = {x};")
    emit("}")
    emit(f"// {nops} VALU ops per step for 32 samples per lane")
    if not npark:
        # packed taps for the host-side identity check
        emit(f"static const uint16_t LUTOPT{n}_NTAPS[{n}] = {{{','.join(str(len(t)) for t in taps)}}};")
        emit(f"static const uint16_t LUTOPT{n}_TAPS[{sum(len(t) for t in taps)}] = {{{flat}}};")
    return "\n".join(out) + "\n", nops


def main():
    n, taps = load(sys.argv[1])
    npark = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lookahead = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    text, nops = generate(n, taps, npark, lookahead)
    open(sys.argv[2], "w").write(text)
    print(f"n={n}: {nops} ops/step, {npark} parked groups -> {sys.argv[2]}")


if __name__ == "__main__":
    main()
