#!/usr/bin/env python3
"""Emit the straight-line bit-sliced LUTOPT step + CLT vertical counter for one matrix.

Input : a recurrence matrix in the reference's text format (software/rnghunt/matrices/N,
        shipped as packed tap lists in basebandboard_amd/data/lutopt_N.taps).
Output: a C++ include for the HIP kernels with

  lutoptN_step(a, b, cnt)   b = A*a over GF(2) on 32 generators per lane
                            (gateware/bbb/rng.py:38-40: row r = XOR of its taps, all rows
                            from the OLD state), and cnt[0..7] = the bit planes of the int8
                            CLTGRNG sample of the OLD state a (rng.py:96-108).
  lutoptN_advance(a, b)     b = A*a only.
  lutoptN_step_new(a,b,cnt) b = A*a and cnt = sample of the NEW state b.
  lutopt256_step_parked(a, pa, b, pb, cnt)   lutopt256_step with a chosen set of planes travelling in AGPRs.

The counter reads the OLD state so that it can share work with the update: a first-level full
adder over y_a, y_b, y_c has the sum output x_a ^ x_b ^ x_c (possibly complemented) -- exactly the
partial XOR of a row whose taps include a, b, c.  A set of disjoint triples, each inside the tap
list of a distinct row, is chosen (`shared_triples`); for those the row's V_BITOP3(0x96) IS the
adder's sum and only the carry costs an extra instruction.

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


def shared_triples(n, taps, tries=40, seed=1):
    """Disjoint state-bit triples, each a subset of the taps of a distinct row: {row: triple}.
    Greedy, most constrained first (smallest number of still-possible triples over its three bits),
    random tie-breaking, best of `tries` (84 of the 85 possible for n256)."""
    import itertools
    import random
    cands = [(r, tr) for r in range(n) for tr in itertools.combinations(taps[r], 3)]
    by_bit = [[] for _ in range(n)]
    for i, (_, tr) in enumerate(cands):
        for b in tr:
            by_bit[b].append(i)
    best = []
    for t in range(tries):
        rnd = random.Random(seed * 1000003 + t)
        alive = [True] * len(cands)
        deg = [len(x) for x in by_bit]
        m = []
        while True:
            pick, pick_key = None, None
            for i, (_, tr) in enumerate(cands):
                if alive[i]:
                    key = deg[tr[0]] + deg[tr[1]] + deg[tr[2]] + 3 * rnd.random()
                    if pick_key is None or key < pick_key:
                        pick, pick_key = i, key
            if pick is None:
                break
            r, tr = cands[pick]
            m.append((r, tr))
            for i, (r2, tr2) in enumerate(cands):
                if alive[i] and (r2 == r or set(tr2) & set(tr)):
                    alive[i] = False
                    for b in tr2:
                        deg[b] -= 1
        if len(m) > len(best):
            best = m
    return {r: tr for r, tr in best}


def parking_set(n, taps, order, budget):
    """Planes to keep in AGPRs between their birth and their first reader in the next step, so
    that at most `budget` planes are VGPR-resident at any row position.  Model of one steady-state
    step in row order: a plane that is not parked is resident from its birth to the end of the
    step (new) and from the start to its last reader (old); a parked plane only from its first to
    its last reader.  Greedy: park the plane that removes the most excess."""
    pos = {r: i for i, r in enumerate(order)}
    rd = [[] for _ in range(n)]
    for r in range(n):
        for c in taps[r]:
            rd[c].append(pos[r])
    first = [min(x) for x in rd]
    last = [max(x) for x in rd]
    born = [pos[p] for p in range(n)]

    def resident(p, t, parked):
        if parked:
            return 1 if first[p] <= t <= last[p] else 0
        return (1 if last[p] >= t else 0) + (1 if born[p] <= t else 0)

    parked = set()
    while True:
        prof = [sum(resident(p, t, p in parked) for p in range(n)) for t in range(n)]
        exc = [max(0, x - budget) for x in prof]
        if not any(exc):
            return parked
        best, gain = None, 0
        for p in range(n):
            if p in parked:
                continue
            g = sum(resident(p, t, False) - resident(p, t, True) for t in range(n) if exc[t])
            if g > gain:
                best, gain = p, g
        if best is None:
            return parked
        parked.add(best)


class StepEmitter:
    """Emits one straight-line function body.  mode: 'advance' (update only), 'step' (update +
    sample of the old state).  parked: planes whose old value arrives in pa[] (AGPR) and whose new
    value leaves in pb[] (AGPR) through BBB_ACC_READ / BBB_ACC_WRITE."""

    def __init__(self, n, taps, order, share, mode, parked=()):
        self.n, self.taps, self.order, self.share = n, taps, order, share
        self.mode, self.parked = mode, set(parked)
        self.logn = n.bit_length() - 1
        self.out = []
        self.nops = 0
        self.nmoves = 0
        self.tmp_id = 0
        self.levels = [[] for _ in range(self.logn + 2)]
        self.inv_of = [bin(c).count("1") & 1 for c in range(n)]   # weight -1 positions enter complemented
        self.loaded = {}                                          # parked plane -> expression of its VGPR copy

    def emit(self, s):
        self.out.append(s)

    def tmp(self):
        self.tmp_id += 1
        return f"t{self.tmp_id}"

    def old(self, c):
        if c not in self.parked:
            return f"a[{c}]"
        if c not in self.loaded:
            v = self.tmp()
            self.emit(f"  uint32_t {v}; BBB_ACC_READ({v}, pa[{c}]);")
            self.nmoves += 1
            self.loaded[c] = v
        return self.loaded[c]

    def push(self, level, item):
        nlev = self.logn
        if level >= nlev:
            return       # carries out of the top output bit are never needed
        lv = self.levels[level]
        lv.append(item)
        while len(lv) >= 3:
            (a, ia), (b, ib), (c, ic) = lv[:3]
            del lv[:3]
            s = self.tmp()
            self.emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: x ^ y ^ z, (ia, ib, ic)):02x});")
            self.nops += 1
            if level + 1 < nlev:
                cy = self.tmp()
                self.emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: (x & y) | (x & z) | (y & z), (ia, ib, ic)):02x});")
                self.nops += 1
                self.push(level + 1, (cy, 0))
            lv.append((s, 0))

    def assign_new(self, r, terms):
        """new plane r = XOR of `terms` (expressions), three at a time"""
        dst = f"b[{r}]"
        if r in self.parked:
            dst = self.tmp()
        decl = "const uint32_t " if r in self.parked else ""
        while len(terms) > 3 or (len(terms) == 3 and False):
            x, terms = terms[:3], terms[3:]
            v = self.tmp()
            self.emit(f"  const uint32_t {v} = __builtin_amdgcn_bitop3_b32({x[0]}, {x[1]}, {x[2]}, 0x96);")
            self.nops += 1
            terms.insert(0, v)
        if len(terms) == 3:
            self.emit(f"  {decl}{dst} = __builtin_amdgcn_bitop3_b32({terms[0]}, {terms[1]}, {terms[2]}, 0x96);")
            self.nops += 1
        elif len(terms) == 2:
            self.emit(f"  {decl}{dst} = {terms[0]} ^ {terms[1]};")
            self.nops += 1
        else:
            self.emit(f"  {decl}{dst} = {terms[0]};")
        if r in self.parked:
            self.emit(f"  BBB_ACC_WRITE(pb[{r}], {dst});")
            self.nmoves += 1

    def body(self):
        n, taps, share = self.n, self.taps, self.share
        nlev = self.logn
        counting = self.mode == "step"
        covered = set(c for tr in share.values() for c in tr) if counting else set()
        pushed = set()
        for r in self.order:
            t = taps[r]
            assert 1 <= len(t) <= 8
            if self.mode == "step_new":
                self.assign_new(r, [self.old(c) for c in t])
                self.push(0, (f"b[{r}]", self.inv_of[r]))
                continue
            if counting and r in share:
                tr = share[r]
                iv = tuple(self.inv_of[c] for c in tr)
                x = [self.old(c) for c in tr]
                v = self.tmp()
                self.emit(f"  const uint32_t {v} = __builtin_amdgcn_bitop3_b32({x[0]}, {x[1]}, {x[2]}, 0x96);   // row {r} partial = adder sum")
                self.nops += 1
                cy = None
                if nlev > 1:
                    cy = self.tmp()
                    self.emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({x[0]}, {x[1]}, {x[2]}, 0x{tt3(lambda p, q, z: (p & q) | (p & z) | (q & z), iv):02x});")
                    self.nops += 1
                self.assign_new(r, [v] + [self.old(c) for c in t if c not in tr])
                if cy is not None:
                    self.push(1, (cy, 0))
                self.push(0, (v, iv[0] ^ iv[1] ^ iv[2]))
            else:
                self.assign_new(r, [self.old(c) for c in t])
            if counting:
                for c in t:              # state bits outside every shared triple enter the counter singly
                    if c not in covered and c not in pushed:
                        pushed.add(c)
                        self.push(0, (self.old(c), self.inv_of[c]))
        if self.mode == "advance":
            return
        if counting:
            assert len(pushed) + len(covered) == n, "a state bit that no row reads is not summed"
        # ---- finish the counter: ripple the leftovers up ------------------------------------
        for lev in range(nlev):
            lv = self.levels[lev]
            while len(lv) > 1:
                assert len(lv) == 2
                (a, ia), (b, ib) = lv
                del lv[:]
                s = self.tmp()
                self.emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x ^ y, (ia, ib, ib)):02x});")
                self.nops += 1
                if lev + 1 < nlev:
                    cy = self.tmp()
                    self.emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x & y, (ia, ib, ib)):02x});")
                    self.nops += 1
                    self.push(lev + 1, (cy, 0))
                lv.append((s, 0))
            (x, ix), = lv
            assert ix == 0
            if lev == nlev - 1:
                self.emit(f"  cnt[{lev}] = ~{x};   // (T - n/2) mod n: flip the top output bit")
                self.nops += 1
            else:
                self.emit(f"  cnt[{lev}] = {x};")


class Packed512Emitter:
    """n = 512 on a 256-register state: register p holds plane p of 16 generators in its low half and plane
    256 + (p ^ 1) of the same 16 generators in its high half (p ^ 1: both planes of a register then carry the same sign in
    the adder tree, popcount(256 + (p ^ 1)) = popcount(p) mod 2, so the counter's input complement folds into its truth
    tables exactly as for n = 256).  New register p = XOR over tap PAIRS: the i-th tap of row p and the i-th tap of row
    256 + (p ^ 1) are brought into one word by ONE V_PERM_B32 (either half of either source register into either half
    of the result; a missing tap is the zero byte selector), then V_BITOP3 / V_XOR as before.  The counter runs on the
    NEW state: 9 planes per half (0..256), the kernel adds the halves."""

    def __init__(self, taps, parked=()):
        """parked: packed registers whose old value arrives in pa[] (AGPR) and whose new value leaves in pb[] (as in
        StepEmitter: BBB_ACC_READ at the first reader, BBB_ACC_WRITE at birth)"""
        assert len(taps) == 512
        self.taps = taps
        self.out, self.nops, self.tmp_id = [], 0, 0
        self.levels = [[] for _ in range(10)]
        self.nlev = 9
        self.parked, self.loaded, self.nmoves = set(parked), {}, 0

    def old(self, r):
        if r not in self.parked:
            return f"a[{r}]"
        if r not in self.loaded:
            v = self.tmp()
            self.emit(f"  uint32_t {v}; BBB_ACC_READ({v}, pa[{r}]);")
            self.nmoves += 1
            self.loaded[r] = v
        return self.loaded[r]

    def emit(self, s):
        self.out.append(s)

    def tmp(self):
        self.tmp_id += 1
        return f"t{self.tmp_id}"

    @staticmethod
    def where(c):
        """plane c -> (register, first byte of its half)"""
        return (c, 0) if c < 256 else ((c - 256) ^ 1, 2)

    def reg_taps(self):
        """registers read by new register p (for the row order)"""
        return [sorted({self.where(c)[0] for c in self.taps[p] + self.taps[256 + (p ^ 1)]}) for p in range(256)]

    def push(self, level, item):
        if level >= self.nlev:
            return
        lv = self.levels[level]
        lv.append(item)
        while len(lv) >= 3:
            (a, ia), (b, ib), (c, ic) = lv[:3]
            del lv[:3]
            s = self.tmp()
            self.emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: x ^ y ^ z, (ia, ib, ic)):02x});")
            self.nops += 1
            if level + 1 < self.nlev:
                cy = self.tmp()
                self.emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{tt3(lambda x, y, z: (x & y) | (x & z) | (y & z), (ia, ib, ic)):02x});")
                self.nops += 1
                self.push(level + 1, (cy, 0))
            lv.append((s, 0))

    def body(self, order):
        for p in order:
            lo, hi = self.taps[p], self.taps[256 + (p ^ 1)]
            terms = []
            for i in range(max(len(lo), len(hi))):
                x = self.where(lo[i]) if i < len(lo) else None
                y = self.where(hi[i]) if i < len(hi) else None
                xb = [x[1], x[1] + 1] if x else [0x0c, 0x0c]
                yb = [4 + y[1], 5 + y[1]] if y else [0x0c, 0x0c]
                sel = xb[0] | xb[1] << 8 | yb[0] << 16 | yb[1] << 24
                xr = self.old(x[0]) if x else "0u"
                yr = self.old(y[0]) if y else "0u"
                if x and y and x[0] == y[0] and sel == 0x07060100:
                    terms.append(xr)                      # both halves already in place
                    continue
                v = self.tmp()
                self.emit(f"  const uint32_t {v} = __builtin_amdgcn_perm({yr}, {xr}, 0x{sel:08x}u);")
                self.nops += 1
                terms.append(v)
            while len(terms) > 3:
                x3, terms = terms[:3], terms[3:]
                v = self.tmp()
                self.emit(f"  const uint32_t {v} = __builtin_amdgcn_bitop3_b32({x3[0]}, {x3[1]}, {x3[2]}, 0x96);")
                self.nops += 1
                terms.insert(0, v)
            dst, decl = f"b[{p}]", ""
            if p in self.parked:
                dst, decl = self.tmp(), "const uint32_t "
            if len(terms) == 3:
                self.emit(f"  {decl}{dst} = __builtin_amdgcn_bitop3_b32({terms[0]}, {terms[1]}, {terms[2]}, 0x96);")
            elif len(terms) == 2:
                self.emit(f"  {decl}{dst} = {terms[0]} ^ {terms[1]};")
            else:
                self.emit(f"  {decl}{dst} = {terms[0]};")
            self.nops += 1
            if p in self.parked:
                self.emit(f"  BBB_ACC_WRITE(pb[{p}], {dst});")
                self.nmoves += 1
            self.push(0, (dst, bin(p).count("1") & 1))
        for lev in range(self.nlev):
            lv = self.levels[lev]
            while len(lv) > 1:
                assert len(lv) == 2
                (a, ia), (b, ib) = lv
                del lv[:]
                s = self.tmp()
                self.emit(f"  const uint32_t {s} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x ^ y, (ia, ib, ib)):02x});")
                self.nops += 1
                if lev + 1 < self.nlev:
                    cy = self.tmp()
                    self.emit(f"  const uint32_t {cy} = __builtin_amdgcn_bitop3_b32({a}, {b}, {b}, 0x{tt3(lambda x, y, z: x & y, (ia, ib, ib)):02x});")
                    self.nops += 1
                    self.push(lev + 1, (cy, 0))
                lv.append((s, 0))
            (x, ix), = lv
            assert ix == 0
            self.emit(f"  cnt[{lev}] = {x};")


def generate_packed512(taps):
    """The n = 512 form: see Packed512Emitter.  cnt[q], q = 0..8: bit q of the number of +1 terms among planes 0..255
    (low half) and among planes 256..511 (high half) of the NEW state, per generator."""
    out = []
    emit = out.append
    emit("// GENERATED by basebandboard_amd/gen_lutopt_kernel.py from lutopt_512.taps -- do not edit.")
    flat = ",".join(",".join(map(str, t)) for t in taps)
    emit(f"#define LUTOPT512_TAPS_CRC 0x{zlib.crc32(flat.encode()) & 0xffffffff:08x}u")
    e = Packed512Emitter(taps)
    order = row_order(256, e.reg_taps())
    e.body(order)
    emit("// packed state: register p = plane p (bits 0..15: generators 0..15) | plane 256 + (p ^ 1) (bits 16..31)")
    emit("static __device__ __forceinline__ void lutopt512p_step_new(const uint32_t (&a)[256], uint32_t (&b)[256], uint32_t (&cnt)[9])")
    emit("{")
    out.extend(e.out)
    emit("}")
    emit(f"// {e.nops} VALU ops per step for 16 samples per lane")
    # the same step with the registers of LUTOPT512_PARKED travelling in AGPRs (pa / pb), placed by the generator instead of
    # hipcc's spilling: at most PACKED512_BUDGET of the 256 state registers are VGPR-resident at any row
    rt = e.reg_taps()
    parked = sorted(parking_set(256, rt, order, PACKED512_BUDGET))
    ep = Packed512Emitter(taps, parked)
    ep.body(order)
    emit("#ifndef BBB_ACC_WRITE   // (a host build of this text defines them as plain assignments)")
    emit('#define BBB_ACC_WRITE(dst, src) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(src))')
    emit('#define BBB_ACC_READ(dst, src) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src))')
    emit("#endif")
    emit(f"#define LUTOPT512_NPARKED {len(parked)}")
    emit(f"static const uint16_t LUTOPT512_PARKED[{len(parked)}] = {{{','.join(map(str, parked))}}};")
    emit("#define LUTOPT512_FOR_PARKED(F) " + " ".join(f"F({q})" for q in parked))
    words = [sum(1 << (q & 31) for q in parked if q >> 5 == w) for w in range(8)]
    emit(f"static constexpr uint32_t LUTOPT512_PARKED_MASK[8] = {{{','.join(hex(w) + 'u' for w in words)}}};")
    emit("static constexpr bool lutopt512_is_parked(int p) { return (LUTOPT512_PARKED_MASK[p >> 5] >> (p & 31)) & 1u; }")
    emit("static __device__ __forceinline__ void lutopt512p_step_new_parked(const uint32_t (&a)[256], const uint32_t (&pa)[256], uint32_t (&b)[256], uint32_t (&pb)[256], uint32_t (&cnt)[9])")
    emit("{")
    out.extend(ep.out)
    emit("}")
    emit(f"// {ep.nops} VALU ops + {ep.nmoves} AGPR moves per step (at most {PACKED512_BUDGET} state registers in VGPRs)")
    emit(f"static const uint16_t LUTOPT512_NTAPS[512] = {{{','.join(str(len(t)) for t in taps)}}};")
    emit(f"static const uint16_t LUTOPT512_TAPS[{sum(len(t) for t in taps)}] = {{{flat}}};")
    return "\n".join(out) + "\n", e.nops


# (suffix, planes resident in VGPRs at any point) of the parked variants (n = 256 only).  180 is the
# measured optimum for the sample kernel: above it hipcc adds its own AGPR spills on top
# for the sample kernel with its plane -> byte round end.  The PLANES kernel (round 3: no LDS staging, no round end, hence far
# fewer other live values) takes 230: 71 parked planes instead of 121, and hipcc adds no spills of its own up to there
# (budget -> registers / VALU per step: 180 -> 436 / 1185, 200 -> 422 / 1120, 220 -> 432 / 1080, 230 -> 434 / 1060,
# 240 -> 460 / 1054: the guests beside the kernel need 72 of the 512)
BER_BUDGET = 222             # the fused BER kernel: the PLANES kernel's loop plus the comparators' live values (see ber_kernels_impl.hpp)
PARK_VARIANTS = (("", 180), ("_hi", 230), ("_ber", BER_BUDGET))
PACKED512_BUDGET = 230        # the packed n512 kernel: state registers resident in VGPRs (its counters and output stage need the rest)


def generate(n, taps):
    logn = n.bit_length() - 1
    assert 1 << logn == n and n >= 16
    out = []
    emit = out.append
    emit(f"// GENERATED by basebandboard_amd/gen_lutopt_kernel.py from lutopt_{n}.taps -- do not edit.")
    flat = ",".join(",".join(map(str, t)) for t in taps)
    emit(f"#define LUTOPT{n}_TAPS_CRC 0x{zlib.crc32(flat.encode()) & 0xffffffff:08x}u")
    order = row_order(n, taps)
    share = shared_triples(n, taps)

    e = StepEmitter(n, taps, order, share, "advance")
    e.body()
    emit(f"static __device__ __forceinline__ void lutopt{n}_advance(const uint32_t (&a)[{n}], uint32_t (&b)[{n}])")
    emit("{")
    out.extend(e.out)
    emit("}")
    emit(f"// {e.nops} VALU ops")

    e = StepEmitter(n, taps, order, share, "step")
    e.body()
    nops = e.nops
    emit(f"static __device__ __forceinline__ void lutopt{n}_step(const uint32_t (&a)[{n}], uint32_t (&b)[{n}], uint32_t (&cnt)[{logn}])")
    emit("{")
    out.extend(e.out)
    emit("}")
    emit(f"// {nops} VALU ops per step for 32 samples per lane")

    e = StepEmitter(n, taps, order, share, "step_new")
    e.body()
    emit(f"// b = A*a and cnt = sample of the NEW state b (no sharing with the update: {e.nops} ops); for kernels with many")
    emit("// other live values, where hipcc allocates this form better")
    emit(f"static __device__ __forceinline__ void lutopt{n}_step_new(const uint32_t (&a)[{n}], uint32_t (&b)[{n}], uint32_t (&cnt)[{logn}])")
    emit("{")
    out.extend(e.out)
    emit("}")

    if n == 256:
        emit("// Variants with explicit register-file placement: planes in LUTOPT256_PARKED* live in AGPRs (pa/pb) from")
        emit("// birth to their first reader of the next step; BBB_ACC_READ / BBB_ACC_WRITE are v_accvgpr moves.")
        emit("#ifndef BBB_ACC_WRITE   // (a host build of this text defines them as plain assignments)")
        emit('#define BBB_ACC_WRITE(dst, src) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(src))')
        emit('#define BBB_ACC_READ(dst, src) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src))')
        emit("#endif")
        for suffix, budget in PARK_VARIANTS:
            parked = sorted(parking_set(n, taps, order, budget))
            e = StepEmitter(n, taps, order, share, "step", parked)
            e.body()
            S = suffix.upper()
            emit(f"#define LUTOPT{n}_NPARKED{S} {len(parked)}")
            emit(f"static const uint16_t LUTOPT{n}_PARKED{S}[{len(parked)}] = {{{','.join(map(str, parked))}}};")
            emit(f"#define LUTOPT{n}_FOR_PARKED{S}(F) " + " ".join(f"F({p})" for p in parked))
            words = [sum(1 << (p & 31) for p in parked if p >> 5 == w) for w in range(n // 32)]
            emit(f"static constexpr uint32_t LUTOPT{n}_PARKED{S}_MASK[{n // 32}] = {{{','.join(hex(w) + 'u' for w in words)}}};")
            emit(f"static constexpr bool lutopt{n}{suffix}_is_parked(int p) {{ return (LUTOPT{n}_PARKED{S}_MASK[p >> 5] >> (p & 31)) & 1u; }}")
            emit(f"static __device__ __forceinline__ void lutopt{n}_step_parked{suffix}(const uint32_t (&a)[{n}], const uint32_t (&pa)[{n}], uint32_t (&b)[{n}], uint32_t (&pb)[{n}], uint32_t (&cnt)[{logn}])")
            emit("{")
            out.extend(e.out)
            emit("}")
            emit(f"// {e.nops} VALU ops + {e.nmoves} AGPR moves per step (at most {budget} planes in VGPRs)")
    # packed taps for the host-side identity check
    emit(f"static const uint16_t LUTOPT{n}_NTAPS[{n}] = {{{','.join(str(len(t)) for t in taps)}}};")
    emit(f"static const uint16_t LUTOPT{n}_TAPS[{sum(len(t) for t in taps)}] = {{{flat}}};")
    return "\n".join(out) + "\n", nops


def main():
    n, taps = load(sys.argv[1])
    text, nops = generate_packed512(taps) if n == 512 else generate(n, taps)
    open(sys.argv[2], "w").write(text)
    print(f"n={n}: {nops} ops/step -> {sys.argv[2]}")


if __name__ == "__main__":
    main()
