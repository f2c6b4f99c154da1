#!/usr/bin/env python3
"""Hand-scheduled gfx950 assembly for the LUTOPT-256 + CLT step (own register allocation).

Why: hipcc keeps the 256-plane state and its successor in 256 VGPRs + AGPR spills and emits ~310
v_accvgpr/v_mov per step on top of the 1002 arithmetic instructions; at one wave per SIMD every
instruction costs an issue slot.  Here the step is scheduled so that

  * H planes ("home") are updated IN PLACE: row h is evaluated after every other row that reads the
    old plane h, so new[h] can overwrite old[h] in its fixed VGPR -- no copy, no second buffer.
    The rows whose dependency graph is acyclic under that rule form H (149 planes for n256).
  * P planes ("parked") break the cycles: their new value is collected four at a time in a staging
    quad and written to lane-private LDS with ONE ds_write_b128; in the next step the quad is read
    back with ONE ds_read_b128 just before its first use.  4 planes per issue slot instead of 1.
  * the CLT carry-save counter consumes every new plane at birth (as in gen_lutopt_kernel.py).

The module produces an abstract instruction list (`Program`), can emulate it on random data against
the reference recurrence (CPU-only check), and prints gfx950 assembly text.
"""
import random
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
from gen_lutopt_kernel import load, tt3  # noqa: E402


# ------------------------------------------------------------------------------------------------
# 1. H / P split: largest set of rows that can be updated in place
# ------------------------------------------------------------------------------------------------
def split_home_parked(n, taps, tries=40):
    users = [[] for _ in range(n)]
    for r in range(n):
        for c in taps[r]:
            users[c].append(r)
    succ = [set(c for c in taps[r] if c != r) for r in range(n)]     # row u -> planes it reads
    pred = [set(u for u in users[h] if u != h) for h in range(n)]    # plane h <- rows reading it
    best = None
    for seed in range(tries):
        rnd = random.Random(seed)
        work = set(range(n))
        parked = []
        while work:
            progress = True
            while progress:
                progress = False
                for v in list(work):
                    i = sum(1 for u in pred[v] if u in work)
                    o = sum(1 for w in succ[v] if w in work)
                    if i == 0 or o == 0:
                        work.discard(v)
                        progress = True
            if not work:
                break
            v = max(work, key=lambda x: (sum(1 for u in pred[x] if u in work) * sum(1 for w in succ[x] if w in work), rnd.random()))
            work.discard(v)
            parked.append(v)
        if best is None or len(parked) < len(best):
            best = parked
    parked = sorted(best)
    while len(parked) % 4:                       # whole quads: move arbitrary home planes over
        for v in range(n):
            if v not in parked:
                parked.append(v)
                break
        parked.sort()
    home = [v for v in range(n) if v not in set(parked)]
    return home, parked, users


# ------------------------------------------------------------------------------------------------
# 2. schedule: order of rows + quads of parked planes
# ------------------------------------------------------------------------------------------------
def make_schedule(n, taps, home, parked, users, seed=0, quads=None, weights=(1.0, 1.5, 1.0, 2.0, 2.0)):
    """List scheduling.  An H row is ready when all OTHER readers of its old plane are done.  A P row
    is always ready; P rows are issued a whole quad at a time (their new values share one
    ds_write_b128).  The priority keeps few parked quads resident at a time: prefer rows that read
    planes of quads already loaded, that retire planes / free whole quads, and avoid rows that
    would pull in a new quad."""
    w_res, w_new, w_kill, w_free, w_h = weights
    rnd = random.Random(seed)
    H = set(home)
    if quads is None:
        ps = sorted(parked)
        quads = [ps[i:i + 4] for i in range(0, len(ps), 4)]
    quad_of = {p: qi for qi, q in enumerate(quads) for p in q}
    remaining_readers = [set(users[c]) for c in range(n)]
    resident = set()                       # quads whose old values are currently loaded
    alive_members = {qi: set(q) for qi, q in enumerate(quads)}
    done = set()
    order = []

    def ready_h(h):
        return all(u in done or u == h for u in users[h])

    def row_score(r, extra_rows=()):
        sc = 0.0
        newq = set()
        for c in taps[r]:
            if c in quad_of:
                qi = quad_of[c]
                if qi in resident:
                    sc += w_res
                else:
                    newq.add(qi)
            if remaining_readers[c] <= ({r} | set(extra_rows)):
                sc += w_kill
                if c in quad_of and alive_members[quad_of[c]] <= {c}:
                    sc += w_free
        return sc, newq

    pending_h = set(home)
    pending_q = set(range(len(quads)))
    while pending_h or pending_q:
        cands = []
        for h in pending_h:
            if ready_h(h):
                sc, newq = row_score(h)
                cands.append((w_h + sc - w_new * len(newq) + 0.01 * rnd.random(), ("H", h)))
        for qi in pending_q:
            rows = quads[qi]
            tot, newq = 0.0, set()
            for r in rows:
                sc, nq = row_score(r, rows)
                tot += sc
                newq |= nq
            unblocks = 0
            for r in rows:
                for c in taps[r]:
                    if c in pending_h and remaining_readers[c] - {c} <= set(rows):
                        unblocks += 1
            cands.append((0.25 * tot + 0.5 * unblocks - w_new * len(newq) + 0.01 * rnd.random(), ("Q", qi)))
        score, pick = max(cands)
        if pick[0] == "H":
            rows = [pick[1]]
            pending_h.discard(pick[1])
        else:
            rows = list(quads[pick[1]])
            pending_q.discard(pick[1])
        for r in rows:
            order.append(r)
            done.add(r)
            for c in taps[r]:
                if c in quad_of:
                    resident.add(quad_of[c])
                remaining_readers[c].discard(r)
                if not remaining_readers[c] and c in quad_of:
                    qi = quad_of[c]
                    alive_members[qi].discard(c)
                    if not alive_members[qi]:
                        resident.discard(qi)
    return order, quads


def refine_quads(n, taps, parked, order):
    """Regroup parked planes into quads by the position of their first reader in `order`."""
    pos = {r: i for i, r in enumerate(order)}
    first = {}
    for p in parked:
        first[p] = min(pos[r] for r in range(n) if p in taps[r])
    ps = sorted(parked, key=lambda p: first[p])
    return [ps[i:i + 4] for i in range(0, len(ps), 4)]


# ------------------------------------------------------------------------------------------------
# 3. program construction with register allocation
# ------------------------------------------------------------------------------------------------
class Program:
    """Flat instruction list.  Operands: ('v', i) VGPR, ('a', i) AGPR.  Ops:
       bitop3 d, a, b, c, tt | xor d, a, b | not d, a | ldsr quad_base_reg, slot | ldsw quad_base_reg, slot
       waitlgkm n | comment"""

    def __init__(self):
        self.ins = []

    def emit(self, *x):
        self.ins.append(x)


def build_step(n, taps, home, parked, users, order, quads, nvgpr_budget, first_free, lookahead=10, verbose=False):
    """Returns (Program, info).  VGPR map: homes first_free.. ; then quad pool; then scalar temps.
    info: home_reg{plane}, quad slot of plane, cnt regs, max regs used, op count."""
    logn = n.bit_length() - 1
    H = list(home)
    home_reg = {p: first_free + i for i, p in enumerate(H)}
    next_reg = first_free + len(H)
    quad_of = {p: qi for qi, q in enumerate(quads) for p in q}
    lane_in_quad = {p: q.index(p) for q in quads for p in q}
    pos = {r: i for i, r in enumerate(order)}
    readers_pos = {c: sorted(pos[r] for r in users[c]) for c in range(n)}
    # quad reload position: first reader of any member, minus lookahead (clamped), and
    # store position: after its last row
    q_first = [min(readers_pos[p][0] for p in q) for q in quads]
    q_store = [max(pos[p] for p in q) for q in quads]
    q_reload = [max(0, f - lookahead) for f in q_first]
    # one LDS slot per quad: the old values must be read before the new ones are written.  Where the
    # new quad is complete before its old values are needed, read early (costs residency)
    bad = [qi for qi in range(len(quads)) if q_reload[qi] > q_store[qi]]
    for qi in bad:
        q_reload[qi] = q_store[qi]
    # --- register pools -------------------------------------------------------------------
    nquad_regs = 0
    free_quads = []            # base registers of free quad slots
    quad_pool_base = next_reg
    free_single = []
    single_base = [None]

    prog = Program()
    stats = {"valu": 0, "lds": 0, "wait": 0, "max_quads": 0, "max_single": 0}

    def alloc_quad():
        nonlocal nquad_regs
        if free_quads:
            return free_quads.pop()
        b = quad_pool_base + nquad_regs
        nquad_regs += 4
        return b

    singles_in_use = set()
    single_next = [0]

    def alloc_single():
        if free_single:
            r = free_single.pop()
        else:
            r = ("S", single_next[0])          # symbolic, fixed up after the quad pool size is known
            single_next[0] += 1
        singles_in_use.add(r)
        stats["max_single"] = max(stats["max_single"], len(singles_in_use))
        return r

    def free_reg(r):
        if isinstance(r, tuple) and r[0] == "S":
            singles_in_use.discard(r)
            free_single.append(r)

    # where does each OLD plane currently live (register), for parked planes only after reload
    old_loc = {p: home_reg[p] for p in H}
    resident_quads = {}        # qi -> (base, set(alive members))
    outstanding = []           # LDS ops in flight, in issue order: ('r'|'w', qi, base)
    staging = {}               # qi -> (base, members computed so far)
    reads_left = {c: len(users[c]) for c in range(n)}

    def wait_for(kind, qi):
        """make sure the LDS op (kind, qi) has completed: s_waitcnt lgkmcnt(k)"""
        for idx, o in enumerate(outstanding):
            if o[0] == kind and o[1] == qi:
                k = len(outstanding) - 1 - idx
                prog.emit("waitlgkm", k)
                stats["wait"] += 1
                done_ops = outstanding[:idx + 1]
                del outstanding[:idx + 1]
                for d in done_ops:
                    if d[0] == "w":           # staging quad of a completed write can be reused
                        free_quads.append(d[2])
                return

    # --- vertical counter ------------------------------------------------------------------
    nlev = logn
    levels = [[] for _ in range(nlev + 2)]     # items: (reg, inverted, is_temp)

    def push(level, item):
        if level >= nlev:
            if item[2]:
                free_reg(item[0])
            return
        levels[level].append(item)
        while len(levels[level]) >= 3:
            (a, ia, ta), (b, ib, tb), (c, ic, tc) = levels[level][:3]
            del levels[level][:3]
            s = alloc_single()
            prog.emit("bitop3", s, a, b, c, tt3(lambda x, y, z: x ^ y ^ z, (ia, ib, ic)))
            stats["valu"] += 1
            cy = None
            if level + 1 < nlev:
                cy = alloc_single()
                prog.emit("bitop3", cy, a, b, c, tt3(lambda x, y, z: (x & y) | (x & z) | (y & z), (ia, ib, ic)))
                stats["valu"] += 1
            for reg, istemp in ((a, ta), (b, tb), (c, tc)):
                if istemp:
                    free_reg(reg)
            if cy is not None:
                push(level + 1, (cy, 0, True))
            levels[level].append((s, 0, True))

    # --- main pass ---------------------------------------------------------------------------
    for position, r in enumerate(order):
        # reloads due now
        for qi in range(len(quads)):
            if q_reload[qi] == position:
                base = alloc_quad()
                prog.emit("ldsr", base, qi)
                stats["lds"] += 1
                outstanding.append(("r", qi, base))
                resident_quads[qi] = [base, set(quads[qi]), False]
                for p in quads[qi]:
                    old_loc[p] = base + lane_in_quad[p]
        stats["max_quads"] = max(stats["max_quads"], nquad_regs // 4 - len(free_quads))
        # operands
        ops = []
        for c in taps[r]:
            if c in quad_of and c not in home_reg:
                qi = quad_of[c]
                if not resident_quads[qi][2]:
                    wait_for("r", qi)
                    resident_quads[qi][2] = True
            ops.append(old_loc[c])
        # destination
        if r in home_reg:
            dst = home_reg[r]
        else:
            qi = quad_of[r]
            if qi not in staging:
                staging[qi] = [alloc_quad(), 0]
            dst = staging[qi][0] + lane_in_quad[r]
            staging[qi][1] += 1
        # put the row's own old plane (if it is a tap) among the first three operands
        if r in taps[r] and r in home_reg:
            me = home_reg[r]
            ops.remove(me)
            ops.insert(0, me)
        cur = None
        rest = ops[:]
        while rest:
            if cur is None:
                if len(rest) >= 3:
                    prog.emit("bitop3", dst, rest[0], rest[1], rest[2], 0x96)
                    rest = rest[3:]
                elif len(rest) == 2:
                    prog.emit("xor", dst, rest[0], rest[1])
                    rest = rest[2:]
                else:
                    prog.emit("mov", dst, rest[0])
                    rest = rest[1:]
                cur = dst
            else:
                if len(rest) >= 2:
                    prog.emit("bitop3", dst, cur, rest[0], rest[1], 0x96)
                    rest = rest[2:]
                else:
                    prog.emit("xor", dst, cur, rest[0])
                    rest = rest[1:]
            stats["valu"] += 1
        # retire old planes
        for c in taps[r]:
            reads_left[c] -= 1
            if reads_left[c] == 0 and c in quad_of and c not in home_reg:
                qi = quad_of[c]
                resident_quads[qi][1].discard(c)
                if not resident_quads[qi][1]:
                    free_quads.append(resident_quads[qi][0])
                    del resident_quads[qi]
        # counter consumes the new plane
        push(0, (dst, bin(r).count("1") & 1, False))
        # store a completed staging quad
        if r not in home_reg:
            qi = quad_of[r]
            if staging[qi][1] == 4:
                # the slot must have been read already in this step
                assert q_reload[qi] <= position, ("slot reuse", qi)
                if any(o[0] == "r" and o[1] == qi for o in outstanding):
                    wait_for("r", qi)
                    if qi in resident_quads:
                        resident_quads[qi][2] = True
                prog.emit("ldsw", staging[qi][0], qi)
                stats["lds"] += 1
                outstanding.append(("w", qi, staging[qi][0]))
                del staging[qi]
    # finish the counter
    cnt = []
    for lev in range(nlev):
        while len(levels[lev]) > 1:
            assert len(levels[lev]) == 2
            (a, ia, ta), (b, ib, tb) = levels[lev]
            levels[lev] = []
            s = alloc_single()
            prog.emit("bitop3", s, a, b, b, tt3(lambda x, y, z: x ^ y, (ia, ib, ib)))
            stats["valu"] += 1
            cy = None
            if lev + 1 < nlev:
                cy = alloc_single()
                prog.emit("bitop3", cy, a, b, b, tt3(lambda x, y, z: x & y, (ia, ib, ib)))
                stats["valu"] += 1
            for reg, istemp in ((a, ta), (b, tb)):
                if istemp:
                    free_reg(reg)
            if cy is not None:
                push(lev + 1, (cy, 0, True))
            levels[lev].append((s, 0, True))
        (x, ix, tx), = levels[lev]
        assert ix == 0
        if lev == nlev - 1:
            prog.emit("not", x, x)            # (T - n/2) mod n: flip the top output bit
            stats["valu"] += 1
        cnt.append(x)
    # all writes must land before the next step's reads of the same slots (in-order LDS queue
    # guarantees order; the staging registers are released by later waits or here)
    prog.emit("waitlgkm", 0)
    stats["wait"] += 1
    for o in outstanding:
        if o[0] == "w":
            free_quads.append(o[2])
    outstanding.clear()
    # fix up symbolic single registers
    sbase = quad_pool_base + nquad_regs
    nsingle = single_next[0]

    def fix(x):
        return sbase + x[1] if isinstance(x, tuple) and x[0] == "S" else x

    fixed = Program()
    for ins in prog.ins:
        fixed.emit(*[fix(x) for x in ins])
    cnt = [fix(x) for x in cnt]
    info = {"home_reg": home_reg, "quad_of": quad_of, "lane_in_quad": lane_in_quad, "quads": quads, "cnt": cnt,
            "vgpr_end": sbase + nsingle, "nquads_pool": nquad_regs // 4, "nsingle": nsingle, "stats": stats,
            "bad_quads": bad, "quad_pool_base": quad_pool_base}
    return fixed, info


# ------------------------------------------------------------------------------------------------
# 4. emulator (one lane; LDS slots are lane-private)
# ------------------------------------------------------------------------------------------------
def emulate_step(prog, info, n, taps, seed=0, steps=3):
    rnd = random.Random(seed)
    M = 0xFFFFFFFF
    regs = {}
    lds = {}
    state = [rnd.getrandbits(32) for _ in range(n)]
    for p, r in info["home_reg"].items():
        regs[r] = state[p]
    for qi, q in enumerate(info["quads"]):
        lds[qi] = [state[p] for p in q]

    def bitop3(a, b, c, tt):
        r = 0
        for i in range(8):
            if (tt >> i) & 1:
                m = M
                m &= a if i & 4 else ~a & M
                m &= b if i & 2 else ~b & M
                m &= c if i & 1 else ~c & M
                r |= m
        return r

    for _ in range(steps):
        pending = []                      # LDS reads not yet waited for: (base, values)
        for ins in prog.ins:
            op = ins[0]
            if op == "bitop3":
                regs[ins[1]] = bitop3(regs[ins[2]], regs[ins[3]], regs[ins[4]], ins[5])
            elif op == "xor":
                regs[ins[1]] = regs[ins[2]] ^ regs[ins[3]]
            elif op == "mov":
                regs[ins[1]] = regs[ins[2]]
            elif op == "not":
                regs[ins[1]] = ~regs[ins[2]] & M
            elif op == "ldsr":
                # data arrives only at the wait: poison until then
                pending.append(["r", ins[1], list(lds[ins[2]])])
                for i in range(4):
                    regs[ins[1] + i] = None
            elif op == "ldsw":
                vals = [regs[ins[1] + i] for i in range(4)]
                assert None not in vals
                pending.append(["w", ins[2], vals])
                lds[ins[2]] = vals        # in-order queue: later reads see it
            elif op == "waitlgkm":
                k = ins[1]
                while len(pending) > k:
                    o = pending.pop(0)
                    if o[0] == "r":
                        for i in range(4):
                            regs[o[1] + i] = o[2][i]
            else:
                raise ValueError(op)
            if op in ("bitop3", "xor", "mov", "not"):
                assert all(regs[x] is not None for x in ins[2:5] if isinstance(x, int) and op != "bitop3" or True)
        # reference
        new = []
        for r in range(n):
            v = 0
            for c in taps[r]:
                v ^= state[c]
            new.append(v)
        state = new
        for p, r in info["home_reg"].items():
            assert regs[r] == state[p], ("home", p)
        for qi, q in enumerate(info["quads"]):
            assert lds[qi] == [state[p] for p in q], ("parked", qi)
        # counter planes
        for g in range(32):
            T = 0
            for i in range(n):
                bit = (state[i] >> g) & 1
                T += bit if bin(i).count("1") % 2 == 0 else 1 - bit
            want = (T ^ (n >> 1)) & (n - 1)
            got = sum(((regs[c] >> g) & 1) << q for q, c in enumerate(info["cnt"]))
            assert got == want, (g, got, want)
    return True


def search(n, taps, tries, verbose=True):
    home, parked, users = split_home_parked(n, taps)
    best = None
    rnd = random.Random(12345)
    for t in range(tries):
        weights = (rnd.uniform(0.3, 2.0), rnd.uniform(0.5, 4.0), rnd.uniform(0.3, 2.0), rnd.uniform(0.5, 4.0), rnd.uniform(0.0, 3.0))
        order, quads = make_schedule(n, taps, home, parked, users, t, None, weights)
        for it in range(3):
            quads = refine_quads(n, taps, parked, order)
            order, quads = make_schedule(n, taps, home, parked, users, t, quads, weights)
        for la in (4, 10):
            prog, info = build_step(n, taps, home, parked, users, order, quads, 256, 8, lookahead=la)
            key = info["vgpr_end"]
            if best is None or key < best[0]:
                best = (key, t, la, prog, info, weights)
                if verbose:
                    print("try", t, "la", la, "vgpr_end", key, "quads", info["nquads_pool"], "singles", info["nsingle"],
                          "lds", info["stats"]["lds"], "waits", info["stats"]["wait"], [round(w, 2) for w in weights], flush=True)
    return best, (home, parked, users)


def main():
    n, taps = load(sys.argv[1])
    tries = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    best, _ = search(n, taps, tries)
    key, seed, la, prog, info, weights = best
    print("best: vgpr_end", info["vgpr_end"], "quads in pool", info["nquads_pool"], "singles", info["nsingle"],
          info["stats"], "early-reload quads", len(info["bad_quads"]))
    emulate_step(prog, info, n, taps)
    print("emulation ok; instructions per step:", len(prog.ins))


if __name__ == "__main__":
    main()
