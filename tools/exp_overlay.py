#!/usr/bin/env python3
"""The laboratory's instrumentation lives OUTSIDE the product sources (round 5): time stamps inside kernels, suppressed stores,
alternative kernel arguments -- everything that used to sit in `#ifdef BBB_EXPERIMENTS` blocks of csrc/*.hip -- is kept as an
OVERLAY per file (experiments/overlays/<file>.json: a list of hunks, each the product lines it replaces with a few lines of
context, and the lines of the experiments build), applied to a copy of the product file when libbbb_hip_exp.so is built.

    exp_overlay.py apply <product file> <overlay.json> <out file>     (the Makefile: no overlay file -> plain copy)
    exp_overlay.py split <annotated file> <product out> <overlay out> (one-time: a file that still carries #ifdef BBB_EXPERIMENTS)
    exp_overlay.py make  <product file> <experiments file> <overlay out>   (after editing the experiments copy by hand)

A hunk must match the product text exactly once: a product edit that touches instrumented lines fails the experiments build
until its overlay follows (`make`), and never changes the product."""
import difflib
import json
import re
import sys


def strip_experiments(lines):
    """the text with every #if(def) ... BBB_EXPERIMENTS block resolved as 'not defined'"""
    out, stack = [], []          # stack of [is_exp_block, emitting_before, in_else]
    for l in lines:
        t = l.strip()
        if re.match(r"#\s*if", t):
            is_exp = bool(re.match(r"#\s*ifdef\s+BBB_EXPERIMENTS\b", t) or re.match(r"#\s*if\s+defined\(BBB_EXPERIMENTS\)", t))
            if re.match(r"#\s*ifndef\s+BBB_EXPERIMENTS\b", t):
                raise SystemExit("ifndef BBB_EXPERIMENTS is not handled")
            stack.append([is_exp, False])
            if is_exp:
                continue
        elif re.match(r"#\s*else\b", t) and stack and stack[-1][0]:
            stack[-1][1] = True
            continue
        elif re.match(r"#\s*endif\b", t) and stack:
            top = stack.pop()
            if top[0]:
                continue
        # emitting? every enclosing experiments block must be in its #else part
        if all((not s[0]) or s[1] for s in stack):
            out.append(l)
    return out


def make_overlay(prod, exp, ctx=3):
    sm = difflib.SequenceMatcher(a=prod, b=exp, autojunk=False)
    hunks = []
    for tag, i1, i2, j1, j2 in sm.get_opcodes():
        if tag == "equal":
            continue
        c = ctx
        while True:                      # enough context for the (context + old) block to be unique
            lo = max(0, i1 - c)
            hi = min(len(prod), i2 + c)
            block = prod[lo:hi]
            n = sum(1 for k in range(len(prod) - len(block) + 1) if prod[k:k + len(block)] == block)
            if n == 1 or (lo == 0 and hi == len(prod)):
                break
            c += 2
        hunks.append({"before": prod[lo:i1], "old": prod[i1:i2], "after": prod[i2:hi], "new": exp[j1:j2]})
    return hunks


def apply_overlay(prod, hunks):
    # every hunk is located in the PRODUCT text (its context may reach into a neighbouring hunk's lines), then the hunks are
    # spliced in from the last to the first
    places = []
    for h in hunks:
        block = h["before"] + h["old"] + h["after"]
        pos = [k for k in range(len(prod) - len(block) + 1) if prod[k:k + len(block)] == block]
        if len(pos) != 1:
            raise SystemExit(f"overlay hunk matches {len(pos)} times (expected once); first lines of its context:\n  "
                             + "\n  ".join((h["before"] + h["old"])[:4]))
        places.append((pos[0] + len(h["before"]), h))
    out = list(prod)
    for k, h in sorted(places, key=lambda x: -x[0]):
        out[k:k + len(h["old"])] = h["new"]
    return out


def main():
    cmd = sys.argv[1]
    if cmd == "apply":
        prod = open(sys.argv[2]).read().split("\n")
        try:
            hunks = json.load(open(sys.argv[3]))
        except FileNotFoundError:
            hunks = []
        open(sys.argv[4], "w").write("\n".join(apply_overlay(prod, hunks)))
    elif cmd == "split":
        exp = open(sys.argv[2]).read().split("\n")
        prod = strip_experiments(exp)
        hunks = make_overlay(prod, exp)
        assert apply_overlay(prod, hunks) == exp
        open(sys.argv[3], "w").write("\n".join(prod))
        json.dump(hunks, open(sys.argv[4], "w"), indent=1)
        print(f"{sys.argv[2]}: {len(exp)} -> {len(prod)} lines, {len(hunks)} hunks")
    elif cmd == "make":
        prod = open(sys.argv[2]).read().split("\n")
        exp = open(sys.argv[3]).read().split("\n")
        hunks = make_overlay(prod, exp)
        assert apply_overlay(prod, hunks) == exp
        json.dump(hunks, open(sys.argv[4], "w"), indent=1)
        print(f"{len(hunks)} hunks")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
