#!/usr/bin/env python3
"""Instruction histogram of a kernel's basic blocks from `hipcc -S` output (evidence for DESIGN.md's budgets).

usage: isa_hist.py <file.hip> <kernel-name-substring> [min block size]
Compiles csrc/<file.hip> to assembly for gfx950 and prints, for every basic block of the first kernel whose
mangled name contains the substring, its size and the most frequent opcodes; inline-asm instructions
(the generator's explicit v_accvgpr moves) are counted separately from hipcc's own.
ISA_HIST_FLAGS in the environment adds compiler flags (-DBBB_BER_PART=3 ...)."""
import collections
import os
import pathlib
import re
import subprocess
import sys
import tempfile


def main():
    src, name = sys.argv[1], sys.argv[2]
    minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    csrc = pathlib.Path(__file__).resolve().parent.parent / "basebandboard_amd" / "csrc"
    with tempfile.TemporaryDirectory() as td:
        out = pathlib.Path(td) / "k.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               *os.environ.get("ISA_HIST_FLAGS", "").split(), str(csrc / src), "-o", str(out)], cwd=str(csrc), stderr=subprocess.DEVNULL)
        lines = out.read_text().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    print(f"# {lines[start].split(':')[0]}")
    for l in lines[end:end + 40]:
        if re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), ", l):
            print("#", l.strip().split(".")[-1])
    blocks, cur, inasm = [], ["entry", collections.Counter()], False
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", t)
        if m:
            blocks.append(cur)
            cur = [m.group(1) + " " + (m.group(2) or ""), collections.Counter()]
            continue
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur[1][("asm:" if inasm else "") + t.split()[0]] += 1
    blocks.append(cur)
    for label, c in blocks:
        n = sum(c.values())
        if n >= minsize:
            print(f"{label[:90]}\n    {n} instructions: " + ", ".join(f"{k} {v}" for k, v in c.most_common(14)))


if __name__ == "__main__":
    main()
