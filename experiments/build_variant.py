#!/usr/bin/env python3
"""Round 5: a ONE-OFF variant of the product library for an A/B that must not disturb the product kernel's code generation (the
experiments build's sample kernel carries run-time switches inside its loop, and hipcc allocates its registers worse: 163 AGPRs
against 95).  usage: build_variant.py <name> <file.hip> <old text> <new text> [<old> <new> ...] [@<other file.hip> <old> <new> ...]
->  basebandboard_amd/libbbb_hip_<name>.so = the product's objects with the named files compiled from patched copies (every <old> must
occur, all its occurrences are replaced)."""
import pathlib, subprocess, sys
root = pathlib.Path(__file__).resolve().parent.parent
csrc = root / "basebandboard_amd" / "csrc"
name = sys.argv[1]
edits, cur, rest = {}, sys.argv[2], sys.argv[3:]
edits[cur] = []
i = 0
while i < len(rest):
    if rest[i].startswith("@"):
        cur = rest[i][1:]; edits.setdefault(cur, []); i += 1
        continue
    edits[cur].append((rest[i], rest[i + 1])); i += 2
objs = [l for l in (csrc / "Makefile").read_text().split("\n") if l.startswith("OBJS =")][0].split("=")[1].split()
objs = [str(csrc / o) for o in objs]
tmp = []
for src, pairs in edits.items():
    text = (csrc / src).read_text()
    for old, new in pairs:
        assert old in text, (src, old)
        text = text.replace(old, new)
    var = csrc / f"zz_{name}_{src}"
    var.write_text(text)
    obj = csrc / f"zz_{name}_{src}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-c", str(var), "-o", str(obj)], cwd=csrc)
    objs = [str(obj) if o == str(csrc / src.replace(".hip", ".o")) else o for o in objs]
    tmp += [var, obj]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(root / "basebandboard_amd" / f"libbbb_hip_{name}.so"), *objs])
for t in tmp:
    t.unlink()
print("built", f"libbbb_hip_{name}.so")
