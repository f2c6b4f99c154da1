#!/usr/bin/env python3
"""Round 5: a ONE-OFF variant of the product library for an A/B that must not disturb the product kernel's code generation (the
experiments build's sample kernel carries run-time switches inside its loop, and hipcc allocates its registers worse: 163 AGPRs
against 95).  usage: build_variant.py <name> <file.hip> <old text> <new text> [<old> <new> ...]  ->  basebandboard_amd/libbbb_hip_<name>.so =
the product's objects with <file.hip> compiled from a patched copy (every <old> must occur, all its occurrences are replaced)."""
import pathlib, subprocess, sys
root = pathlib.Path(__file__).resolve().parent.parent
csrc = root / "basebandboard_amd" / "csrc"
name, src = sys.argv[1], sys.argv[2]
text = (csrc / src).read_text()
for old, new in zip(sys.argv[3::2], sys.argv[4::2]):
    assert old in text, old
    text = text.replace(old, new)
var = csrc / f"zz_{name}_{src}"
var.write_text(text)
obj = csrc / f"zz_{name}_{src}.o"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-c", str(var), "-o", str(obj)], cwd=csrc)
objs = [l for l in (csrc / "Makefile").read_text().split("\n") if l.startswith("OBJS =")][0].split("=")[1].split()
objs = [str(obj) if o == src.replace(".hip", ".o") else str(csrc / o) for o in objs]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(root / "basebandboard_amd" / f"libbbb_hip_{name}.so"), *objs])
var.unlink(); obj.unlink()
print("built", f"libbbb_hip_{name}.so")
