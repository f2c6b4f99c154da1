#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes into profiles/*.json.

usage: summarise_pmc.py <out.json> <label> <counter_collection.csv> [<counter_collection.csv> ...]
Per kernel and counter: average value per launch and the launch count.  For the sample kernel the HBM
bytes per launch are derived as bench.py reports them (`roofline.traffic`): WRITE_SIZE / FETCH_SIZE are
in KiB; FETCH_SIZE is doubled on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def main():
    out, label, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = [r for f in files for r in csv.DictReader(open(f))]
    # the sample kernel is also launched for other fills (TX waveform): keep the launches of the timed
    # workload only, 1e9 samples = 1018 waves
    big = 1018 * 64
    for r in rows:
        if "awgn256_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) != big:
            continue
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"source": label, "kernels": {}}
    for k, cs in agg.items():
        res["kernels"][k] = {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
    # round 2: the sample kernel is awgn256_kernel<TX, STAGED>; the timed workload launches <false, true> (staged, the
    # default of bench.py) or <false, false> (BENCH_ONE_KERNEL=1)
    a = {}
    for key in ("bbb::awgn256_kernel<false, true>", "bbb::awgn256_kernel<false, false>", "bbb::awgn256_kernel"):
        if key in res["kernels"]:
            a = res["kernels"][key]
            res["awgn256_kernel_variant"] = key
            break
    if "WRITE_SIZE" in a and "FETCH_SIZE" in a:
        w = a["WRITE_SIZE"]["avg"] * 1024
        f = a["FETCH_SIZE"]["avg"] * 1024 * 2
        res["units"] = "WRITE_SIZE / FETCH_SIZE values are KiB; FETCH_SIZE doubled (gfx950 correction of the microarch guide)"
        res["awgn256_kernel_hbm_write_bytes_per_launch"] = w
        res["awgn256_kernel_hbm_fetch_bytes_per_launch_corrected"] = f
        res["awgn256_kernel_hbm_bytes_per_launch"] = w + f
    # what one launch of the timed workload produces: 1e9 samples, or m x 1e9 with bench.py's look-ahead (BENCH_LOOK_AHEAD,
    # default 2); bench.py only quotes a summary whose launches are the size of its own
    import os
    res["samples_per_launch"] = int(os.environ.get("BBB_SAMPLES_PER_LAUNCH", "1000000000"))
    if "GRBM_GUI_ACTIVE" in a and "SQ_INSTS_VALU" in a:
        res["awgn256_kernel_shader_cycles_per_xcd"] = a["GRBM_GUI_ACTIVE"]["avg"] / 8
    json.dump(res, open(out, "w"), indent=1)
    print(out, "kernels:", len(res["kernels"]))


if __name__ == "__main__":
    main()
