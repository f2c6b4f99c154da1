#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes into profiles/*.json (round 3 layout).

usage: summarise_pmc.py <out.json> <label> <counter_collection.csv> [<counter_collection.csv> ...]
Per kernel and counter: average value per launch and the launch count.  Then, for the headline path (the PLANES form of
the staged stream: awgn256_planes_kernel -> unplane_kernel, seeding kernels beside them), what bench.py reports:
  sample_kernel  HBM bytes per launch (WRITE_SIZE / FETCH_SIZE are KiB; FETCH_SIZE doubled on gfx950 as
                 /opt/skills/guides/MI355X_MICROARCH.md prescribes), VALU instructions per launch / per step and wave
  mover          HBM bytes per launch of unplane_kernel (one launch delivers one read of the stream)
  seeding        HBM bytes of one start-state derivation (seed_store16 + seed_level x N + bitslice)
Counter collection serialises the kernels, so every figure is for a kernel running ALONE."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def main():
    out, label, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    per_launch = int(os.environ.get("BBB_SAMPLES_PER_LAUNCH", "2000000000"))     # bench.py: stream object, two reads per kernel
    per_read = int(os.environ.get("BBB_SAMPLES_PER_READ", "1000000000"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    grids = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grids[k].add(int(r["Grid_Size"]))
    res = {"source": label, "units": "WRITE_SIZE / FETCH_SIZE values are KiB; FETCH_SIZE doubled for the byte figures (gfx950 correction of the "
                                     "microarch guide); counter collection serialises kernels: every kernel ran alone",
           "samples_per_launch": per_launch, "samples_per_read": per_read, "kernels": {}}
    for k, cs in agg.items():
        res["kernels"][k] = {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}

    def hbm(k):
        a = res["kernels"].get(k, {})
        if "WRITE_SIZE" not in a or "FETCH_SIZE" not in a:
            return None
        return {"write": a["WRITE_SIZE"]["avg"] * 1024, "fetch_corrected": a["FETCH_SIZE"]["avg"] * 1024 * 2,
                "total": a["WRITE_SIZE"]["avg"] * 1024 + a["FETCH_SIZE"]["avg"] * 1024 * 2, "launches": a["WRITE_SIZE"]["launches"]}

    sk = next((k for k in res["kernels"] if "awgn256_planes_kernel" in k), None)
    mv = next((k for k in res["kernels"] if "unplane_kernel<false" in k), None)      # (<false> until round 5, <false, false> since the no-wrap parameter)
    if sk:
        rec = {"kernel": sk, "hbm_bytes_per_launch": hbm(sk)}
        a = res["kernels"][sk]
        if "SQ_INSTS_VALU" in a:
            waves = max(grids[sk]) // 64
            L = -(-per_launch // (1024 * 2048))
            L = (L + 15) // 16 * 16
            rec.update({"valu_insts_per_launch": a["SQ_INSTS_VALU"]["avg"], "waves": waves, "steps_per_wave": L,
                        "valu_insts_per_step_and_wave": a["SQ_INSTS_VALU"]["avg"] / waves / L})
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE",
                      "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"):
                if c in a:
                    rec[c] = a[c]["avg"]
            if "SQ_WAVE_CYCLES" in a:
                rec["valu_busy_frac_of_wave_cycles"] = a.get("SQ_ACTIVE_INST_VALU", {"avg": 0})["avg"] / a["SQ_WAVE_CYCLES"]["avg"]
        res["sample_kernel"] = rec
    if mv:
        res["mover"] = {"kernel": mv, "hbm_bytes_per_launch": hbm(mv)}
        a = res["kernels"][mv]
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"):
            if c in a:
                res["mover"][c] = a[c]["avg"]
    # one seeding = one seed_store16 launch and what follows it; bytes = sum over its kernels, per seeding
    seed = {}
    for k in res["kernels"]:
        if any(x in k for x in ("seed_store16_kernel<8>", "seed_level_kernel<8>", "bitslice_kernel<8>")):
            h = hbm(k)
            if h:
                seed[k] = h
    n_seed = res["kernels"].get(next((k for k in seed if "seed_store16" in k), ""), {}).get("WRITE_SIZE", {}).get("launches", 0)
    if seed and n_seed:
        tot = sum(h["total"] * h["launches"] for h in seed.values()) / n_seed
        res["seeding"] = {"kernels": sorted(seed), "hbm_bytes_per_seeding": tot, "seedings": n_seed}
    if res.get("sample_kernel", {}).get("hbm_bytes_per_launch") and res.get("mover", {}).get("hbm_bytes_per_launch"):
        m = per_launch / per_read
        step = res["sample_kernel"]["hbm_bytes_per_launch"]["total"] / m + res["mover"]["hbm_bytes_per_launch"]["total"] + \
            res.get("seeding", {}).get("hbm_bytes_per_seeding", 0.0) / m
        res["traffic_per_read"] = {"bytes": step, "algorithmic_bytes": per_read, "ratio": step / per_read,
                                   "what": "sample kernel / reads per launch + mover + seeding / reads per launch"}
    json.dump(res, open(out, "w"), indent=1)
    print(out, "kernels:", len(res["kernels"]))


if __name__ == "__main__":
    main()
