#!/bin/bash
# round 5: counters of the transmitter stream's kernels (product build)
O=gpurun_out/r05_tx
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 experiments/tx_pmc.py > $O/pmc.log 2>&1; echo "sq rc=$?"
python3 - $O <<'PY'
import csv, collections, sys, json, glob
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "experiments/r05_tx2.sh: rocprofv3 --pmc SQ_* --kernel-trace over experiments/tx_pmc.py (twelve transmitter-stream calls of 1e9 samples: six noise kernels of 2e9, twelve shaping movers)", "kernels": {}}
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    a["launches"] = len(next(iter(c.values())))
    out["kernels"][k] = a
json.dump(out, open(f"{O}/r05_tx_pmc.json", "w"), indent=1)
for k, a in out["kernels"].items(): print(k[:50], a["launches"], round(a.get("SQ_INSTS_VALU", 0)), round(a.get("SQ_INSTS_LDS", 0)), round(a.get("GRBM_GUI_ACTIVE", 0)))
PY
rm -rf $O/sq
