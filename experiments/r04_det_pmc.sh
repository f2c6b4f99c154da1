#!/bin/bash
# round 4: the detector's chunk kernel -- kernel-trace stats and an SQ counter pass over experiments/det_rate.py.  usage: r04_det_pmc.sh <tag>
set -e
T=${1:-det}
O=gpurun_out/r04_$T
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 experiments/det_rate.py > $O/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 experiments/det_rate.py > $O/sq.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/sq/*/*counter_collection.csv $O/pmc_sq.csv
python3 - $O <<'PY'
import csv, collections, sys, json
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{O}/pmc_sq.csv")):
    if "det_" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    a["launches"] = len(next(iter(c.values())))
    out[k] = a
json.dump(out, open(f"{O}/det_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
grep -i "det_\|Name" $O/kernel_stats.csv | head -8
grep call $O/stats.log | tail -3
