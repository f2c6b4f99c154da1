#!/bin/bash
# round 4: the BER trial kernel -- kernel-trace stats of ten 11-point sweeps (experiments/ber_rate.py) and an SQ counter pass over
# three sweeps (experiments/ber_prof.py; counter collection serialises kernels).  usage: r04_ber_pmc.sh <tag>
set -e
T=${1:-ber}
O=gpurun_out/r04_$T
mkdir -p $O
export TMPDIR=/tmp
python3 experiments/ber_rate.py > $O/ber_rate.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 experiments/ber_rate.py > $O/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 experiments/ber_prof.py > $O/sq.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/sq/*/*counter_collection.csv $O/pmc_sq.csv
python3 - $O <<'PY'
import csv, collections, sys, json
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{O}/pmc_sq.csv")):
    if "ber256" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    a["launches"] = len(next(iter(c.values())))
    out[k] = a
json.dump(out, open(f"{O}/ber_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $O/ber_rate.log
grep -i "ber256\|Name" $O/kernel_stats.csv | head -5
