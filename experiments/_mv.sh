export EXP=1 MS=1,1
run() { env "$@" timeout -k 10 100 python experiments/la_ab.py 2>&1 | grep -E "^m=" | awk -v c="$*" '{printf "%s | ", $5} END {print c}'; }
for i in 1 2 3; do
run BBB_UNSTAGE_THREADS=128
run BBB_UNSTAGE_THREADS=64 BBB_UNSTAGE_BLOCKS=384
run BBB_UNSTAGE_THREADS=64 BBB_UNSTAGE_BLOCKS=448
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MS=1 BBB_UNSTAGE_THREADS=64 BBB_UNSTAGE_BLOCKS=384 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mv_stats -- python3 experiments/la_ab.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/mv_stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("unstage","awgn256")): print("K", r["Name"][:44].ljust(44), r["Calls"], round(float(r["AverageNs"])/1e6,4), round(float(r["MaxNs"])/1e6,4))
PY
