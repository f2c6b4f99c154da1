#!/bin/bash
# round 5, final state, beyond r05_final.sh: bench.py's multi-rank path with FOUR ranks sharing the one GPU (gloo; the 88 trials' eight groups
# over four ranks), and the random-mix soak over 300 seeds
T=${1:-z}
O=gpurun_out/r05_more_$T
mkdir -p $O
export TMPDIR=/tmp
BENCH_BACKEND=gloo BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_4rank_shared_gpu.json 2> $O/bench_4rank.err
rc=$?; echo "4-rank rc=$rc"
if [ $rc -ne 0 ]; then cp $O/bench_4rank.err $O/FAILED_bench_4rank.log; tail -30 $O/bench_4rank.err; exit 1; fi
python3 - $O <<'PY'
import json, sys
d = json.loads(open(f"{sys.argv[1]}/bench_4rank_shared_gpu.json").read().strip().splitlines()[-1])
e = d["extra"]["ber_sweep_88"]
print("4 ranks on one GPU: value", d["value"], "ranks", d.get("n_ranks_seen"), "| 88 trials:", e["seconds"], e["n_devices"], e["counters_first_point_first_seed"])
PY
BBB_SOAK_SEEDS=300 timeout -k 10 900 python3 -m pytest tests/test_gpu_staged.py -x -q -k random_mix > $O/soak_300_seeds.log 2>&1
rc=$?; echo "soak rc=$rc"; tail -2 $O/soak_300_seeds.log
if [ $rc -ne 0 ]; then cp $O/soak_300_seeds.log $O/FAILED_soak.log; exit 1; fi
