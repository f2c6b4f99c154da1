#!/bin/bash
O=gpurun_out/r03_final2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_profiled_noextra.json 2> $O/bench_profiled_noextra.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_noextra.csv 2>/dev/null
timeout -k 10 900 python3 -m pytest tests/ -m gpu -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
