#!/bin/bash
# round 3, final state: counter passes, kernel stats of the bench command, the default bench line
O=gpurun_out/r03_final4
mkdir -p $O
export TMPDIR=/tmp
bash experiments/r03_pmc.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/r03_pmc/r03_awgn_pmc.json gpurun_out/r03_pmc/pmc_write_size.csv gpurun_out/r03_pmc/pmc_fetch_size.csv $O/ 2>/dev/null
cp $O/r03_awgn_pmc.json profiles/r03_awgn_pmc.json 2>/dev/null      # (the bench line below cites the passes of THIS state)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_profiled_noextra.json 2> $O/bench_profiled_noextra.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_noextra.csv 2>/dev/null
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 400 $O/bench_default.json
