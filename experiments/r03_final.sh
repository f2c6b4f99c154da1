#!/bin/bash
# round 3, final state: the rest of the GPU tests, the default bench line, the profiled bench (kernel stats)
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python3 -m pytest tests/ -m gpu -q --deselect tests/test_gpu_awgn.py --deselect tests/test_gpu_staged.py --deselect tests/test_gpu_ber.py > $O/gputests_rest.log 2>&1; tail -3 $O/gputests_rest.log
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
head -12 $O/kernel_stats.csv | cut -c1-160
