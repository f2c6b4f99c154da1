#!/bin/bash
O=gpurun_out/r05_tx
mkdir -p $O
export TMPDIR=/tmp
for lv in 2 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/lt$lv -- python3 experiments/noise_level_trace.py $lv > $O/lt$lv.log 2>&1
  python3 tools/trace_timeline.py $O/lt$lv/*/*kernel_trace.csv 40 > $O/noise_timeline_level$lv.txt 2>&1
  rm -rf $O/lt$lv
done
tail -28 $O/noise_timeline_level4.txt
