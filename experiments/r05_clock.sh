#!/bin/bash
O=gpurun_out/r05_tx
mkdir -p $O
for v in st st_nostore st_nomover st st_nostore st_nomover; do
  timeout -k 10 200 python3 experiments/clock_ab.py basebandboard_amd/libbbb_hip_$v.so 2>/dev/null | grep Gsample >> $O/clock_ab.log || { echo FAILED $v; break; }
done
cat $O/clock_ab.log
