#!/bin/bash
# round 5: timelines of the transmitter stream with the mover above the noise kernel in wave priority, and with the noise kernel's stores off
O=gpurun_out/r05_tx
mkdir -p $O
export TMPDIR=/tmp
for cfg in "1 16" "1 17" "0 1" "0 0"; do
  set -- $cfg
  BBB_EXP_MOVER_FLAGS=$1 BBB_EXP_PLANES_FLAGS=$2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1_$2 -- python3 experiments/r05_mover.py exp > $O/trace_$1_$2.log 2>&1 || { echo "FAILED $cfg"; break; }
  python3 tools/trace_timeline.py $O/trace_$1_$2/*/*kernel_trace.csv 60 > $O/tx_timeline_mover$1_planes$2.txt 2>&1
  grep "noise stream" $O/trace_$1_$2.log
  rm -rf $O/trace_$1_$2
done
