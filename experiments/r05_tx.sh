#!/bin/bash
# round 5: the shaping mover beside the transmitter's noise kernel -- phases and wave priorities (experiments/tx_phases.py)
O=gpurun_out/r05_tx
mkdir -p $O
export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 16" "1 8" "0 4" "2 0" "0 0"; do
  set -- $cfg
  BBB_EXP_MOVER_FLAGS=$1 BBB_EXP_PLANES_FLAGS=$2 timeout -k 10 120 python3 experiments/tx_phases.py >> $O/phases.log 2>&1 || { echo "FAILED $cfg" >> $O/phases.log; break; }
done
grep -v amdgpu.ids $O/phases.log
# kernel trace of the transmitter stream as it is (product build)
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 experiments/r05_mover.py > $O/trace.log 2>&1
python3 tools/trace_timeline.py $O/trace/*/*kernel_trace.csv 90 > $O/tx_timeline.txt 2>&1
tail -2 $O/trace.log
rm -rf $O/trace
