#!/bin/bash
# round 5: kernel trace of the transmitter stream after the no-wrap mover and the two data-bit buffers per slot (product build)
O=gpurun_out/r05_tx
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- python3 experiments/r05_mover.py > $O/trace2.log 2>&1
python3 tools/trace_timeline.py $O/trace2/*/*kernel_trace.csv 70 > $O/tx_timeline_after.txt 2>&1
grep "noise stream" $O/trace2.log
tail -42 $O/tx_timeline_after.txt
rm -rf $O/trace2
