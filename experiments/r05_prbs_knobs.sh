#!/bin/bash
# round 5: the PRBS loopback against the partition (waves per CU) and the lane width, experiments build, same box
O=gpurun_out/r05_prbs_knobs
mkdir -p $O
for rep in 1 2; do
for w in 4 6 8 10; do
  for wpl in 1 2; do
    echo "== waves/CU $w  fill WPL $wpl (rep $rep)" >> $O/knobs.log
    BBB_PRBS_WAVES_PER_CU=$w BBB_PRBS_FILL_WPL=$wpl python3 experiments/prbs_loopback.py exp 2>/dev/null | grep fill >> $O/knobs.log
  done
done
done
cat $O/knobs.log
