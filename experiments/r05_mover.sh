#!/bin/bash
O=gpurun_out/r05_mover
mkdir -p $O
for rep in 1 2; do
for b in 1 2 3 4 8; do
  echo "== blocks per CU $b (rep $rep)" >> $O/mover.log
  BBB_UNPLANE_BLOCKS_PER_CU=$b python3 experiments/r05_mover.py exp 2>/dev/null | grep noise >> $O/mover.log
done
done
cat $O/mover.log
