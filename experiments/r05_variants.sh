#!/bin/bash
# round 5: one-off variant libraries (experiments/build_variant.py) against the product, one process each, the product first and last
# usage: r05_variants.sh <log name> <variant> ...
O=gpurun_out/r05_tx
mkdir -p $O
LOG=$O/$1; shift
for v in product "$@" product; do
  if [ $v = product ]; then a=""; else a="basebandboard_amd/libbbb_hip_$v.so"; fi
  echo "== $v" >> $LOG
  timeout -k 10 200 python3 experiments/r05_mover.py $a 2>/dev/null | grep "noise stream" >> $LOG || { echo FAILED $v; break; }
done
cat $LOG
