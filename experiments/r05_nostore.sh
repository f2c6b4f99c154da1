#!/bin/bash
# round 5: what the sample kernel's staging stores cost it BESIDE its guests -- the product library against a one-off variant whose
# kernel is the product's minus its four store statements (experiments/build_variant.py nostore ...), alternating
O=gpurun_out/r05_tx
mkdir -p $O
for v in product nostore product nostore; do
  if [ $v = product ]; then a=""; else a="basebandboard_amd/libbbb_hip_$v.so"; fi
  echo "== $v" >> $O/nostore_ab.log
  timeout -k 10 200 python3 experiments/r05_mover.py $a 2>/dev/null | grep "noise stream" >> $O/nostore_ab.log || { echo FAILED; break; }
done
cat $O/nostore_ab.log
