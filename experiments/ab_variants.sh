#!/bin/bash
# round 3: same-box comparison of builds of the sample kernel (libbbb_hip_v*.so: parking budget, row order, with / without the
# advance in front of the loop at a chosen register footprint): alone, and the noise stream (two reads per kernel) beside its guests
O=gpurun_out/r09b; mkdir -p $O
for rep in 1 2 3; do
for v in A B C D E F G; do
  lib=basebandboard_amd/libbbb_hip_v$v.so
  small=0; case $v in D|E|F|G) small=1;; esac
  BBB_EXP_NOISE_SMALL=$small AB_LIB=$lib timeout -k 10 120 python experiments/ab_alone.py 2>&1 | grep alone | sed "s/^/$v /" >> $O/ab.log
  BBB_EXP_NOISE_SMALL=$small AB_ONLY2=1 AB_LIB=$lib timeout -k 10 120 python experiments/ab_lib.py 2>&1 | grep "level 2" | sed "s/^/$v /" >> $O/ab.log
done; done
cat $O/ab.log
