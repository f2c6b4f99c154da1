#!/bin/bash
# round 3: same-box comparison of builds of the library (basebandboard_amd/libbbb_hip_v*.so): the sample kernel alone, and the
# noise stream (two reads per kernel) beside its guests; AB_SMALL lists the variants that need BBB_EXP_NOISE_SMALL=1
O=${AB_OUT:-gpurun_out/r09c}; mkdir -p $O
for rep in 1 2 3; do
for lib in basebandboard_amd/libbbb_hip_v*.so; do
  v=$(basename $lib .so | sed 's/libbbb_hip_v//')
  small=0; case " $AB_SMALL " in *" $v "*) small=1;; esac
  BBB_EXP_NOISE_SMALL=$small AB_LIB=$lib timeout -k 10 120 python experiments/ab_alone.py 2>&1 | grep alone | sed "s/^/$v /" >> $O/ab.log
  BBB_EXP_NOISE_SMALL=$small AB_ONLY2=1 AB_LIB=$lib timeout -k 10 120 python experiments/ab_lib.py 2>&1 | grep "level 2" | sed "s/^/$v /" >> $O/ab.log
done; done
cat $O/ab.log
