#!/bin/bash
# round 5: the tail seeding kernel with a quarter of its lookups through the vector cache (experiments build, BBB_EXP_TAIL_GLOBAL_WORDS=2)
O=gpurun_out/r05_tail
mkdir -p $O
export TMPDIR=/tmp
for v in 0 2 0 2; do
  echo "== BBB_EXP_TAIL_GLOBAL_WORDS=$v" >> $O/ab.log
  EXP=1 BBB_EXP_TAIL_GLOBAL_WORDS=$v python3 experiments/ber_multi_rate.py 2>/dev/null | grep "sweep_multi x1\|ber_trials:" | cut -c1-160 >> $O/ab.log
done
EXP=1 BBB_EXP_TAIL_GLOBAL_WORDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python3 experiments/ber_prof5.py > $O/p2.log 2>&1
EXP=1 BBB_EXP_TAIL_GLOBAL_WORDS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 experiments/ber_prof5.py > $O/p0.log 2>&1
grep -h "seed_tail" $O/p0/*/*kernel_stats.csv $O/p2/*/*kernel_stats.csv | cut -c1-60,150-230 >> $O/ab.log
rm -rf $O/p0 $O/p2
cat $O/ab.log
