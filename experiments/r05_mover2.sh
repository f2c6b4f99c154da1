#!/bin/bash
# round 5: the mover with its tile in the raw buffer's place (64 / 76 KiB of LDS): parity, then the stream rates against blocks per CU
O=gpurun_out/r05_mover2
mkdir -p $O
python3 -m pytest tests/test_gpu_staged.py tests/test_gpu_stream.py tests/test_gpu_tx.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; cp $O/tests.log $O/FAILED_tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
for b in 1 2; do
  echo "== blocks per CU $b (rep $rep)" >> $O/mover.log
  BBB_UNPLANE_BLOCKS_PER_CU=$b python3 experiments/r05_mover.py exp 2>/dev/null | grep noise >> $O/mover.log
done
done
echo "== product build" >> $O/mover.log
python3 experiments/r05_mover.py 2>/dev/null | grep noise >> $O/mover.log
cat $O/mover.log
