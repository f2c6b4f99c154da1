#!/bin/bash
# round 5: the isolated BER sweep after a change -- parity first, then wall times and the kernel timeline.  usage: r05_ber.sh <tag>
set -e
T=${1:-a}
O=gpurun_out/r05_ber_$T
mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_ber.py tests/test_gpu_search.py::test_found_k256_matrix_runs_ber_trials_and_tx -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; cp $O/tests.log profiles/r05_fail_ber_$T.log 2>/dev/null; exit 1; }
tail -3 $O/tests.log
python3 experiments/ber_isolated.py > $O/ber_isolated.log 2>&1
python3 experiments/ber_host2.py > $O/ber_host2.log 2>&1
python3 experiments/ber_multi_rate.py > $O/ber_multi_rate.log 2>&1
python3 experiments/ber_rate.py > $O/ber_rate.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 experiments/ber_isolated.py > $O/trace.log 2>&1
cp $O/trace/*/*kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/trace
python3 tools/trace_timeline.py $O/kernel_trace.csv 24 > $O/timeline.txt
cat $O/ber_isolated.log $O/ber_host2.log $O/ber_multi_rate.log $O/ber_rate.log $O/timeline.txt
