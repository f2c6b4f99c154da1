#!/bin/bash
# round 3, first GPU call: today's state on a fresh box -- bench line, kernel timeline, SQ counters of the sample kernel
set -e
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 experiments/staged_trace.py > $O/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_guests -- python3 experiments/staged_trace.py > $O/pmc_sq_guests.log 2>&1
echo "pmc guests done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_alone -- python3 experiments/staged_parts.py > $O/pmc_sq_alone.log 2>&1
echo "pmc alone done"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_sq2_alone -- python3 experiments/staged_parts.py > $O/pmc_sq2_alone.log 2>&1
echo "pmc2 alone done"
find $O -name "*.csv" | head -40
