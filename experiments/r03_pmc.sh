#!/bin/bash
# round 3: counter passes over the bench command (kernels run alone under counter collection), summarised into profiles/r03_awgn_pmc.json
set -e
O=gpurun_out/r03_pmc
mkdir -p $O
export TMPDIR=/tmp BENCH_RAMP_STEPS=0
CMD="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1
python3 tools/summarise_pmc.py $O/r03_awgn_pmc.json "rocprofv3 --pmc WRITE_SIZE | FETCH_SIZE | SQ_* (four passes, --kernel-trace) -- BENCH_RAMP_STEPS=0 $CMD" $O/write/*/*counter_collection.csv $O/fetch/*/*counter_collection.csv $O/sq/*/*counter_collection.csv $O/sq2/*/*counter_collection.csv
cp $O/write/*/*counter_collection.csv $O/pmc_write_size.csv; cp $O/fetch/*/*counter_collection.csv $O/pmc_fetch_size.csv
