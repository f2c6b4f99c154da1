#!/bin/bash
# round 4, last state: the headline kernels' counter passes once more (the scheduling around them changed, the kernels did not).
O=gpurun_out/r04_final3
mkdir -p $O
export TMPDIR=/tmp BENCH_RAMP_STEPS=0
CMD="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1; echo "sq2 rc=$?"
python3 tools/summarise_pmc.py $O/r04_awgn_pmc.json "rocprofv3 --pmc WRITE_SIZE | FETCH_SIZE | SQ_* (four passes, --kernel-trace) -- BENCH_RAMP_STEPS=0 $CMD" $O/write/*/*counter_collection.csv $O/fetch/*/*counter_collection.csv $O/sq/*/*counter_collection.csv $O/sq2/*/*counter_collection.csv > $O/summarise.log 2>&1; echo "summarise rc=$?"
cp $O/write/*/*counter_collection.csv $O/pmc_write_size.csv; cp $O/fetch/*/*counter_collection.csv $O/pmc_fetch_size.csv
tail -5 $O/summarise.log
