#!/bin/bash
# round 4, final state: counter passes over the bench command, kernel stats, the default and the driver-shaped bench lines,
# the BER and detector kernels' own passes.  Everything lands under gpurun_out/r04_final; the summaries are copied to profiles/ by hand.
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp BENCH_RAMP_STEPS=0
CMD="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1; echo "sq2 rc=$?"
python3 tools/summarise_pmc.py $O/r04_awgn_pmc.json "rocprofv3 --pmc WRITE_SIZE | FETCH_SIZE | SQ_* (four passes, --kernel-trace) -- BENCH_RAMP_STEPS=0 $CMD" $O/write/*/*counter_collection.csv $O/fetch/*/*counter_collection.csv $O/sq/*/*counter_collection.csv $O/sq2/*/*counter_collection.csv > $O/summarise.log 2>&1; echo "summarise rc=$?"
cp $O/write/*/*counter_collection.csv $O/pmc_write_size.csv; cp $O/fetch/*/*counter_collection.csv $O/pmc_fetch_size.csv
cp $O/r04_awgn_pmc.json profiles/r04_awgn_pmc.json 2>/dev/null      # (the bench lines below cite the passes of THIS state)
unset BENCH_RAMP_STEPS
bash experiments/r04_ber_pmc.sh final_ber > $O/ber_pmc.log 2>&1; echo "ber rc=$?"
bash experiments/r04_det_pmc.sh final_det > $O/det_pmc.log 2>&1; echo "det rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_profiled_noextra.json 2> $O/bench_profiled_noextra.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_noextra.csv 2>/dev/null
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; echo "bench20 rc=$?"
python3 experiments/ber_run_rate.py > $O/ber_run_rate.log 2>&1
python3 experiments/det_rate.py > $O/det_rate.log 2>&1
EXP= python3 experiments/prbs_loopback2.py > $O/prbs_loopback.log 2>&1
tail -c 300 $O/bench_default.json
