#!/bin/bash
# round 5, first GPU call: where the round starts on this box -- isolated BER sweeps (wall, host part, kernel timeline of the
# seeding alone), and the counter passes over the PRBS loopback kernels the round-4 verdict asked for.
set -e
O=gpurun_out/r05_base
mkdir -p $O
export TMPDIR=/tmp
python3 experiments/ber_isolated.py > $O/ber_isolated.log 2>&1
python3 experiments/ber_host2.py > $O/ber_host2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/ber_trace -- python3 experiments/ber_isolated.py > $O/ber_trace.log 2>&1
python3 experiments/prbs_loopback.py > $O/prbs_loopback.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prbs_stats -- python3 experiments/prbs_pmc.py > $O/prbs_stats.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prbs_wr -- python3 experiments/prbs_pmc.py > $O/prbs_wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prbs_rd -- python3 experiments/prbs_pmc.py > $O/prbs_rd.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prbs_sq -- python3 experiments/prbs_pmc.py > $O/prbs_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/prbs_tcc -- python3 experiments/prbs_pmc.py > $O/prbs_tcc.log 2>&1 || echo "tcc pass failed" >> $O/prbs_tcc.log
for d in prbs_stats prbs_wr prbs_rd prbs_sq prbs_tcc ber_trace; do
  for f in $O/$d/*/*.csv; do [ -f "$f" ] && cp "$f" $O/${d}_$(basename $f | sed 's/^[0-9]*_//'); done
  rm -rf $O/$d
done
cat $O/ber_isolated.log $O/ber_host2.log $O/prbs_loopback.log
