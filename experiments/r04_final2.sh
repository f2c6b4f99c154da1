#!/bin/bash
# round 4, final state (after the look-ahead delivery lost its handover event): kernel trace + stats of the bench command, the
# sample kernel's time on the machine from the trace, the default and the driver-shaped bench lines.  The counter passes of
# experiments/r04_final.sh stay valid: no kernel changed.  Everything lands under gpurun_out/r04_final2.
O=gpurun_out/r04_final2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_profiled_noextra.json 2> $O/bench_profiled_noextra.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_noextra.csv 2>/dev/null
python3 tools/trace_spacing.py $O/prof/*/*kernel_trace.csv > $O/planes_spacing.txt 2>&1; tail -1 $O/planes_spacing.txt
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; echo "bench20 rc=$?"
python3 - <<'PY'
import json
for f in ("bench_profiled_noextra", "bench_default", "bench_driver_shape"):
    d = json.loads(open(f"gpurun_out/r04_final2/{f}.json").read().strip().splitlines()[-1])
    e = d.get("extra", {})
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"],
          {k: (v.get("gbit_s") or v.get("gsample_s") or v.get("loopback_hbm_frac")) for k, v in e.items() if isinstance(v, dict)})
PY
python3 experiments/ber_rate.py > $O/ber_rate.log 2>&1; tail -2 $O/ber_rate.log
python3 experiments/det_rate.py > $O/det_rate.log 2>&1; tail -2 $O/det_rate.log
EXP= python3 experiments/prbs_loopback2.py > $O/prbs_loopback.log 2>&1; tail -3 $O/prbs_loopback.log
