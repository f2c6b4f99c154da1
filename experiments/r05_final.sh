#!/bin/bash
# round 5, final state: the whole GPU suite, smoke, the bench lines, the kernel trace + stats of the bench command, the headline kernels'
# counter passes, the BER path's (seeding + trial kernel) trace and counters, and the multi-rank bench path rehearsed with two ranks on the
# one GPU (gloo).  usage: r05_final.sh <tag>.  A failing step copies its log to FAILED_*.log (kept: profiles/r05_fail_*).
T=${1:-z}
O=gpurun_out/r05_final_$T
mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
rc=$?
tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then cp $O/gpu_tests.log $O/FAILED_gpu_tests.log; tail -60 $O/gpu_tests.log; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cp $O/smoke.log $O/FAILED_smoke.log; cat $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench20.err || { cp $O/bench20.err $O/FAILED_bench20.log; tail -30 $O/bench20.err; exit 1; }
python3 bench.py > $O/bench_default.json 2> $O/bench.err || { cp $O/bench.err $O/FAILED_bench.log; tail -30 $O/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_profiled_noextra.json 2> $O/bench_profiled_noextra.err; echo "prof rc=$?"
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_noextra.csv 2>/dev/null
python3 tools/trace_spacing.py $O/prof/*/*kernel_trace.csv > $O/planes_spacing.txt 2>&1; tail -1 $O/planes_spacing.txt
rm -rf $O/prof
# headline kernels: counter passes
export BENCH_RAMP_STEPS=0
CMD="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1; echo "sq rc=$?"
python3 tools/summarise_pmc.py $O/r05_awgn_pmc.json "rocprofv3 --pmc WRITE_SIZE | FETCH_SIZE | SQ_* (three passes, --kernel-trace) -- BENCH_RAMP_STEPS=0 $CMD" $O/write/*/*counter_collection.csv $O/fetch/*/*counter_collection.csv $O/sq/*/*counter_collection.csv > $O/summarise.log 2>&1; echo "summarise rc=$?"
cp $O/write/*/*counter_collection.csv $O/pmc_write_size.csv; cp $O/fetch/*/*counter_collection.csv $O/pmc_fetch_size.csv
rm -rf $O/write $O/fetch $O/sq
unset BENCH_RAMP_STEPS
# the BER path
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bprof -- python3 experiments/ber_prof5.py > $O/ber_prof.log 2>&1; echo "ber prof rc=$?"
cp $O/bprof/*/*kernel_stats.csv $O/ber_kernel_stats.csv 2>/dev/null; rm -rf $O/bprof
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bsq -- python3 experiments/ber_prof5.py > $O/ber_sq.log 2>&1; echo "ber sq rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bwr -- python3 experiments/ber_prof5.py > $O/ber_wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/brd -- python3 experiments/ber_prof5.py > $O/ber_rd.log 2>&1
python3 - $O <<'PY'
import csv, collections, sys, json, glob
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("bsq", "bwr", "brd"):
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "ber256" in n or "seed_" in n:
                agg[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "experiments/r05_final.sh: rocprofv3 --pmc passes over experiments/ber_prof5.py (four isolated 11 x 1e9-bit sweeps, each seeding itself); WRITE_SIZE / FETCH_SIZE in KiB (FETCH_SIZE to be doubled on gfx950)", "kernels": {}}
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    a["launches"] = len(next(iter(c.values())))
    out["kernels"][k] = a
fk = next((k for k in out["kernels"] if "ber256_fused" in k), None)
if fk and "SQ_INSTS_VALU" in out["kernels"][fk]:
    waves, L = 1022, 478
    out["fused"] = {"kernel": fk, "valu_insts_per_step_and_wave": out["kernels"][fk]["SQ_INSTS_VALU"] / waves / L, "waves": waves, "steps_per_wave": L}
json.dump(out, open(f"{O}/r05_ber_pmc.json", "w"), indent=1)
print(json.dumps(out.get("fused"), indent=1))
PY
rm -rf $O/bsq $O/bwr $O/brd
# the multi-rank path of bench.py with two ranks sharing the one GPU (gloo for the collectives)
BENCH_BACKEND=gloo BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2> $O/bench_2rank.err; echo "2-rank rc=$?"
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_steps20_warmup5.json", "bench_default.json", "bench_profiled_noextra.json", "bench_2rank_shared_gpu.json"):
    try:
        d = json.loads(open(f"{O}/{f}").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    e = d.get("extra", {})
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], "ranks", d.get("n_ranks_seen"))
    if "ber_sweep" in e:
        print("  ber isolated", e["ber_sweep"]["gbit_s"], "b2b", e["ber_sweep"]["back_to_back_gbit_s"], "| 88:", e["ber_sweep_88"]["gbit_s"], e["ber_sweep_88"]["seconds"],
              e["ber_sweep_88"].get("projected_8_gpu", {}).get("speedup"), "| multi", e.get("ber_sweep_multi_c_abi", {}).get("gbit_s"))
        print("  prbs", e["prbs31_loopback"]["loopback_hbm_frac"], "| det", e["detector_stream"]["gbit_s"], "| tx", e["tx_waveform"]["gsample_s"], "| fill_", d["roofline"]["streaming_fill_gb_s"])
PY
grep -i "ber256\|seed_\|Name" $O/ber_kernel_stats.csv | cut -c1-160 | head -8
