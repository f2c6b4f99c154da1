#!/bin/bash
# round 5: the whole GPU suite, smoke, then the driver's bench invocation and the default one.  usage: r05_full.sh <tag>
T=${1:-a}
O=gpurun_out/r05_full_$T
mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
rc=$?
tail -5 $O/gpu_tests.log
if [ $rc -ne 0 ]; then cp $O/gpu_tests.log $O/FAILED_gpu_tests.log; tail -60 $O/gpu_tests.log; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench20.err || { tail -30 $O/bench20.err; exit 1; }
python3 bench.py > $O/bench_default.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_steps20_warmup5.json", "bench_default.json"):
    d = json.loads(open(f"{O}/{f}").read().strip().splitlines()[-1])
    e = d["extra"]
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "verified", d["config"]["verified_vs_oracle"])
    print("  ber_sweep isolated", e["ber_sweep"]["gbit_s"], "b2b", e["ber_sweep"]["back_to_back_gbit_s"], "| multi", e["ber_sweep_multi_c_abi"]["gbit_s"],
          "| 88:", e["ber_sweep_88"]["gbit_s"], e["ber_sweep_88"]["seconds"], "proj", e["ber_sweep_88"]["projected_8_gpu"]["speedup"], "| cont", e["ber_sweep_continued"]["gbit_s"])
    print("  prbs loopback frac", e["prbs31_loopback"]["loopback_hbm_frac"], "fill", e["prbs31_loopback"]["fill_tb_s"], "chk", e["prbs31_loopback"]["check_after_fill_tb_s"],
          "| det", e["detector_stream"]["gbit_s"], "| tx", e["tx_waveform"]["gsample_s"], "| fill_", d["roofline"]["streaming_fill_gb_s"])
PY
