"""Design experiment: PRBS-31 fill / check bandwidth vs words-per-lane and waves per CU."""
import os, subprocess, sys, json
if len(sys.argv) > 1:
    import torch, ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import basebandboard_amd as bbb
    bbb._lib.select_build("experiments")   # the knobs below exist only in that build
    nbits = 10_000_000_000
    p = bbb.PRBS(31); det = bbb.PRBSErrorDetector(31)
    buf = torch.empty((nbits + 63)//64, dtype=torch.int64, device="cuda")
    p.generate(nbits, out=buf); n0 = det.count_errors(buf, nbits)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    g = c = 0.0
    for _ in range(5):
        e[0].record(); p.generate(nbits, out=buf); e[1].record(); n = det.count_errors(buf, nbits); e[2].record()
        torch.cuda.synchronize(); g += e[0].elapsed_time(e[1]); c += e[1].elapsed_time(e[2])
    print(json.dumps({"cfg": sys.argv[1], "errors": n0, "gen_TBs": nbits/8*5/g/1e9, "chk_TBs": nbits/8*5/c/1e9}))
else:
    def run(tag, **kv):
        env = dict(os.environ, **{k: str(v) for k, v in kv.items()})
        subprocess.run([sys.executable, __file__, tag + " " + " ".join(f"{k[9:]}={v}" for k, v in kv.items())], env=env)
    for cap in (0, 4, 8):
        for wpl in (1, 2):
            for lw in (0, 1):
                if wpl == 2 and lw == 0:
                    continue            # spills to scratch: known slow
                run("FILL ", BBB_PRBS_FILL_WPL=wpl, BBB_PRBS_FILL_LW=lw, BBB_PRBS_WAVES_PER_CU=cap)
                run("CHECK", BBB_PRBS_CHECK_WPL=wpl, BBB_PRBS_CHECK_LW=lw, BBB_PRBS_WAVES_PER_CU=cap)
