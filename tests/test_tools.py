"""The profile helpers under tools/ that turn a rocprofv3 kernel trace into the figures DESIGN.md and bench.py quote."""
import subprocess
import sys

from conftest import ROOT

HEADER = "Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp\n"


def _trace(tmp_path, rows):
    p = tmp_path / "1_kernel_trace.csv"
    p.write_text(HEADER + "".join(f'KERNEL_DISPATCH,1,{q},1,1,{i},1,"{name}",{i},{s},{e}\n' for i, (name, q, s, e) in enumerate(rows)))
    return p


def test_trace_spacing_counts_a_kernel_from_its_predecessors_end(tmp_path):
    """Three launches of 2.0 ms on the machine, the second and third dispatched 0.4 ms before their predecessor ended (what the
    staged sample kernel does since round 4): End - Start says 2.0 / 2.4 / 2.4 ms, the time on the machine 2.0 each."""
    k = "void bbb::awgn256_planes_kernel<false>(unsigned int const*)"
    rows = [(k, 2, 0, 2_000_000), ("void bbb::unplane_kernel<false>()", 1, 2_050_000, 2_600_000), (k, 3, 1_600_000, 4_000_000),
            (k, 2, 3_600_000, 6_000_000)]
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "trace_spacing.py"), str(_trace(tmp_path, rows))], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert "mean end-start 2266.7 us" in last and "mean on-machine 2000.0 us" in last and "2 of 2" in last


def test_trace_timeline_lists_kernels_in_start_order(tmp_path):
    rows = [("void bbb::b_kernel()", 1, 500, 900), ("void bbb::a_kernel()", 2, 100, 700)]
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "trace_timeline.py"), str(_trace(tmp_path, rows)), "10"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert "a_kernel" in lines[0] and "q2" in lines[0] and "b_kernel" in lines[1] and lines[1].split()[0] == "0.4"
