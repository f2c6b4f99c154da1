#!/usr/bin/env python3
"""Time on the machine of a kernel whose launches are dispatched into each other's tail, from a rocprofv3 kernel trace.

    python3 tools/trace_spacing.py <..._kernel_trace.csv> [kernel-name substring, default awgn256_planes_kernel]

rocprofv3's Start_Timestamp is the DISPATCH of a kernel.  The staged sample kernel (one 512-register wave per SIMD, 1024
workgroups) is dispatched as soon as its start states exist -- while its predecessor still holds every SIMD -- so its
`End - Start` contains the wait for those waves to retire and `--stats` averages that.  What the roofline record of bench.py
uses is the kernel's time on the machine: End - max(Start, End of the previous launch of the same kernel), which is what
`bbb_lutopt_profile_read` measures with hipEvents.  This prints both, per launch and averaged, so the two can be compared."""
import csv, sys

def main():
    path = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "awgn256_planes_kernel"
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if name in r["Kernel_Name"])
    if not ks:
        sys.exit(f"no kernel matching {name!r} in {path}")
    rows, prev_end = [], None
    for s, e in ks:
        on = e - max(s, prev_end) if prev_end is not None else e - s
        rows.append((e - s, on, (s - prev_end) if prev_end is not None else 0))
        prev_end = e
    print(f"{name}: {len(rows)} launches")
    print("launch  end-start_us  on_machine_us  start-prev_end_us")
    for i, (d, on, gap) in enumerate(rows):
        print(f"{i:6d}  {d/1e3:12.1f}  {on/1e3:13.1f}  {gap/1e3:17.1f}")
    n = len(rows)
    print(f"mean end-start {sum(r[0] for r in rows)/n/1e3:.1f} us; mean on-machine {sum(r[1] for r in rows)/n/1e3:.1f} us; "
          f"launches dispatched before the previous one ended: {sum(1 for r in rows[1:] if r[2] < 0)} of {n - 1}")

if __name__ == "__main__":
    main()
