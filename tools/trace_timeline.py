#!/usr/bin/env python3
"""Print the last N kernels of a rocprofv3 kernel trace as a timeline: start, end (us, relative), duration, queue, name.

    python3 tools/trace_timeline.py <..._kernel_trace.csv> [N=60] [name filter substring]"""
import csv, sys

def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void bbb::", "")[:44], r["Queue_Id"])
                for r in csv.DictReader(open(path)) if flt in r["Kernel_Name"])
    ks = ks[-n:]
    t0 = ks[0][0]
    for s, e, name, q in ks:
        print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  q{q}  {name}")

if __name__ == "__main__":
    main()
