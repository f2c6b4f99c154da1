"""Print a kernel timeline (start / end in us relative to the first) from a rocprofv3 kernel trace csv."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in rows[skip:]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bbb::", "")
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f}  {name[:60]}")
