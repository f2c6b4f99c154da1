"""Prints the kernel timeline between the last two BER kernels of a rocprofv3 --kernel-trace run of experiments/ber_rate.py.
usage: ber_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ber256' in r['Kernel_Name']]
i1, i2 = idx[-3], idx[-2]
t0 = int(rows[i1]['End_Timestamp'])
for r in rows[i1:i2 + 1]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} q{r.get('Queue_Id','')} {r['Kernel_Name'][:80]} grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
