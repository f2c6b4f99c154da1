import csv,collections,sys
for d in sys.argv[1:]:
    rows=list(csv.DictReader(open(f'/root/repo/gpurun_out/{d}/{d}_counter_collection.csv')))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'awgn256' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    kt=list(csv.DictReader(open(f'/root/repo/gpurun_out/{d}/{d}_kernel_trace.csv')))
    t=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in kt if 'awgn256' in r['Kernel_Name']]
    us=sum(t)/len(t)/1e3
    a={c:sum(x)/len(x) for c,x in agg.items()}
    print(d,'us',round(us,1),'clk GHz',round(a['GRBM_GUI_ACTIVE']/8/us/1e3,3),'VALU/wave-ish',a['SQ_INSTS_VALU'],'cyc/instr',round(a['GRBM_GUI_ACTIVE']/8/(a['SQ_INSTS_VALU']/1017.25),3), {k:round(v) for k,v in a.items()})
