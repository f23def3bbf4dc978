import csv,sys,glob
d=sys.argv[1]
f = max(glob.glob(d+'/*/*_kernel_trace.csv'), key=__import__('os').path.getmtime)     # newest run in the directory
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_dev_kernel' in r['Kernel_Name']]
s,e=idx[-3],idx[-2]
t0=int(rows[s]['Start_Timestamp'])
agg={}
for r in rows[s:e]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('_ZN12_GLOBAL__N_1','')
    d_=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if len(sys.argv)>2: print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d_:8.1f}us grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) vgpr={r.get('VGPR_Count','?')} {n[:70]}")
    k=n[:48]; a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d_
print("step span us:", (int(rows[e]['Start_Timestamp'])-t0)/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(f"{v[1]:9.1f}us x{v[0]:3d} {k}")
