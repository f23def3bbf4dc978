import csv,sys,glob,collections
def load(d):
    f = max(glob.glob(d+'/*/*_counter_collection.csv'), key=__import__('os').path.getmtime)     # newest run in the directory
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in rows:
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('_ZN12_GLOBAL__N_1','')[:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    return agg,cnt
a,ca=load(sys.argv[1]); b,cb=load(sys.argv[2])
keys=sorted(a, key=lambda k:-a[k].get('SQ_WAVE_CYCLES',0))[:9]
for k in keys:
    n=len(ca[k]); A=a[k]; B=b.get(k,{})
    wc=A['SQ_WAVE_CYCLES']
    print(f"\n{k}  dispatches={n}")
    print("  wave_cycles/disp=%.3g busy_cyc/disp=%.3g  wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%%"%(wc/n, A['SQ_BUSY_CYCLES']/n, 100*A['SQ_WAIT_ANY']/wc, 100*A['SQ_WAIT_INST_ANY']/wc, 100*A['SQ_ACTIVE_INST_ANY']/wc))
    print("  insts/disp: valu=%.3g lds=%.3g vmem=%.3g salu=%.3g mfma=%.3g"%(A['SQ_INSTS_VALU']/n, A['SQ_INSTS_LDS']/n, A['SQ_INSTS_VMEM']/n, B.get('SQ_INSTS_SALU',0)/max(1,len(cb.get(k,[1]))), B.get('SQ_INSTS_MFMA',0)/max(1,len(cb.get(k,[1])))))
    if B:
        nb=len(cb[k]); ia=B['SQ_LDS_IDX_ACTIVE']
        print("  lds_idx_active/disp=%.3g bank_conflict=%.0f%% of lds-active; active_valu=%.3g active_lds=%.3g wait_inst_lds=%.3g mfma_busy=%.3g"%(ia/nb, 100*B['SQ_LDS_BANK_CONFLICT']/max(ia,1), B['SQ_ACTIVE_INST_VALU']/nb, B['SQ_ACTIVE_INST_LDS']/nb, B['SQ_WAIT_INST_LDS']/nb, B['SQ_VALU_MFMA_BUSY_CYCLES']/nb))
