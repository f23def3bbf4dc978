cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_ga -- python tools/bench_gemm.py > gpurun_out/pmc_ga.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_gb -- python tools/bench_gemm.py > gpurun_out/pmc_gb.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_gc -- python tools/bench_gemm.py > gpurun_out/pmc_gc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_ga gpurun_out/pmc_gb > gpurun_out/pmc_gemm_summary.txt 2>&1
python - <<'PY' >> gpurun_out/pmc_gemm_summary.txt 2>&1
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_gc/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    if 'gemm' in k: print(k, len(cnt[k]), {c: round(x/len(cnt[k])) for c,x in v.items()})
PY
