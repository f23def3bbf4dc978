cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p35
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p35 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/p35.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p35/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_dev_pc' in r['Kernel_Name']]
s, e = idx[-3], idx[-2]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = {}
for r in rows[s:e + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    q = r.get('Queue_Id', '?')
    print(f"{st:9.1f} +{du:7.1f}us q={q:>3} {n[:60]}")
print("span", (int(rows[e]['Start_Timestamp']) - t0) / 1e3)
PY
find gpurun_out/p35 -name "*.db" -delete
