"""Timeline of one replayed training step from a rocprofv3 kernel trace of the default bench (graph + overlap):
start offset, duration, queue and name of every kernel between two consecutive optimizer launches (the kernel that
applies Adam once per iteration: model 1 simnn_adam_kernel, model 2 dcnn_slab_sum<2>; older traces adam_dev_kernel)."""
import csv, sys, glob
d = sys.argv[1]
f = max(glob.glob(d + '/*/*_kernel_trace.csv'), key=__import__('os').path.getmtime)     # newest run in the directory
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for marker in ('simnn_adam_kernel', 'dcnn_slab_sum<2>', 'adam_dev_kernel'):
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    if len(idx) >= 3:
        break
assert len(idx) >= 3, "no once-per-iteration optimizer kernel found in the trace"
s, e = idx[-3], idx[-2]
t0 = int(rows[s]['Start_Timestamp'])
busy = 0.0
for r in rows[s:e]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += du
    print(f"{st:9.1f} +{du:7.1f}us q={r.get('Queue_Id','?'):>3} {n[:64]}")
print("step span us:", (int(rows[e]['Start_Timestamp']) - t0) / 1e3, " sum of kernel durations:", round(busy, 1))
