"""Timeline of one replayed training step from a rocprofv3 kernel trace of the default bench (graph + overlap):
start offset, duration, queue and name of every kernel between two consecutive Adam launches."""
import csv, sys, glob
d = sys.argv[1]
f = max(glob.glob(d + '/*/*_kernel_trace.csv'), key=__import__('os').path.getmtime)     # newest run in the directory
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_dev_kernel' in r['Kernel_Name']]
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
