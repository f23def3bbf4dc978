#!/usr/bin/env python3
"""Per-kernel launch durations of a rocprofv3 kernel trace, split into the long (2B batch) and short (B batch) launches:
tools/trace_split.py gpurun_out/<dir>"""
import csv, glob, collections, sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
rows = []
for k, v in d.items():
    v2 = v[len(v) // 4:]
    m = sum(v2) / len(v2)
    big = [x for x in v2 if x > m * 1.15] or v2
    small = [x for x in v2 if x <= m * 1.15] or v2
    rows.append((sum(v2), k.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:64], len(v),
                 sum(big) / len(big), len(big), sum(small) / len(small), len(small)))
for r in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r[1]:64s} n={r[2]:4d} long {r[3]:7.1f} us (n={r[4]:3d})  short {r[5]:7.1f} us (n={r[6]:3d})")
