#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams -> doubled (MI355X_MICROARCH.md, HBM)."""
import csv, glob, sys, collections, json
def load(d, name):
    f = max(glob.glob(d + '/*/*_counter_collection.csv'), key=__import__('os').path.getmtime)     # newest run in the directory
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg
fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(fe, key=lambda k: -sum(fe[k])):
    f = sum(fe[k]) / len(fe[k]) * 1024 * 2
    w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024
    short = k.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:60]
    out[short] = {"launches": len(fe[k]), "fetch_bytes_per_launch_x2": round(f), "write_bytes_per_launch": round(w),
                  "hbm_bytes_per_launch": round(f + w)}
    print(f"{short:60s} n={len(fe[k]):3d} fetch(x2)={f/1e6:9.2f} MB write={w/1e6:9.2f} MB")
json.dump(out, open(sys.argv[3], 'w'), indent=1) if len(sys.argv) > 3 else None
