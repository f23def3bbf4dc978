#!/usr/bin/env python3
"""Per-kernel table of one model-1 iteration from the committed profiles: average launch time (single-stream eager
kernel trace), measured HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes) and the
bandwidth that implies.  python tools/kernel_roofline.py [tag] > profiles/<tag>_kernel_roofline.md"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_simnn_eager_kernel_stats.csv"))))
traffic = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_simnn_b256_bf16_hbm_traffic.json")))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    for key in ("conv2_bwd_data_kernel", "conv1_fwd_kernel", "conv2_bwd_weight_kernel", "conv2_fwd_kernel", "simnn_adam_kernel",
                "adam_dev_pc_kernel",
                "gemm_splitk_reduce", "gen_l1_kernel", "convt_k5_bn_sigmoid_kernel", "bn_finalize", "simnn_head_kernel",
                "simnn_head_final", "adam_prep_kernel", "conv2_pack_kernel"):
        if key in n:
            return key
    if "gemm_bf16_fast" in n:
        return "gemm_bf16_fast " + ("(fc1 dX: f32/bf16 A, K = 128)" if "Lb1EDF16bLb0E" in n else
                                    "(fc1 dW: A = dh^T)" if "Lb0EDF16bLb0E" in n else "(fc1 forward, split-K)")
    if "convt_s2_bn_kernel" in n:
        return "convt_s2_bn_kernel<128,64>" if "Li128E" in n else "convt_s2_bn_kernel<64,32>"
    if "slab_sum_kernel" in n:
        return "slab_sum_kernel" + n[n.index("<"):n.index(">") + 1]
    if "adam_dev_kernel" in n:
        return "adam_dev_kernel"
    return n[:40]


adam = [r for r in stats if "simnn_adam_kernel" in r["Name"] or "adam_dev_pc" in r["Name"]]
iters = int(adam[0]["Calls"]) if adam else 1
rows = []
for r in stats:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    if calls < iters:
        continue
    t = next((v for k, v in traffic.items() if short(k) == short(r["Name"])), None)
    mb = t["hbm_bytes_per_launch"] / 1e6 if t else None
    rows.append((calls / iters * avg, short(r["Name"]), calls / iters, avg, mb))
rows.sort(reverse=True)
print(f"# Model 1, one faithful iteration at B = 256 (bf16): kernels by time ({tag})\n")
print("Average launch time from `%s_simnn_eager_kernel_stats.csv` (single stream, so no kernel runs beside another); HBM "
      "bytes per launch from `%s_simnn_b256_bf16_hbm_traffic.json` (counters, FETCH_SIZE x2 + WRITE_SIZE); the launches of a "
      "kernel within an iteration differ in batch (2B and B), the averages are over both.\n" % (tag, tag))
print("| kernel | launches / iteration | avg us / launch | us / iteration | measured MB / launch | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for per_it, name, n, avg, mb in rows:
    tot += per_it
    if mb is None:
        print(f"| `{name}` | {n:g} | {avg:.1f} | {per_it:.1f} | - | - | - |")
    else:
        gbs = mb / avg * 1e3
        print(f"| `{name}` | {n:g} | {avg:.1f} | {per_it:.1f} | {mb:.1f} | {gbs:.0f} | {gbs / 8000:.2f} |")
print(f"\nSum of kernel time per iteration: {tot:.0f} us.")
