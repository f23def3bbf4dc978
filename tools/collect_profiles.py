#!/usr/bin/env python3
"""Copy the summaries tools/profile_round.sh (and tools/pmc_ops.sh) left under gpurun_out/ into profiles/ (tracked).
Run in the container after the gpurun call: python tools/collect_profiles.py [round-tag, default r01]."""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(out, pattern))
    return max(files, key=os.path.getmtime) if files else None


def copy(src, name):
    if src and os.path.exists(src):
        shutil.copyfile(src, os.path.join(prof, name))
        print(f"{os.path.relpath(src, ROOT)} -> profiles/{name}")
    else:
        print(f"missing: {name}")


copy(newest("final_simnn/*/*_kernel_stats.csv"), f"{tag}_simnn_kernel_stats.csv")
copy(newest("final_simnn_eager/*/*_kernel_stats.csv"), f"{tag}_simnn_eager_kernel_stats.csv")
copy(newest("final_mmgan/*/*_kernel_stats.csv"), f"{tag}_mmgan_kernel_stats.csv")
copy(os.path.join(out, "hbm_traffic.json"), f"{tag}_simnn_b256_bf16_hbm_traffic.json")
copy(os.path.join(out, "pmc_sq.txt"), f"{tag}_simnn_b256_bf16_pmc_sq.txt")
copy(os.path.join(out, "step_breakdown.txt"), f"{tag}_simnn_b256_bf16_step_breakdown.txt")
copy(os.path.join(out, "bench_default.json"), f"{tag}_bench_default.json")
copy(os.path.join(out, "bench_mmgan.json"), f"{tag}_bench_mmgan.json")
copy(os.path.join(out, "pmc_ops_summary.txt"), f"{tag}_conv_kernels_b512_pmc_sq.txt")
for name in ("bench_default_20", "bench_simnn_fp32_nopipeline", "bench_simnn_eager", "bench_simnn_elided", "bench_simnn_nopipeline", "bench_simnn_fp32", "bench_mmgan_eager",
             "bench_mmgan_b16", "bench_simnn_c1_b16_w64", "bench_simnn_c1_b16_w216", "bench_simnn_c5_b128_w216"):
    copy(os.path.join(out, name + ".json"), f"{tag}_{name}.json")
copy(os.path.join(out, f"pytest_gpu_{tag}.log"), f"{tag}_pytest_gpu.log")
copy(os.path.join(out, f"parity_{tag}.jsonl"), f"{tag}_parity_measurements.jsonl")
copy(os.path.join(out, "mmgan_replay_timeline.txt"), f"{tag}_mmgan_replay_timeline.txt")

# bench.py reads the dominant kernel's measured HBM bytes per launch from profiles/traffic.json
src = os.path.join(out, "hbm_traffic.json")
if os.path.exists(src):
    d = json.load(open(src))
    fused = [v for k, v in d.items() if "conv2_bwd_data_kernel" in k and "true" in k]
    if fused:
        # whole-iteration traffic: every kernel's bytes x its launches, over the iterations of the counter passes (Adam on
        # fc1.weight runs once per iteration); algorithmic bytes per step from SURVEY.md section 8d (B = 256, 128x256:
        # 395.2 KB of data per sample + 504 MB of parameter / optimizer traffic; code2 is half as wide since round 3)
        adam = [v for k, v in d.items() if "simnn_adam_kernel" in k or "adam_dev_pc" in k]
        assert adam, "no optimizer kernel in the traffic profile: cannot count iterations"
        iters = adam[0]["launches"] if adam else 1
        step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in d.values()) / iters
        t = {"simnn_bf16": int(fused[0]["hbm_bytes_per_launch"]),
             "simnn_bf16_step": {"measured_bytes_per_step": int(step_bytes),
                                 "algorithmic_bytes_per_step": int(256 * 395.2e3 + 504e6),
                                 "source": f"profiles/{tag}_simnn_b256_bf16_hbm_traffic.json (sum over all kernels of one "
                                           "faithful iteration, rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes) over "
                                           "SURVEY.md 8d's 605 MB"},
             "_source": f"profiles/{tag}_simnn_b256_bf16_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                        "passes, FETCH_SIZE x2 on gfx950), average over the 2B and B launches of one faithful iteration; "
                        "kernel: conv2_bwd_data_kernel<FUSE>"}
        json.dump(t, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
        print("profiles/traffic.json:", t["simnn_bf16"])

# per-kernel time / measured-traffic table of the model-1 iteration
import subprocess
md = os.path.join(prof, f"{tag}_kernel_roofline.md")
with open(md, "w") as f:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_roofline.py"), tag], stdout=f, check=False)
print(f"profiles/{tag}_kernel_roofline.md")
