#!/usr/bin/env python3
"""Micro-benchmark of single C-ABI entry points at the benchmark geometry (HIP events, median of N)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops, synthetic
from gan_des_midi_music_gen_amd.ops import BF16

def timeit(fn, n=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts), min(ts)

def main():
    B, H, W = int(os.environ.get("B", 512)), 128, 256
    dev = "cuda"
    torch.manual_seed(0)
    x = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
    w1 = (torch.randn(16, 1, 2, 2) * 0.1).to(dev); b1 = torch.full((16,), 2.0, device=dev)
    w2 = (torch.randn(32, 16, 3, 3) * 0.05).to(dev); b2 = torch.zeros(32, device=dev)
    p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, BF16)
    pack = ops.simnn_conv2_pack(w2, BF16)
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
    dp2 = torch.randn_like(p2.float()).to(torch.bfloat16)
    res = {}
    only = os.environ.get("ONLY")              # one op by name
    fns = {"conv1_fwd": lambda: ops.simnn_conv1_fwd(x, w1, b1, BF16),
           "conv2_fwd": lambda: ops.simnn_conv2_fwd(p1, pack, b2),
           "conv2_bwd_fused": lambda: ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x),
           "conv2_bwd_data": lambda: ops.simnn_conv2_bwd_data(dp2, code2, pack, p1.shape[1], p1.shape[2]),
           "conv2_bwd_weight": lambda: ops.simnn_conv2_bwd_weight(dp2, code2, p1)}
    for k, fn in fns.items():
        if only is None or k == only:
            res[k] = timeit(fn)
    for k, (med, mn) in res.items():
        print(f"{k:20s} B={B} median {med:8.1f} us  min {mn:8.1f} us")

if __name__ == "__main__":
    main()
