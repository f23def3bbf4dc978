#!/usr/bin/env python3
"""Micro-benchmark of the three fc1 products of model 1 at the benchmark geometry (HIP events, median of N)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops
from gan_des_midi_music_gen_amd.ops import BF16, F32, ACT_RELU

def timeit(fn, n=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts), min(ts)

def main():
    B, K = int(os.environ.get("B", 512)), 65536
    dev = "cuda"
    torch.manual_seed(0)
    flat = torch.randn(B, K, device=dev).to(torch.bfloat16)             # p2, channels-last flatten
    wf1p = (torch.randn(128, K, device=dev) * 0.01).to(torch.bfloat16)  # permuted bf16 shadow of fc1.weight
    bf1 = torch.zeros(128, device=dev)
    dh1 = torch.randn(B, 128, device=dev)
    res = {}
    res["fc1 fwd  (B,K)x(K,128)"] = (timeit(lambda: ops.gemm(flat, wf1p.t(), bias_n=bf1, act=ACT_RELU, compute=BF16)), (B * K * 2 + 128 * K * 2) / 1e6)
    res["fc1 dW   (128,B)x(B,K)"] = (timeit(lambda: ops.gemm(dh1.t(), flat, compute=BF16)), (B * K * 2 + 128 * K * 4) / 1e6)
    res["fc1 dX   (B,128)x(128,K)"] = (timeit(lambda: ops.gemm(dh1, wf1p, compute=BF16, out_dtype=BF16)), (B * K * 2 + 128 * K * 2) / 1e6)
    for sk in (32, 64, 128, 256):
        res[f"fc1 fwd split_k={sk}"] = (timeit(lambda: ops.gemm(flat, wf1p.t(), bias_n=bf1, act=ACT_RELU, compute=BF16, split_k=sk)), (B * K * 2 + 128 * K * 2) / 1e6)
    dh16 = dh1.to(torch.bfloat16)
    res["fc1 dW, bf16 dh"] = (timeit(lambda: ops.gemm(dh16.t(), flat, compute=BF16)), (B * K * 2 + 128 * K * 4) / 1e6)
    res["fc1 dX, bf16 dh"] = (timeit(lambda: ops.gemm(dh16, wf1p, compute=BF16, out_dtype=BF16)), (B * K * 2 + 128 * K * 2) / 1e6)
    for k, ((med, mn), mb) in res.items():
        print(f"{k:28s} B={B} median {med:8.1f} us  min {mn:8.1f} us   {mb:6.1f} MB -> {mb / med:5.2f} TB/s")

if __name__ == "__main__":
    main()
