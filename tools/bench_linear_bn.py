#!/usr/bin/env python3
"""Model 2's generator blocks alone: one Linear+BatchNorm1d+Sigmoid launch per layer shape, timed as a dependent chain
inside a hipGraph (how the trainer runs them) -- us per launch, against the same chain of one-workgroup no-op kernels
would be ideal; here against a chain of tiny fill kernels as the launch-latency floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops
from gan_des_midi_music_gen_amd.ops import ACT_SIGMOID

dev, B = "cuda", int(os.environ.get("B", 256))
dims = [100, 256, 128, 64, 4096]
torch.manual_seed(0)
ws = [torch.randn(dims[i + 1], dims[i], device=dev) * 0.1 for i in range(4)]
ps = [[torch.zeros(d, device=dev), torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev),
       torch.ones(d, device=dev)] for d in dims[1:]]
nbt = torch.zeros((), dtype=torch.long, device=dev)
x0 = torch.randn(B, dims[0], device=dev)


xin = [torch.randn(B, d, device=dev) for d in dims[:4]]


def chain(layers, act=ACT_SIGMOID, training=True):
    x = x0
    for i in layers:
        b, g, be, rm, rv = ps[i]
        x = ops.linear_bn_act_fwd(x if x.shape[1] == dims[i] else xin[i], ws[i], b, g, be, rm, rv, nbt, act=act,
                                  training=training)[0]
    return x


def fills(n):
    t = torch.zeros(64, device=dev)
    for _ in range(n):
        t.add_(1.0)
    return t


def graph_time(fn, reps=200):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


print(f"4-layer generator chain        : {graph_time(lambda: chain([0, 1, 2, 3])):7.1f} us  (4 launches)")
for i in range(4):
    print(f"  layer {i} ({dims[i]:4d} -> {dims[i + 1]:4d}) x 8     : {graph_time(lambda: [chain([i]) for _ in range(8)]) / 8:7.1f} us per launch")
from gan_des_midi_music_gen_amd.ops import ACT_NONE
print(f"  layer 1, eval-mode statistics   : {graph_time(lambda: [chain([1], training=False) for _ in range(8)]) / 8:7.1f} us per launch")
print(f"  layer 1, no activation          : {graph_time(lambda: [chain([1], act=ACT_NONE) for _ in range(8)]) / 8:7.1f} us per launch")
print(f"8 dependent 64-element adds    : {graph_time(lambda: fills(8)) / 8:7.1f} us per launch")
