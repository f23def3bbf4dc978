#!/usr/bin/env python3
"""How much of the independent branches is hidden in the captured iteration?  Times hipGraph replays of the model-1
step as built by bench.py, then with single branches replaced by no-ops (never shipped: a measuring aid).  The kernel
trace of a replayed graph serialises branches under rocprofv3, so overlap has to be inferred from wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import SIMNN, synthetic, functional as Fn, ops
from gan_des_midi_music_gen_amd.train import SimnnTrainer

def build(overlap=True, pipelined=False):
    torch.manual_seed(0)
    dev = "cuda"
    gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
    disc = SIMNN.Discriminator(input_hw=(128, 256)).apply(SIMNN.weights_init).to(dev)
    tr = SimnnTrainer(gen, disc, compute_dtype="bf16", overlap=overlap)
    real, fake, noise = synthetic.simnn_inputs(256, (128, 256), seed=1234, device=dev)
    tr.capture(real, noise, fake, pipelined=pipelined)
    return tr

def timed(tr, n=200):
    for _ in range(20): tr.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

print(f"graph, pipelined schedule : {timed(build(True, True)):8.1f} us/step")
print(f"graph, overlap            : {timed(build(True)):8.1f} us/step")
print(f"graph, single stream      : {timed(build(False)):8.1f} us/step")
orig_gen = Fn.simnn_gen_forward
cache = {}
def fake_gen(noise, ws, bns, training, dt, **kw):
    if "o" not in cache: cache["o"] = orig_gen(noise, ws, bns, training, dt, **kw)
    return cache["o"]
Fn.simnn_gen_forward = fake_gen
print(f"graph, overlap, no G fwd  : {timed(build(True)):8.1f} us/step")
cache.clear()
print(f"graph, pipelined, no G fwd: {timed(build(True, True)):8.1f} us/step")
Fn.simnn_gen_forward = orig_gen
orig_bw = ops.simnn_conv2_bwd_weight
ops.simnn_conv2_bwd_weight = lambda dp2, code2, p1, out=None: out
print(f"graph, overlap, no conv2 dW: {timed(build(True)):8.1f} us/step")
ops.simnn_conv2_bwd_weight = orig_bw
