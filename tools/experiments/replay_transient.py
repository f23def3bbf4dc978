#!/usr/bin/env python3
"""How long is the start-up transient of the replayed iteration?  Device time of each of the first N replays after set-up
(HIP events between replays), model 1 default configuration.  tools/experiments/replay_transient.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_des_midi_music_gen_amd import SIMNN, synthetic
from gan_des_midi_music_gen_amd.train import SimnnTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
disc = SIMNN.Discriminator(input_hw=(128, 256)).apply(SIMNN.weights_init).to(dev)
tr = SimnnTrainer(gen, disc, compute_dtype="bf16")
real, fake, noise = synthetic.simnn_inputs(256, (128, 256), seed=1234, device=dev)
tr.capture(real, noise, fake, pipelined=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    tr.replay()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("ms per replay:", " ".join(f"{t:.3f}" for t in ts))
print("first 5: %.3f  5-20: %.3f  20-40: %.3f  last 16: %.3f" % (sum(ts[:5]) / 5, sum(ts[5:20]) / 15, sum(ts[20:40]) / 20, sum(ts[-16:]) / 16))
