#!/usr/bin/env python3
"""Model 2: what bounds the replayed iteration -- the discriminator chain or the generator chains beside it?  Times
hipGraph replays of the step as bench.py builds it, then with the generators' forwards replaced by cached outputs (never
shipped: a measuring aid).  End of round 2: B = 256: 119.6 vs 121.5 us (the generators' graph is hidden completely);
B = 16: 91.4 vs 74.7 us (there the generators' five launches are the longer chain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import network_tests as NT, synthetic, ops
from gan_des_midi_music_gen_amd.train import MmganTrainer

B, T = int(os.environ.get("B", 256)), 50


def build():
    torch.manual_seed(0)
    dev = "cuda"
    mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, T), input_dim=50, output_dim=20, instrument=0,
                          start=100, end=100 + T, device=dev)
    mm.train()
    tr = MmganTrainer(mm, compute_dtype="bf16")
    d = synthetic.mmgan_inputs(B, T, seed=1234, device=dev)
    tr.capture(d["piano_roll"], d["durations"], d["beats"], d["noise1"], d["noise2"], d["fake_a"], d["fake_b"],
               d["g1_in_a"], d["g1_in_b"])
    return tr


def timed(tr, n=200):
    for _ in range(20): tr.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


print(f"graph                         : {timed(build()):8.1f} us/step")
orig = MmganTrainer._generators_forward_both       # (the trainers' fused path: both forwards of an iteration in one chain)
cache = {}
def cached(self, *a, **k):
    if "o" not in cache: cache["o"] = orig(self, *a, **k)
    return cache["o"]
MmganTrainer._generators_forward_both = cached
print(f"graph, generators cached      : {timed(build()):8.1f} us/step")
MmganTrainer._generators_forward_both = orig
