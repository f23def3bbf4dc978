#!/usr/bin/env python3
"""Throughput of the device mel-spectrogram featuriser (util.get_melspectrogram_db_tensor) on B 5-second windows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import util

B = int(os.environ.get("B", 256))
x = (torch.randn(B, 5 * 44100) * 0.1).cuda()
for _ in range(3):
    util.get_melspectrogram_db_tensor(x)
torch.cuda.synchronize(); t = time.perf_counter()
n = 10
for _ in range(n):
    out = util.get_melspectrogram_db_tensor(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
flops = B * 216 * (2048 * 2050 * 2 + 1028 * 128 * 2)
print(f"mel featuriser: B={B} {dt * 1e3:.2f} ms/batch = {B / dt:.0f} windows/s, {flops / dt / 1e12:.1f} TFLOP/s fp32 "
      f"(exact-fp32 MFMA peak 157); output {tuple(out.shape)}")
