#!/usr/bin/env python3
"""Diagnostic (GDM_STAMPS build, library tag "stamps"): phases of model 1's one-launch optimizer step, workgroup 0
(small parameters, conv2 re-pack, next step's bias-correction terms, then its tile) beside the median tile workgroup.
s_memtime ticks at 100 MHz: 1 tick = 10 ns."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gan_des_midi_music_gen_amd import SIMNN, synthetic, train, _lib
from gan_des_midi_music_gen_amd.ops import BF16

def read():
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    lib.gdm_debug_read_stamps.restype = ctypes.c_int
    lib.gdm_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.gdm_debug_read_stamps(buf, 1024 * 8) == 0
    return np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)

B, H, W = 256, 128, 256
dev = "cuda"
torch.manual_seed(0)
gen, disc = SIMNN.Generator().to(dev), SIMNN.Discriminator(input_hw=(H, W)).to(dev)
tr = train.SimnnTrainer(gen, disc, compute_dtype=BF16)
real = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
fake = synthetic.spectrogram_batch(B, (H, W), seed=2, device=dev)
noise = torch.randn(B, 100, 1, 1, device=dev)
for _ in range(3):
    tr.step(real, noise, fake)
torch.cuda.synchronize()
for rep in range(3):
    tr._adam()
    torch.cuda.synchronize()
    st = read()
    names = ["terms", "small Adam", "re-pack", "next terms (pow)", "tile"]
    print("workgroup 0   : " + "  ".join(f"{n} {st[0, k] / 100:.1f} us" for k, n in enumerate(names)))
    med = np.median(st[1:, :], axis=0)
    print("median others : " + "  ".join(f"{n} {med[k] / 100:.1f} us" for k, n in enumerate(names)),
          " | tile min/max %.1f / %.1f us" % (st[1:, 4].min() / 100, st[1:, 4].max() / 100))
