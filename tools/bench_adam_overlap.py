#!/usr/bin/env python3
"""Would the HBM-bound Adam on fc1.weight hide behind the issue-bound convolution forwards of the next iteration?
conv1 forward (real), conv1 forward (fake), conv2 forward (2B) on one stream; gdm_adam_step_dev_pc on another."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops, synthetic
from gan_des_midi_music_gen_amd.ops import BF16

def main():
    B, H, W = 256, 128, 256
    dev = "cuda"
    real = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
    fake = synthetic.spectrogram_batch(B, (H, W), seed=2, device=dev)
    w1 = (torch.randn(16, 1, 2, 2) * 0.1).to(dev); b1 = torch.full((16,), 2.0, device=dev)
    w2 = (torch.randn(32, 16, 3, 3) * 0.05).to(dev); b2 = torch.zeros(32, device=dev)
    pack = ops.simnn_conv2_pack(w2, BF16)
    p1 = torch.empty((2 * B, 64, 128, 16), dtype=torch.bfloat16, device=dev)
    code1 = torch.empty((2 * B, 64, 128), dtype=torch.int64, device=dev)
    n, c, pix = 128, 32, 2048
    p = torch.randn(n * c * pix, device=dev); g = torch.randn(n * c * pix, device=dev) * 1e-3
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    shadow = torch.empty((n, pix, c), dtype=torch.bfloat16, device=dev)
    hyper = ops.adam_hyper(torch.device(dev), 2e-5, 0.5, 0.999, 1e-8, 1.0)
    side = torch.cuda.Stream()
    def convs():
        ops.simnn_conv1_fwd(real, w1, b1, BF16, out=(p1[:B], code1[:B]))
        ops.simnn_conv1_fwd(fake, w1, b1, BF16, out=(p1[B:], code1[B:]))
        ops.simnn_conv2_fwd(p1, pack, b2)
    def adam(): ops.adam_step_dev_pc(p, g, m, v, n, c, pix, shadow, hyper, advance_step=True)
    def seq(): adam(); convs()
    def par():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side): adam()
        convs()
        main_s.wait_stream(side)
    def t(fn, n=30):
        for _ in range(5): fn()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        return statistics.median(ts)
    with torch.cuda.stream(side): adam()
    torch.cuda.synchronize()
    print(f"convs {t(convs):.1f} us   adam {t(adam):.1f} us   one after the other {t(seq):.1f} us   side by side {t(par):.1f} us")

if __name__ == "__main__":
    main()
