#!/usr/bin/env python3
"""conv2's two backward kernels (weight gradient; fused data gradient + conv1 weight gradient) one after the other on
one stream vs side by side on two streams, at the benchmark geometry.  Grid caps come from the environment
(GDM_BD_CAP, GDM_BW_CAP: experiments only), so that co-residency on a CU (LDS: fused 54.8 KB, weight 34.8 KB per
workgroup, 160 KB per CU) can be arranged."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops, synthetic
from gan_des_midi_music_gen_amd.ops import BF16

def main():
    B, H, W = int(os.environ.get("B", 512)), 128, 256
    dev = "cuda"
    x = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
    w1 = (torch.randn(16, 1, 2, 2) * 0.1).to(dev); b1 = torch.full((16,), 2.0, device=dev)
    w2 = (torch.randn(32, 16, 3, 3) * 0.05).to(dev); b2 = torch.zeros(32, device=dev)
    p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, BF16)
    pack = ops.simnn_conv2_pack(w2, BF16)
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
    dp2 = torch.randn_like(p2.float()).to(torch.bfloat16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def seq():
        ops.simnn_conv2_bwd_weight(dp2, code2, p1)
        ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x)
    def par():
        main_s = torch.cuda.current_stream()
        s1.wait_stream(main_s); s2.wait_stream(main_s)
        with torch.cuda.stream(s1):
            ops.simnn_conv2_bwd_weight(dp2, code2, p1)
        with torch.cuda.stream(s2):
            ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x)
        main_s.wait_stream(s1); main_s.wait_stream(s2)
    def only_w(): ops.simnn_conv2_bwd_weight(dp2, code2, p1)
    def only_f(): ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x)
    def t(fn, n=30):
        for _ in range(5): fn()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        return statistics.median(ts)
    print(f"caps BD={os.environ.get('GDM_BD_CAP','dflt')} BW={os.environ.get('GDM_BW_CAP','dflt')} tag={os.environ.get('GDM_LIB_TAG','')}: "
          f"weight {t(only_w):6.1f}  fused {t(only_f):6.1f}  sequential {t(seq):6.1f}  side-by-side {t(par):6.1f} us")

if __name__ == "__main__":
    main()
