#!/usr/bin/env python3
"""Do two kernels on two HIP streams really run at the same time?  A 64-workgroup launch of conv2's weight-gradient
kernel (GDM_BW_CAP=64: a quarter of the CUs, so two of them cannot compete for resources) alone, twice on one stream,
and once on each stream of several stream pairs.  Concurrent = the time of one; serialised = the time of two."""
import os, sys, statistics
os.environ.setdefault("GDM_BW_CAP", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops, synthetic
from gan_des_midi_music_gen_amd.ops import BF16

def main():
    B, H, W = 128, 128, 256
    dev = "cuda"
    x = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
    w1 = (torch.randn(16, 1, 2, 2) * 0.1).to(dev); b1 = torch.full((16,), 2.0, device=dev)
    w2 = (torch.randn(32, 16, 3, 3) * 0.05).to(dev); b2 = torch.zeros(32, device=dev)
    p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, BF16)
    pack = ops.simnn_conv2_pack(w2, BF16)
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
    dp2 = torch.randn_like(p2.float()).to(torch.bfloat16)
    streams = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1)]
    outs = [(torch.empty((32, 16, 3, 3), device=dev), torch.empty(32, device=dev)) for _ in range(2)]
    def k(i): ops.simnn_conv2_bwd_weight(dp2, code2, p1, out=outs[i])
    def t(fn, n=20):
        for _ in range(3): fn()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        return statistics.median(ts)
    # warm the per-stream workspaces
    for s in streams:
        with torch.cuda.stream(s): k(0)
    torch.cuda.synchronize()
    one = t(lambda: k(0)); two = t(lambda: (k(0), k(1)))
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES','dflt')}: one launch {one:.1f} us, two on one stream {two:.1f} us")
    main_s = torch.cuda.current_stream()
    for i, j in ((0, 1), (0, 2), (1, 2), (2, 3), (0, 4), (3, 5), (0, 6)):
        def par():
            streams[i].wait_stream(main_s); streams[j].wait_stream(main_s)
            with torch.cuda.stream(streams[i]): k(0)
            with torch.cuda.stream(streams[j]): k(1)
            main_s.wait_stream(streams[i]); main_s.wait_stream(streams[j])
        print(f"  streams ({i},{j}): {t(par):.1f} us")
    def par_main():
        streams[0].wait_stream(main_s)
        with torch.cuda.stream(streams[0]): k(0)
        k(1)
        main_s.wait_stream(streams[0])
    print(f"  main + stream 0: {t(par_main):.1f} us")

if __name__ == "__main__":
    main()
