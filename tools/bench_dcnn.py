import sys, os, statistics
sys.path.insert(0, "/root/repo")
import torch
from gan_des_midi_music_gen_amd import ops, network_tests as NT, synthetic
from gan_des_midi_music_gen_amd.train import MmganTrainer
def timeit(fn, n=40):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts), min(ts)
dev = "cuda"
torch.manual_seed(0)
mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, instrument=0, start=100, end=150, device=dev)
mm.train()
tr = MmganTrainer(mm, compute_dtype="bf16")
w1, b1, w2, b2, wf, bf = tr.d.views
pack = ops.dcnn_pack(w1, b1, w2, b2, wf, bf, 50)
loss = torch.zeros(1, device=dev)
for B in (64, 128, 256, 512):
    xa = torch.rand(B, 2, 128, 50, device=dev)
    p0 = torch.rand(B, 128, 50, device=dev); p1 = torch.rand(B, 128, 50, device=dev)
    g = [torch.empty_like(v) for v in tr.d.grad_views]
    print(B, "xa only, grad :", timeit(lambda: ops.dcnn_fused(xa, None, 50, 1.0, 1.0, pack, loss_out=loss, grad_out=g)))
    print(B, "xa only, nograd:", timeit(lambda: ops.dcnn_fused(xa, None, 50, 1.0, 1.0, pack, loss_out=loss, want_grad=False)))
    print(B, "xa+planes grad :", timeit(lambda: ops.dcnn_fused(xa, (p0, p1), 50, 0.0, 1.0, pack, loss_out=loss, grad_out=g)))
