# per-phase shader-clock spans of the model-2 discriminator kernel (needs the -DGDM_DCNN_STAMPS variant library:
#   GDM_BUILD_TAG=dstamps GDM_HIPCC_FLAGS="-DGDM_DCNN_STAMPS" python -m gan_des_midi_music_gen_amd.build)
GDM_LIB_TAG=dstamps python - <<'PY' 2>&1 | grep -v Warn | tail -4
import sys; sys.path.insert(0, ".")
import torch
from gan_des_midi_music_gen_amd import ops, network_tests as NT
from gan_des_midi_music_gen_amd.train import MmganTrainer
dev = "cuda"
torch.manual_seed(0)
mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, instrument=0, start=100, end=150, device=dev)
mm.train()
tr = MmganTrainer(mm, compute_dtype="bf16")
w1, b1, w2, b2, wf, bf = tr.d.views
pack = ops.dcnn_pack(w1, b1, w2, b2, wf, bf, 50)
loss = torch.zeros(1, device=dev)
B = 512
xa = torch.rand(B, 2, 128, 50, device=dev)
g = [torch.empty_like(v) for v in tr.d.grad_views]
for _ in range(2):
    ops.dcnn_fused(xa, None, 50, 1.0, 1.0, pack, loss_out=loss, grad_out=g)
    torch.cuda.synchronize()
print("nograd:")
ops.dcnn_fused(xa, None, 50, 1.0, 1.0, pack, loss_out=loss, want_grad=False)
torch.cuda.synchronize()
PY
