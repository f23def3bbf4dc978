import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_des_midi_music_gen_amd import ops
from gan_des_midi_music_gen_amd.ops import BF16, F32
for dt in (F32, BF16):
    for (b, h, w) in [(2, 128, 216), (3, 128, 256), (2, 16, 24), (1, 10, 300), (2, 22, 30), (40, 128, 256)]:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(b, h, w, generator=g) * 18 - 35).clamp(-80, 30).cuda()
        w1 = (torch.randn(16, 1, 2, 2, generator=g) * 0.1).cuda(); b1 = (torch.randn(16, generator=g) * 0.5 + 2).cuda()
        w2 = (torch.randn(32, 16, 3, 3, generator=g) * 0.05).cuda(); b2 = (torch.randn(32, generator=g) * 0.1).cuda()
        p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, dt)
        pack = ops.simnn_conv2_pack(w2, dt)
        p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
        up = torch.randn(p2.shape, generator=g).cuda().to(p2.dtype)
        dp1 = ops.simnn_conv2_bwd_data(up, code2, pack, p1.shape[1], p1.shape[2])
        dw1, db1 = ops.simnn_conv1_bwd_weight(dp1, code1, x)
        dw1f, db1f, dp1f = ops.simnn_conv2_bwd_fused(up, code2, pack, code1, x, want_dp1=True)
        torch.cuda.synchronize()
        print(dt, (b, h, w), "dp1 equal", torch.equal(dp1, dp1f), "dw rel err",
              ((dw1 - dw1f).abs().max() / dw1.abs().max()).item(), "db rel err",
              ((db1 - db1f).abs().max() / db1.abs().max()).item())
