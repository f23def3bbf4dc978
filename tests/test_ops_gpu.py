"""GPU: every C-ABI kernel against plain PyTorch fp32 CPU math on the same seeded inputs.

fp32 mode (v_mfma_f32_16x16x4_f32) is held to summation-order noise (rtol 2e-5 of the result scale); bf16 mode is
compared with a reference evaluated on the same bf16-rounded operands, so what is left is accumulation order and
the rounding of stored bf16 outputs (rtol 1e-2 = 2.5 bf16 ulps).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_des_midi_music_gen_amd import ops  # noqa: E402
from gan_des_midi_music_gen_amd.ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F32  # noqa: E402

DEV = "cuda"


def _close(got, want, rtol, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rtol {rtol})"


def _act(x, act, slope=0.2):
    return {ACT_NONE: lambda v: v, ACT_RELU: torch.relu, ACT_LEAKY: lambda v: F.leaky_relu(v, slope),
            ACT_SIGMOID: torch.sigmoid}[act](x)


def _rb(x):  # round through bf16
    return x.to(torch.bfloat16).float()


def test_gemm_identity_asymmetric():
    # A = I with an asymmetric B catches a transposed C/D fragment map
    n = 64
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    for comp in (F32, BF16):
        out = ops.gemm(torch.eye(n, device=DEV), b.to(DEV), compute=comp)
        _close(out, _rb(b) if comp == BF16 else b, 1e-6 if comp == F32 else 1e-6, f"identity comp={comp}")


@pytest.mark.parametrize("comp", [F32, BF16])
@pytest.mark.parametrize("m,n,k", [(16, 256, 100), (37, 20, 64), (256, 128, 4096), (128, 1000, 70), (5, 1, 128),
                                   (100, 2048, 100)])
def test_gemm_shapes_and_epilogues(comp, m, n, k):
    g = torch.Generator().manual_seed(m * 1000 + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5   # Linear weight (N,K): b = w.t()
    bias = torch.randn(n, generator=g)
    ar, wr = (_rb(a), _rb(w)) if comp == BF16 else (a, w)
    rt = 2e-5 if comp == F32 else 2e-3
    for act in (ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID):
        out = ops.gemm(a.to(DEV), w.to(DEV).t(), bias_n=bias.to(DEV), act=act, slope=0.2, compute=comp)
        _close(out, _act(ar @ wr.t() + bias, act), rt, f"linear act={act}")
    # split-K (deterministic slabs) and transposed-A (dW form): C = a^T @ a2
    a2 = torch.randn(m, n, generator=g)
    out = ops.gemm(a.to(DEV).t(), a2.to(DEV), compute=comp, split_k=3)
    _close(out, ar.t() @ (_rb(a2) if comp == BF16 else a2), rt * 2, "dW form split_k=3")
    out2 = ops.gemm(a.to(DEV).t(), a2.to(DEV), compute=comp, split_k=3)
    assert torch.equal(out, out2), "split-K result must be bit-reproducible"
    # bf16 storage operands / outputs
    ab = a.to(DEV).to(torch.bfloat16)
    out = ops.gemm(ab, w.to(DEV).t(), bias_n=bias.to(DEV), compute=comp, out_dtype=BF16)
    _close(out, _rb(a) @ wr.t() + bias, 1e-2, "bf16 in/out")
    # bias along M
    bm = torch.randn(m, generator=g)
    out = ops.gemm(a.to(DEV), w.to(DEV).t(), bias_m=bm.to(DEV), compute=comp)
    _close(out, ar @ wr.t() + bm[:, None], rt, "bias_m")


@pytest.mark.parametrize("ta", ["f32", "bf16"])
@pytest.mark.parametrize("tb", ["f32", "bf16"])
@pytest.mark.parametrize("a_kmaj", [True, False])
@pytest.mark.parametrize("b_kmaj", [True, False])
def test_gemm_bf16_vector_path_all_layouts(ta, tb, a_kmaj, b_kmaj):
    """The 128x128 vectorised kernel: every (dtype, K-major / row-major) operand combination, ragged M/N/K tails,
    fused epilogue, split-K, bf16 output."""
    g = torch.Generator().manual_seed(17)
    m, n, k = 200, 264, 136
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g) / k ** 0.5
    bias = torch.randn(n, generator=g)
    tda = torch.float32 if ta == "f32" else torch.bfloat16
    tdb = torch.float32 if tb == "f32" else torch.bfloat16
    # physical layouts: K-major A = (m,k) contiguous; row-major A = stored as (k,m), viewed transposed
    ad = a.to(tda).to(DEV).contiguous() if a_kmaj else a.t().contiguous().to(tda).to(DEV).t()
    bd = b.t().contiguous().to(tdb).to(DEV).t() if b_kmaj else b.to(tdb).to(DEV).contiguous()
    ref = _rb(a) @ _rb(b)
    for split in (1, 3):
        out = ops.gemm(ad, bd, bias_n=bias.to(DEV), act=ACT_LEAKY, slope=0.2, compute=BF16, split_k=split)
        _close(out, F.leaky_relu(ref + bias, 0.2), 2e-3, f"fast gemm split={split}")
    out = ops.gemm(ad, bd, compute=BF16, out_dtype=BF16)
    _close(out, ref, 1e-2, "fast gemm bf16 out")
    if b_kmaj:   # N not a multiple of 4 -> scalar epilogue (only reachable with a K-major B)
        bd2 = b[:, :262].t().contiguous().to(tdb).to(DEV).t()
        out = ops.gemm(ad, bd2, bias_n=bias[:262].to(DEV), compute=BF16)
        _close(out, ref[:, :262] + bias[:262], 2e-3, "fast gemm ragged N")


@pytest.mark.parametrize("ta", ["f32", "bf16"])
@pytest.mark.parametrize("out_dt", [F32, BF16])
@pytest.mark.parametrize("m,n", [(200, 1096), (512, 4096), (64, 1024)])
def test_gemm_k128_row_major_b(ta, out_dt, m, n):
    """fc1's data-gradient shape: K == 128, K-major A, row-major bf16 B, no epilogue; ragged M and N tails (N % 16 == 8
    exercises the row-contiguous epilogue with half-valid lanes)."""
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, 128, generator=g)
    b = torch.randn(128, n, generator=g) / 128 ** 0.5
    ad = a.to(DEV) if ta == "f32" else a.to(torch.bfloat16).to(DEV)
    bd = b.to(torch.bfloat16).to(DEV)
    out = ops.gemm(ad, bd, compute=BF16, out_dtype=out_dt)
    _close(out, _rb(a) @ _rb(b), 2e-3 if out_dt == F32 else 1e-2, "k128 gemm")
    out2 = ops.gemm(ad, bd, compute=BF16, out_dtype=out_dt)
    assert torch.equal(out, out2)


def test_bce_with_logits():
    g = torch.Generator().manual_seed(3)
    for n, target in ((16, 0.9), (256, 0.1), (2048, 1.0), (7, 0.0)):
        x = torch.randn(n, generator=g) * 3
        xr = x.clone().requires_grad_(True)
        loss = F.binary_cross_entropy_with_logits(xr, torch.full((n,), target))
        loss.backward()
        l, dx = ops.bce_with_logits(x.to(DEV), target)
        assert abs(l.item() - loss.item()) < 2e-6 * max(1, abs(loss.item()))
        _close(dx, xr.grad, 1e-5, "bce dx")
        # model 1: sigmoid output fed to the logits loss, gradient chained to the pre-sigmoid value
        z = torch.randn(n, generator=g).requires_grad_(True)
        p = torch.sigmoid(z)
        loss = F.binary_cross_entropy_with_logits(p, torch.full((n,), target))
        loss.backward()
        l, dz = ops.bce_with_logits(p.detach().to(DEV), target, fuse_sigmoid_backward=True)
        assert abs(l.item() - loss.item()) < 2e-6
        _close(dz, z.grad, 1e-5, "bce dz (fused sigmoid)")


@pytest.mark.parametrize("n,n0,y0,y1", [(4, 2, 0.9, 0.1), (512, 256, 0.9, 0.1), (256, 256, 1.0, 1.0), (77, 30, 0.0, 1.0)])
def test_simnn_head_fused_forward_loss_backward(n, n0, y0, y1):
    g = torch.Generator().manual_seed(n)
    h_pre = torch.randn(n, 128, generator=g, requires_grad=True)
    w2 = (torch.randn(1, 128, generator=g) * 0.2).requires_grad_(True)
    b2 = torch.randn(1, generator=g).requires_grad_(True)
    h1 = torch.relu(h_pre)
    p = torch.sigmoid(F.linear(h1, w2, b2)).reshape(-1)
    loss = F.binary_cross_entropy_with_logits(p[:n0], torch.full((n0,), y0))
    if n0 < n:
        loss = loss + F.binary_cross_entropy_with_logits(p[n0:], torch.full((n - n0,), y1))
    loss.backward()
    lo = torch.full((1,), 123.0, device=DEV)
    prob, dh1, (dw2, db2, db1) = ops.simnn_head(h1.detach().to(DEV), w2.detach().to(DEV), b2.detach().to(DEV), n0, y0,
                                                y1, loss_out=lo)
    assert abs(lo.item() - loss.item()) < 2e-6 * max(1, abs(loss.item()))
    _close(prob, p, 1e-6, "head prob")
    _close(dh1, h_pre.grad, 2e-5, "head dh1 (through ReLU)")
    _close(dw2, w2.grad, 2e-5, "head dw2")
    _close(db2, b2.grad, 2e-5, "head db2")
    _close(db1, h_pre.grad.sum(0), 2e-5, "head db1")
    lo2 = torch.full((1,), 1.5, device=DEV)
    ops.simnn_head(h1.detach().to(DEV), w2.detach().to(DEV), b2.detach().to(DEV), n0, y0, y1, loss_out=lo2,
                   accumulate_loss=True, want_grad=False)
    assert abs(lo2.item() - 1.5 - loss.item()) < 1e-5


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(4)
    for n, lr, betas in ((1000, 2e-5, (0.5, 0.999)), (21041, 0.01, (0.9, 0.999)), (7, 1e-3, (0.9, 0.99))):
        p0 = torch.randn(n, generator=g)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=lr, betas=betas)
        p = p0.clone().to(DEV)
        m = torch.zeros(n, device=DEV)
        v = torch.zeros(n, device=DEV)
        for step in range(1, 6):
            grad = torch.randn(n, generator=g) * (10.0 ** (-step))
            ref.grad = grad.clone()
            opt.step()
            ops.adam_step(p, grad.to(DEV), m, v, step, lr, betas[0], betas[1], 1e-8)
            err = (p.cpu() - ref.detach()).abs().max().item()
            # one ulp of a parameter of magnitude < 8 is 4.8e-7: the update itself agrees far tighter than that
            assert err <= 6e-7, f"adam step {step}: max |dp| {err:.3e}"
            upd_err = ((p.cpu() - p0) - (ref.detach() - p0)).abs().max().item()
            assert upd_err <= 6e-7 + 1e-4 * lr * step, f"adam step {step}: update err {upd_err:.3e}"


@pytest.mark.parametrize("n,c,pix,sdt", [(128, 32, 2048, torch.bfloat16), (3, 32, 100, torch.float32),
                                         (2, 20, 37, torch.bfloat16), (5, 32, 128, torch.bfloat16)])
def test_adam_with_permuted_gradient_and_operand_copy(n, c, pix, sdt):
    """gdm_adam_step_dev_pc == gdm_adam_step_dev on the un-permuted gradient, bit for bit, and the operand copy it
    writes is the updated parameter transposed to (n, pix, c) in the copy's dtype (model 1's fc1.weight)."""
    g = torch.Generator().manual_seed(n * pix + c)
    p0 = torch.randn(n, c, pix, generator=g).to(DEV)
    p_a, p_b = p0.clone(), p0.clone()
    m_a, v_a = torch.zeros_like(p0), torch.zeros_like(p0)
    m_b, v_b = torch.zeros_like(p0), torch.zeros_like(p0)
    shadow = torch.empty((n, pix, c), dtype=sdt, device=DEV)
    h_a = ops.adam_hyper(torch.device(DEV), 2e-5, 0.5, 0.999, 1e-8, 0.5)
    h_b = ops.adam_hyper(torch.device(DEV), 2e-5, 0.5, 0.999, 1e-8, 0.5)
    for step in range(3):
        g_pc = (torch.randn(n, pix, c, generator=g) * 10.0 ** (-step)).to(DEV)
        ops.adam_step_dev(p_a.view(-1), g_pc.permute(0, 2, 1).contiguous().view(-1), m_a.view(-1), v_a.view(-1), h_a)
        ops.adam_step_dev_pc(p_b.view(-1), g_pc.view(-1), m_b.view(-1), v_b.view(-1), n, c, pix, shadow, h_b,
                             advance_step=True)
        assert torch.equal(p_a, p_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b), step
        assert torch.equal(shadow, p_b.permute(0, 2, 1).contiguous().to(sdt)), step
    assert torch.equal(h_a, h_b)
    # a misaligned parameter slot takes the scalar path: same result
    buf = torch.zeros(n * c * pix + 1, device=DEV)
    pu = buf[1:]
    pu.copy_(p0.view(-1))
    mu, vu = torch.zeros(n * c * pix + 1, device=DEV)[1:], torch.zeros(n * c * pix + 1, device=DEV)[1:]
    h_u = ops.adam_hyper(torch.device(DEV), 2e-5, 0.5, 0.999, 1e-8, 0.5)
    g_pc = torch.randn(n, pix, c, generator=torch.Generator().manual_seed(1)).to(DEV)
    ops.adam_step_dev_pc(pu, g_pc.view(-1), mu, vu, n, c, pix, shadow, h_u, advance_step=True)
    p_c, m_c, v_c = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    h_c = ops.adam_hyper(torch.device(DEV), 2e-5, 0.5, 0.999, 1e-8, 0.5)
    ops.adam_step_dev(p_c.view(-1), g_pc.permute(0, 2, 1).contiguous().view(-1), m_c.view(-1), v_c.view(-1), h_c)
    assert torch.equal(pu, p_c.view(-1))


@pytest.mark.parametrize("rows,c,act", [(16, 256, ACT_SIGMOID), (4, 4096, ACT_SIGMOID), (256, 20, ACT_SIGMOID),
                                        (16 * 64, 64, ACT_RELU), (5000, 32, ACT_RELU), (65536, 32, ACT_RELU),
                                        (4096, 128, ACT_RELU), (777, 8, ACT_RELU)])
def test_batchnorm_act_fwd_bwd(rows, c, act):
    g = torch.Generator().manual_seed(rows + c)
    y = torch.randn(rows, c, generator=g) * 2 + 0.5
    gamma = torch.randn(c, generator=g)
    beta = torch.randn(c, generator=g)
    bn = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    yr = y.clone().requires_grad_(True)
    out_ref = _act(bn(yr), act)
    r = torch.randn(rows, c, generator=g)
    (out_ref * r).sum().backward()
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    out, mean, invstd = ops.bn_act_fwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), rm, rv, nbt, act=act)
    _close(out, out_ref, 2e-5, "bn out")
    _close(rm, bn.running_mean, 2e-5, "running_mean")
    _close(rv, bn.running_var, 2e-5, "running_var")
    assert int(nbt.item()) == 1
    dy, dgamma, dbeta = ops.bn_act_bwd(r.to(DEV), out, y.to(DEV), gamma.to(DEV), mean, invstd, act=act)
    _close(dy, yr.grad, 2e-4, "bn dy")
    _close(dgamma, bn.weight.grad, 1e-4, "bn dgamma")
    _close(dbeta, bn.bias.grad, 1e-4, "bn dbeta")
    # eval mode uses the running statistics and leaves them alone
    bn.eval()
    out_e, _, _ = ops.bn_act_fwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), rm, rv, nbt, act=act, training=False)
    _close(out_e, _act(bn(y), act), 2e-5, "bn eval")
    assert int(nbt.item()) == 1


def test_pointwise_helpers():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(300, 77, generator=g)
    bias = torch.randn(77, generator=g)
    for act in (ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID):
        out = ops.bias_act_fwd(x.to(DEV), bias.to(DEV), act=act, slope=0.2)
        ref = _act(x + bias, act)
        _close(out, ref, 1e-6, f"bias_act {act}")
        d = torch.randn(300, 77, generator=g)
        xr = x.clone().requires_grad_(True)
        (_act(xr + bias, act) * d).sum().backward()
        dx = ops.act_bwd(d.to(DEV), out, act=act, slope=0.2)
        _close(dx, xr.grad, 1e-5, f"act_bwd {act}")
    _close(ops.colsum(x.to(DEV)), x.sum(0), 1e-5, "colsum")
    big = torch.randn(20000, 16, generator=g)
    _close(ops.colsum(big.to(DEV)), big.sum(0), 1e-4, "colsum big")
    _close(ops.cast(x.to(DEV), BF16), x.to(torch.bfloat16), 0.0, "cast")


def _disc_trunk_ref(x, w1, b1, w2, b2):
    a1 = F.max_pool2d(torch.relu(F.conv2d(x.unsqueeze(1), w1, b1, padding=1)), 2, 2)
    a2 = F.max_pool2d(torch.relu(F.conv2d(a1, w2, b2, padding=1)), 2, 2)
    return a1, a2


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("b,h,w", [(2, 128, 216), (3, 128, 256), (2, 16, 24), (1, 10, 300), (2, 22, 30),
                                   (1, 3, 4), (2, 31, 65), (1, 64, 130), (3, 12, 258), (1, 9, 513)])
def test_simnn_conv_trunk_forward_backward(dt, b, h, w):
    g = torch.Generator().manual_seed(h * w + b)
    x = (torch.randn(b, h, w, generator=g) * 18 - 35).clamp(-80, 30)
    w1 = (torch.randn(16, 1, 2, 2, generator=g) * 0.1).requires_grad_(True)
    b1 = (torch.randn(16, generator=g) * 0.5 + 2.0).requires_grad_(True)
    w2 = (torch.randn(32, 16, 3, 3, generator=g) * 0.05).requires_grad_(True)
    b2 = (torch.randn(32, generator=g) * 0.1).requires_grad_(True)
    a1, a2 = _disc_trunk_ref(x, w1, b1, w2, b2)
    rt = 3e-5 if dt == F32 else 2e-2
    xd = x.to(DEV)
    p1, code1 = ops.simnn_conv1_fwd(xd, w1.detach().to(DEV), b1.detach().to(DEV), dt)
    _close(p1.float().permute(0, 3, 1, 2), a1, 1e-5 if dt == F32 else 1e-2, "conv1+relu+pool")
    pack = ops.simnn_conv2_pack(w2.detach().to(DEV), dt)
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2.detach().to(DEV))
    assert code2.shape == p2.shape[:3] + (16,) and int(code2.max().item()) <= 8 * 24 and not bool((code2 % 8).any())
    up = torch.randn(a2.shape, generator=g)
    if dt == F32:
        _close(p2.permute(0, 3, 1, 2), a2, rt, "conv2+relu+pool")
        a1.retain_grad()
        (a2 * up).sum().backward()
        ref_dp1, ref_dw2, ref_db2, ref_dw1, ref_db1 = a1.grad, w2.grad, b2.grad, w1.grad, b1.grad
    else:
        # continue the reference from the bf16-rounded activations / weights / gradients the kernels consume
        a1r = p1.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        w2r = _rb(w2.detach()).requires_grad_(True)
        b2r = b2.detach().clone().requires_grad_(True)
        a2b = F.max_pool2d(torch.relu(F.conv2d(a1r, w2r, b2r, padding=1)), 2, 2)
        _close(p2.permute(0, 3, 1, 2), a2b, rt, "conv2+relu+pool")
        (a2b * _rb(up)).sum().backward()
        ref_dp1, ref_dw2, ref_db2 = a1r.grad, w2r.grad, b2r.grad
        # conv1's weight gradient given that dp1 (fp32 inside the fused kernel)
        w1r = w1.detach().clone().requires_grad_(True)
        b1r = b1.detach().clone().requires_grad_(True)
        a1f, _ = _disc_trunk_ref(x, w1r, b1r, w2.detach(), b2.detach())
        (a1f * ref_dp1).sum().backward()
        ref_dw1, ref_db1 = w1r.grad, b1r.grad
    upd = up.permute(0, 2, 3, 1).contiguous().to(DEV).to(ops.torch_dtype(dt))
    h1, w1d = p1.shape[1], p1.shape[2]
    dp1 = ops.simnn_conv2_bwd_data(upd, code2, pack, h1, w1d)
    _close(dp1.float().permute(0, 3, 1, 2), ref_dp1, 5e-5 if dt == F32 else 2e-2, "conv2 bwd data")
    dw2, db2 = ops.simnn_conv2_bwd_weight(upd, code2, p1)
    _close(dw2, ref_dw2, 1e-4 if dt == F32 else 2e-2, "conv2 bwd weight")
    _close(db2, ref_db2, 1e-4 if dt == F32 else 2e-2, "conv2 bwd bias")
    dw2b, db2b = ops.simnn_conv2_bwd_weight(upd, code2, p1)
    assert torch.equal(dw2, dw2b) and torch.equal(db2, db2b), "slab reduction must be bit-reproducible"
    # ---- conv1 weight gradient: standalone kernel from the stored dp1, and fused into the data-gradient kernel
    dw1, db1 = ops.simnn_conv1_bwd_weight(dp1, code1, xd)
    _close(dw1, ref_dw1, 2e-4 if dt == F32 else 2e-2, "conv1 bwd weight (standalone)")
    _close(db1, ref_db1, 2e-4 if dt == F32 else 2e-2, "conv1 bwd bias (standalone)")
    dw1f, db1f, dp1f = ops.simnn_conv2_bwd_fused(upd, code2, pack, code1, xd, want_dp1=True)
    _close(dw1f, ref_dw1, 2e-4 if dt == F32 else 2e-2, "conv1 bwd weight (fused)")
    _close(db1f, ref_db1, 2e-4 if dt == F32 else 2e-2, "conv1 bwd bias (fused)")
    assert torch.equal(dp1f, dp1)
    if b >= 2:   # the 2B-batch form: two input tensors behind one gradient batch
        dw1s, db1s, _ = ops.simnn_conv2_bwd_fused(upd, code2, pack, code1, xd[:1].contiguous(), xd[1:].contiguous())
        assert torch.equal(dw1s, dw1f) and torch.equal(db1s, db1f)
    # accumulate flag of the standalone kernel
    dw1a, db1a = dw1.clone(), db1.clone()
    ops.simnn_conv1_bwd_weight(dp1, code1, xd, out=(dw1a, db1a), accumulate=True)
    _close(dw1a, 2 * dw1, 1e-6, "accumulate")


@pytest.mark.parametrize("planar,c,hw,k,s,p", [(True, 2, (128, 50), 4, 2, 1), (False, 16, (64, 25), 4, 2, 1),
                                               (False, 8, (9, 7), 3, 1, 1)])
def test_im2col_col2im(planar, c, hw, k, s, p):
    g = torch.Generator().manual_seed(c + k)
    b, (h, w) = 3, hw
    x = torch.randn(b, c, h, w, generator=g)
    src = x if planar else x.permute(0, 2, 3, 1).contiguous()
    cols, oh, ow = ops.im2col(src.to(DEV), planar=planar, b=b, h=h, w=w, c=c, kh=k, kw=k, stride=s, pad=p,
                              out_dtype=F32)
    ref = F.unfold(x, k, padding=p, stride=s)                      # (b, c*k*k, L) with (c, kh, kw) order
    ref = ref.permute(0, 2, 1).reshape(b * oh * ow, c * k * k)
    _close(cols, ref, 0.0, "im2col")
    cg = torch.randn(b * oh * ow, k * k * c, generator=g)
    back = ops.col2im(cg.to(DEV), b=b, h=h, w=w, c=c, kh=k, kw=k, stride=s, pad=p, oh=oh, ow=ow, out_dtype=F32,
                      planar=planar)
    refc = cg.view(b, oh * ow, c * k * k).permute(0, 2, 1)
    refb = F.fold(refc, (h, w), k, padding=p, stride=s)
    _close(back if planar else back.permute(0, 3, 1, 2), refb, 1e-5, "col2im")


def test_permute_pc():
    x = torch.randn(3, 45, 70)
    for dt in (torch.float32, torch.bfloat16):
        got = ops.permute_pc(x.to(DEV).to(dt), 3, 45, 70)
        assert torch.equal(got.cpu(), x.to(dt).permute(0, 2, 1).contiguous())
    got = ops.permute_pc(x.to(DEV), 3, 45, 70, out_dtype=BF16)
    assert torch.equal(got.cpu(), x.permute(0, 2, 1).contiguous().to(torch.bfloat16))


def test_cpu_tensor_is_rejected():
    with pytest.raises(ops.GdmError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_conv1_forward_of_a_two_tensor_batch_in_one_launch(dt):
    """gdm_simnn_conv1_fwd_pair: the batch [x0 ; x1] (different sizes) in ONE launch equals the two launches bit for bit."""
    dtc = ops.BF16 if dt == "bf16" else ops.F32
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(3, 30, 44, generator=g) * 18 - 35).to(DEV)
    x1 = (torch.randn(5, 30, 44, generator=g) * 18 - 35).to(DEV)
    w = (torch.randn(16, 1, 2, 2, generator=g) * 0.2).to(DEV)
    bias = (torch.randn(16, generator=g) * 0.5 + 2.0).to(DEV)
    pa, ca = ops.simnn_conv1_fwd(x0, w, bias, dtc)
    pb, cb = ops.simnn_conv1_fwd(x1, w, bias, dtc)
    p, c = ops.simnn_conv1_fwd(x0, w, bias, dtc, x1=x1)
    assert p.shape[0] == 8
    assert torch.equal(p[:3], pa) and torch.equal(p[3:], pb)
    assert torch.equal(c[:3], ca) and torch.equal(c[3:], cb)
