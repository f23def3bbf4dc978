"""Network-level forward/backward composites over the C-ABI kernels, and their autograd.Function shells.

Each composite is a fixed sequence of ``ops.*`` calls (all enqueued on the current HIP stream, no host sync), returns
what backward needs as a plain tuple, and is used both by the nn.Module classes (through the autograd Functions
below) and by the fused training steps in ``train.py`` (directly, without autograd).

compute dtype: ``F32`` = exact-fp32 parity mode, ``BF16`` = bf16 activations/MFMA with fp32 accumulation and fp32
parameters.  ``set_compute_dtype`` / ``compute_dtype`` select the process-wide default used by the modules.
"""
import contextlib

import torch

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F32

_DEFAULT = {"dtype": F32}
_NAMES = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, F32: F32, BF16: BF16}


def set_compute_dtype(name):
    _DEFAULT["dtype"] = _NAMES[name]


def get_compute_dtype():
    return _DEFAULT["dtype"]


@contextlib.contextmanager
def compute_dtype(name):
    old = _DEFAULT["dtype"]
    _DEFAULT["dtype"] = _NAMES[name]
    try:
        yield
    finally:
        _DEFAULT["dtype"] = old


def _f32c(t):
    return t.detach().contiguous() if t.dtype == torch.float32 else t.detach().float().contiguous()


# ======================================================================================================================
# Model 1 discriminator (GAN_DES/SIMNN.py:115-142)
# ======================================================================================================================
def simnn_disc_prepare(w2, wf1, dt, out=None):
    """Per-weight-version operands: conv2's packed MFMA images and fc1's weight permuted to the channels-last flatten
    order of the feature map (and cast to the activation dtype).  (128, 32*P) -> (128, P*32).
    ``out`` = a previous result to refresh in place (persistent buffers: required under graph replay)."""
    n, k = wf1.shape
    if out is not None:
        pack, wf1p = out
        ops.simnn_conv2_pack(w2, dt, out=pack)
        ops.permute_pc(wf1, n, 32, k // 32, out=wf1p)
        return pack, wf1p
    pack = ops.simnn_conv2_pack(w2, dt)
    wf1p = ops.permute_pc(wf1, n, 32, k // 32, out_dtype=dt).view(n, k)
    return pack, wf1p


def simnn_disc_features(x, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out=None):
    """x (B,H,W) fp32 -> h1 (B,128) fp32 = relu(fc1(flatten(trunk(x)))); returns (h1, saved).

    pack, wf1p come from simnn_disc_prepare.  trunk_out = (p1, code1) buffers already filled by conv1 (2B batches)."""
    if trunk_out is None:
        x = _f32c(x)
        p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, dt)
    else:
        p1, code1 = trunk_out
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
    b = p1.shape[0]
    flat = p2.view(b, -1)
    if flat.shape[1] != wf1p.shape[1]:
        raise ValueError(f"Discriminator.fc1 expects {wf1p.shape[1]} features but this input gives {flat.shape[1]} "
                         "(construct Discriminator(input_hw=...) for this geometry)")
    h1 = ops.gemm(flat, wf1p.t(), bias_n=bf1, act=ACT_RELU, compute=dt)
    return h1, (x, p1, code1, flat, code2, h1)


def simnn_disc_forward(x, w1, b1, pack, b2, wf1p, bf1, wf2, bf2, dt, trunk_out=None):
    """... -> p (B,1) fp32 = sigmoid(fc2(h1)); returns (p, saved)."""
    h1, saved = simnn_disc_features(x, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out)
    p = ops.gemm(h1, wf2.t(), bias_n=bf2, act=ACT_SIGMOID, compute=dt)
    return p, saved + (p,)


def simnn_disc_backward_from_dh1(saved, dh1, pack, wf1p, dt, out=None, x_pair=None, w1_for_dx=None):
    """dh1 (B,128) fp32 = gradient w.r.t. fc1's pre-activation (already through the ReLU).

    Returns (dw1, db1, dw2, db2, dwf1, dx); ``out`` = the 8 gradient tensors in parameter order (only the first five
    are written).  x_pair = (x0, x1) when the batch is the concatenation of two input tensors.  dx (B,H,W) = gradient
    w.r.t. the input, computed only when ``w1_for_dx`` (conv1.weight) is given (else None)."""
    x, p1, code1, flat, code2 = saved[:5]
    b = p1.shape[0]
    o = out if out is not None else [None] * 8
    n, k = wf1p.shape
    dwf1p = ops.gemm(dh1.t(), flat, compute=dt)                               # (128, P*32) channels-last order
    dwf1 = ops.permute_pc(dwf1p, n, k // 32, 32, out=o[4]).view(n, k)         # back to the parameter's (c, pix) order
    dflat = ops.gemm(dh1, wf1p, compute=dt, out_dtype=dt)                     # (B,K) = dp2, channels-last
    h1s, w1s = p1.shape[1], p1.shape[2]
    dp2 = dflat.view(b, h1s // 2, w1s // 2, 32)
    dw2, db2 = ops.simnn_conv2_bwd_weight(dp2, code2, p1, out=None if out is None else (o[2], o[3]))
    x0, x1 = x_pair if x_pair is not None else (x, None)
    dw1, db1, dp1 = ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x0, x1,
                                              out=None if out is None else (o[0], o[1]), want_dp1=w1_for_dx is not None)
    dx = None
    if w1_for_dx is not None:
        dx = ops.simnn_conv1_bwd_data(dp1, code1, w1_for_dx.contiguous(), x0.shape[1], x0.shape[2])
    return dw1, db1, dw2, db2, dwf1, dx


def simnn_disc_backward(saved, dz, pack, wf1p, wf2, dt, out=None, x_pair=None, w1_for_dx=None):
    """dz (B,1) fp32 = d loss / d (pre-sigmoid logit).  Returns grads in parameter order
    (w1,b1,w2,b2,wf1,bf1,wf2,bf2) + the input gradient (None unless ``w1_for_dx`` = conv1.weight is given);
    ``out`` = 8 preallocated gradient tensors to fill instead."""
    h1 = saved[5]
    b = h1.shape[0]
    o = out if out is not None else [None] * 8
    dz = dz.reshape(b, 1).contiguous()
    dwf2 = ops.gemm(dz.t(), h1, compute=dt, out=o[6])                         # (1,128)
    dbf2 = ops.colsum(dz, out=o[7])
    dh1 = ops.act_bwd(ops.gemm(dz, wf2, compute=dt), h1, act=ACT_RELU)        # (B,128)
    dbf1 = ops.colsum(dh1, out=o[5])
    dw1, db1, dw2, db2, dwf1, dx = simnn_disc_backward_from_dh1(saved, dh1, pack, wf1p, dt, out, x_pair, w1_for_dx)
    return dw1, db1, dw2, db2, dwf1, dbf1, dwf2, dbf2, dx


def _anomaly_guard(cls):
    """torch.autograd.set_detect_anomaly(True) -- the reference's entry point sets it (network_tests.py:211) -- makes
    torch raise when a backward function returns NaN.  The kernels behind these Functions are compiled without NaN
    semantics, so under anomaly mode the Functions look at the bits themselves (ops.assert_finite: one small counting
    launch per tensor + a host read -- anomaly mode is a debugging mode in torch too): inputs and outputs of forward,
    and every gradient backward returns.  Raises ops.NonFiniteError (a RuntimeError, like torch's)."""
    fwd, bwd = cls.forward, cls.backward

    def _tensors(xs):
        out = []
        for x in xs:
            if isinstance(x, torch.Tensor) and x.is_floating_point() and x.is_cuda:
                out.append(x.detach())
            elif isinstance(x, (tuple, list)):
                out.extend(_tensors(x))
        return out

    def forward(ctx, *args):
        if not torch.is_anomaly_enabled():
            return fwd(ctx, *args)
        ops.assert_finite(_tensors(args), f"Function '{cls.__name__}' received")
        out = fwd(ctx, *args)
        ops.assert_finite(_tensors(out if isinstance(out, (tuple, list)) else (out,)), f"Function '{cls.__name__}' returned")
        return out

    def backward(ctx, *grads):
        out = bwd(ctx, *grads)
        if torch.is_anomaly_enabled():
            ops.assert_finite(_tensors(grads), f"Function '{cls.__name__}Backward' received")
            ops.assert_finite(_tensors(out if isinstance(out, (tuple, list)) else (out,)),
                              f"Function '{cls.__name__}Backward' returned nan values:")
        return out

    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls



@_anomaly_guard
class SimnnDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wf1, bf1, wf2, bf2, dt):
        pack, wf1p = simnn_disc_prepare(w2.detach(), wf1.detach(), dt)
        p, saved = simnn_disc_forward(x, w1.detach(), b1.detach(), pack, b2.detach(), wf1p, bf1.detach(),
                                      wf2.detach(), bf2.detach(), dt)
        ctx.saved = saved
        ctx.weights = (pack, wf1p, wf2.detach(), w1.detach())
        ctx.dt = dt
        ctx.x_needs_grad = x.requires_grad
        ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
        return p

    @staticmethod
    def backward(ctx, dp):
        p = ctx.saved[-1]
        dz = ops.act_bwd(dp.contiguous().float(), p, act=ACT_SIGMOID)
        pack, wf1p, wf2, w1 = ctx.weights
        # the reference's own loops never ask for the input gradient (data / detached bridge outputs, SIMNN.py:283,
        # 299-306), but the module is an ordinary autograd citizen: dx on request
        *grads, dx = simnn_disc_backward(ctx.saved, dz, pack, wf1p, wf2, ctx.dt,
                                         w1_for_dx=w1 if ctx.x_needs_grad else None)
        if dx is not None:
            dx = dx.view(ctx.x_shape).to(ctx.x_dtype)
        return (dx, *grads, None)


# ======================================================================================================================
# ConvTranspose2d / Conv2d as GEMM + patch lowering (channels-last activations)
# ======================================================================================================================
def _tap_major_weight(w, cache, slot):
    """(Cin, Cout, KH, KW) -> (Cin, KH*KW*Cout) copy: the GEMM then writes its columns in (kh, kw, cout) order, which
    col2im reads coalesced (torch's order costs a 64-byte stride per lane there).

    ``cache`` is a dict owned by whoever owns the weight (a trainer, a module): entry ``slot`` holds
    (storage address, version, copy) and is refreshed in place when the weight changed, so the copy's storage is
    stable under graph replay.  Without a cache the copy is rebuilt on every call."""
    cin, cout, kh, kw = w.shape
    wc = w.detach().contiguous()

    def build(out=None):     # (Cin, Cout, KH*KW) -> (Cin, KH*KW, Cout): gdm_permute_pc per input channel
        return ops.permute_pc(wc, cin, cout, kh * kw, out=out).view(cin, kh * kw * cout)
    if cache is None:
        return build()
    hit = cache.get(slot)
    if hit is not None and hit[0] == w.data_ptr() and hit[1] == w._version and hit[2].numel() == w.numel():
        return hit[2]
    if hit is not None and hit[2].numel() == w.numel():
        wp = build(out=hit[2].view(cin, kh * kw, cout))
    else:
        if torch.cuda.is_current_stream_capturing():
            raise ops.GdmError("a ConvTranspose2d weight was first seen inside a graph capture: run one eager step first")
        wp = build()
    cache[slot] = (w.data_ptr(), w._version, wp)
    return wp


def convT_forward(x2d, w, b, ih, iw, stride, pad, dt, cache=None, slot=0, act=ACT_NONE):
    """x2d (B*IH*IW, Cin) -> y (B*OH*OW, Cout) fp32 channels-last; w is torch's (Cin, Cout, KH, KW).
    cache / slot: see _tap_major_weight.  act: activation fused into the scatter pass."""
    cin, cout, kh, kw = w.shape
    oh = (ih - 1) * stride - 2 * pad + kh
    ow = (iw - 1) * stride - 2 * pad + kw
    tap_major = cout >= 16               # with few output channels torch's order is already the contiguous one
    w2d = _tap_major_weight(w, cache, slot) if tap_major else w.view(cin, cout * kh * kw)
    cols = ops.gemm(x2d, w2d, compute=dt, out_dtype=F32)
    if tap_major and ih == 1 and iw == 1 and stride == 1 and pad == 0:
        # a 1x1 input (the generator's first layer, SIMNN.py:70): output pixel (oh, ow) IS tap (kh, kw) = (oh, ow) of the
        # only input pixel, so the tap-major GEMM result already is the channels-last output -- no scatter pass
        assert act == ACT_NONE
        return cols.view(b * oh * ow, cout), oh, ow
    y = ops.col2im(cols, b=b, h=oh, w=ow, c=cout, kh=kh, kw=kw, stride=stride, pad=pad, oh=ih, ow=iw, out_dtype=F32,
                   tap_major=tap_major, act=act)
    return y.view(b * oh * ow, cout), oh, ow


def convT_backward(dy2d, x2d, w, b, ih, iw, oh, ow, stride, pad, dt, need_dx=True):
    """dy2d (B*OH*OW, Cout) fp32 -> (dx2d (B*IH*IW, Cin) fp32 or None, dw like w)."""
    cin, cout, kh, kw = w.shape
    dcols, oh2, ow2 = ops.im2col(dy2d, planar=False, b=b, h=oh, w=ow, c=cout, kh=kh, kw=kw, stride=stride, pad=pad,
                                 out_dtype=F32)
    # im2col's output grid for a transposed conv is the transposed conv's INPUT grid
    if (oh2, ow2) != (ih, iw):
        raise AssertionError(f"transposed-conv geometry mismatch {(oh2, ow2)} vs {(ih, iw)}")
    dw = ops.gemm(x2d.t(), dcols, compute=dt).view(cin, cout, kh, kw)
    dx = ops.gemm(dcols, w.view(cin, cout * kh * kw).t(), compute=dt) if need_dx else None
    return dx, dw


# ======================================================================================================================
# Model 1 generator (GAN_DES/SIMNN.py:62-112)
# ======================================================================================================================
_G_GEOM = ((1, 0), (2, 1), (2, 1), (1, 0))   # (stride, padding) of conv1..conv4


def _gen_pack(ws, cache):
    """bf16 GEMM images of conv2 / conv3 for the fused generator kernels, cached per weight version like
    _tap_major_weight (refreshed in place when a weight changed; the generators are frozen during training)."""
    key = "gen_pack"
    ver = tuple(v for w in ws[:3] for v in (w.data_ptr(), w._version))
    hit = cache.get(key) if cache is not None else None
    if hit is not None and hit[0] == ver:
        return hit[1]
    if hit is None and cache is not None and torch.cuda.is_current_stream_capturing():
        raise ops.GdmError("the generator's weights were first seen inside a graph capture: run one eager step first")
    pack = ops.simnn_gen_pack(ws[0].contiguous(), ws[1].contiguous(), ws[2].contiguous(),
                              out=None if hit is None else hit[1])
    if cache is not None:
        cache[key] = (ver, pack)
    return pack


def simnn_gen_forward_fused(noise, ws, bns, dt, cache=None):
    """The training loop's generator forward (train-mode BatchNorm, reference geometry, forward only): layer 1 as a
    GEMM with its batch statistics in the same launch, layers 2..4 as three fused kernels that normalise their input
    while they stage it (csrc/simnn_gen.hip).  6 launches instead of 20; nothing is saved for a backward."""
    b = noise.shape[0]
    x = _f32c(noise).view(b, -1)
    pack = _gen_pack(ws, cache)
    g1, be1, rm1, rv1, nbt1 = bns[0]
    if b <= 256:          # layer 1 and its exact batch statistics in one launch (a workgroup owns the whole batch)
        y1, mean, invstd = ops.simnn_gen_first(x, pack, rm1, rv1, nbt1)
    else:
        y1, _, _ = convT_forward(x, ws[0], b, 1, 1, 1, 0, dt, cache=cache, slot=0)          # (B*16, 128): no scatter pass
        mean, invstd = ops.bn_stats(y1, rm1, rv1, nbt1)
    y2, part, chunks = ops.simnn_gen_convt_bn(2, y1, mean, invstd, g1, be1, b, pack)
    g2, be2, rm2, rv2, nbt2 = bns[1]
    mean, invstd = ops.bn_finalize(part, chunks, b * 64, 64, rm2, rv2, nbt2)
    y3, part, chunks = ops.simnn_gen_convt_bn(3, y2, mean, invstd, g2, be2, b, pack)
    g3, be3, rm3, rv3, nbt3 = bns[2]
    mean, invstd = ops.bn_finalize(part, chunks, b * 256, 32, rm3, rv3, nbt3)
    return ops.simnn_gen_last(y3, mean, invstd, g3, be3, ws[3].contiguous(), b)


def _gen_fused_ok(noise, ws, training, dt, need_backward):
    return (training and not need_backward and dt == BF16 and noise.shape[0] > 1 and tuple(ws[0].shape[1:]) == (128, 4, 4)
            and ws[0].shape[0] <= 128
            and tuple(ws[1].shape) == (128, 64, 4, 4) and tuple(ws[2].shape) == (64, 32, 4, 4)
            and tuple(ws[3].shape) == (32, 1, 5, 5))


def simnn_gen_forward(noise, ws, bns, training, dt, cache=None, need_backward=True):
    """noise (B,100,1,1); ws = 4 ConvTranspose2d weights; bns = 3 x (gamma, beta, rmean, rvar, nbt).

    Returns (out (B,1,20,20) fp32, saved).  Activations are channels-last 2-D matrices (B*H*W, C).
    need_backward=False (the trainers: no gradient ever reaches a generator) takes the fused
    forward-only kernels when the geometry is the reference's; ``saved`` is then None.
    """
    if _gen_fused_ok(noise, ws, training, dt, need_backward):
        return simnn_gen_forward_fused(noise, ws, bns, dt, cache), None
    b = noise.shape[0]
    x = _f32c(noise).view(b, -1)
    ih = iw = 1
    saved = []
    for li in range(4):
        stride, pad = _G_GEOM[li]
        w = ws[li]
        # the last layer's sigmoid (SIMNN.py:110) rides the scatter pass; backward only needs the activated output
        y, oh, ow = convT_forward(x, w, b, ih, iw, stride, pad, dt, cache=cache, slot=li,
                                  act=ACT_SIGMOID if li == 3 else ACT_NONE)
        if li < 3:
            gamma, beta, rm, rv, nbt = bns[li]
            out, mean, invstd = ops.bn_act_fwd(y, gamma, beta, rm, rv, nbt, act=ACT_RELU, out_dtype=F32,
                                               training=training)
            saved.append((x, y, out, mean, invstd, ih, iw, oh, ow))
            x = out
        else:
            out = y
            saved.append((x, None, out, None, None, ih, iw, oh, ow))
        ih, iw = oh, ow
    c_out = ws[3].shape[1]
    # channels-last (B,20,20,C) -> (B,C,20,20); for the reference's C == 1 this is a free view
    img = out.view(b, ih, iw, c_out)
    img = img.view(b, 1, ih, iw) if c_out == 1 else ops.permute_pc(out, b, ih * iw, c_out).view(b, c_out, ih, iw)
    return img, saved


def simnn_gen_backward(saved, dimg, ws, bns, dt, need_dnoise=True):
    """dimg (B,C,20,20) -> (dnoise or None, [dw1..dw4], [(dgamma, dbeta) x3])."""
    b = dimg.shape[0]
    c_out = ws[3].shape[1]
    dimg = _f32c(dimg)
    d = dimg.view(-1, 1) if c_out == 1 else ops.permute_pc(dimg, b, c_out, dimg.shape[2] * dimg.shape[3]).view(-1, c_out)
    dws, dbn = [None] * 4, [None] * 3
    for li in (3, 2, 1, 0):
        x, y, out, mean, invstd, ih, iw, oh, ow = saved[li]
        stride, pad = _G_GEOM[li]
        if li == 3:
            dy = ops.act_bwd(d.contiguous(), out, act=ACT_SIGMOID)
        else:
            gamma = bns[li][0]
            dy, dgamma, dbeta = ops.bn_act_bwd(d.contiguous(), out, y, gamma, mean, invstd, act=ACT_RELU)
            dbn[li] = (dgamma, dbeta)
        d, dws[li] = convT_backward(dy, x, ws[li], b, ih, iw, oh, ow, stride, pad, dt,
                                    need_dx=(li > 0 or need_dnoise))
    dnoise = d.view(b, -1, 1, 1) if d is not None else None
    return dnoise, dws, dbn


@_anomaly_guard
class SimnnGenFn(torch.autograd.Function):
    """args: noise, w1..w4, (gamma,beta) x3, then non-differentiable: buffers tuple, training flag, dtype."""

    @staticmethod
    def forward(ctx, noise, w1, w2, w3, w4, g1, be1, g2, be2, g3, be3, buffers, training, dt):
        ws = [w.detach() for w in (w1, w2, w3, w4)]
        bns = [(g.detach(), be.detach(), *buf) for (g, be), buf in zip(((g1, be1), (g2, be2), (g3, be3)), buffers)]
        img, saved = simnn_gen_forward(noise, ws, bns, training, dt)
        if not training:
            ctx.eval_mode = True
        ctx.saved, ctx.ws, ctx.bns, ctx.dt = saved, ws, bns, dt
        ctx.need_dnoise = noise.requires_grad
        ctx.training = training
        return img

    @staticmethod
    def backward(ctx, dimg):
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not on the reference's path")
        dnoise, dws, dbn = simnn_gen_backward(ctx.saved, dimg, ctx.ws, ctx.bns, ctx.dt, ctx.need_dnoise)
        return (dnoise, *dws, dbn[0][0], dbn[0][1], dbn[1][0], dbn[1][1], dbn[2][0], dbn[2][1], None, None, None)


# ======================================================================================================================
# Model 2 generators: 4 x [Linear -> BatchNorm1d -> Sigmoid] (MMGAN_MIDI_DES/network_tests.py:67-80, 102-115)
# ======================================================================================================================
def mlp_bn_sigmoid_forward(x, layers, training, dt, need_backward=True, groups=1, stat_repeats=1):
    """layers: list of (W (out,in), b, gamma, beta, rmean, rvar, nbt).  Returns (out fp32, saved).

    bf16 mode with <= 256 rows: one fused Linear+BN+Sigmoid launch per block (batch statistics never leave the
    workgroup); otherwise GEMM + the three-launch batch norm (any batch size, exact-fp32 mode).
    groups / stat_repeats (fused path only, forward only): x stacks ``groups`` batches that are normalised separately,
    running statistics updated in order, each update applied ``stat_repeats`` times (ops.linear_bn_act_fwd)."""
    saved = []
    x = _f32c(x)
    rows = x.shape[0] // groups
    fused = dt == BF16 and rows <= ops.linear_bn_act_max_rows() and (not training or rows > 1)
    if (groups > 1 or stat_repeats > 1) and (not fused or need_backward):
        raise ops.GdmError("stacked / repeated generator forwards exist on the fused forward-only path only")
    for (w, bias, gamma, beta, rm, rv, nbt) in layers:
        if fused:
            out, y, mean, invstd = ops.linear_bn_act_fwd(x, w, bias, gamma, beta, rm, rv, nbt, act=ACT_SIGMOID,
                                                         training=training, save_y=need_backward, groups=groups,
                                                         stat_repeats=stat_repeats)
        else:
            y = ops.gemm(x, w.t(), bias_n=bias, compute=dt)
            out, mean, invstd = ops.bn_act_fwd(y, gamma, beta, rm, rv, nbt, act=ACT_SIGMOID, out_dtype=F32,
                                               training=training)
        saved.append((x, y, out, mean, invstd))
        x = out
    return x, saved


def mlp_bn_sigmoid_backward(saved, dout, layers, dt, need_dx=True):
    """Returns (dx or None, [(dW, db, dgamma, dbeta)] per layer)."""
    grads = [None] * len(layers)
    d = _f32c(dout)
    for li in range(len(layers) - 1, -1, -1):
        x, y, out, mean, invstd = saved[li]
        w, gamma = layers[li][0], layers[li][2]
        dy, dgamma, dbeta = ops.bn_act_bwd(d, out, y, gamma, mean, invstd, act=ACT_SIGMOID)
        dw = ops.gemm(dy.t(), x, compute=dt)
        db = ops.colsum(dy)
        grads[li] = (dw, db, dgamma, dbeta)
        d = ops.gemm(dy, w, compute=dt) if (li > 0 or need_dx) else None
    return d, grads


def mlp_bn_sigmoid_forward_global(x, layers, group=None):
    """Train-mode generator MLP (network_tests.py:75-80, 110-115) whose BatchNorm1d statistics are taken over the GLOBAL
    batch of a data-parallel job (SURVEY.md 8e "exact mode"): every rank computes y = x W^T + b on its shard (exact-fp32
    MFMA), its per-row-chunk Welford triples, all ranks exchange those (a few KB per layer) and merge them in rank order
    (gdm_bn_finalize: identical on every rank, running statistics updated with the global batch), then normalise +
    sigmoid.  With equal shards the result equals one process on the concatenated batch up to summation order."""
    from . import dp
    world = dp.world_size(group)
    h = _f32c(x)
    for (w, b, gamma, beta, rm, rv, nbt) in layers:
        y = ops.gemm(h, w.t(), bias_n=b, compute=F32)
        part = dp.all_gather_cat(ops.bn_partials(y), group)
        mean, invstd = ops.bn_finalize(part, part.shape[0], y.shape[0] * world, y.shape[1], rm, rv, nbt)
        h = ops.bn_apply(y, gamma, beta, mean, invstd, act=ACT_SIGMOID)
    return h


@_anomaly_guard
class MlpBnSigmoidFn(torch.autograd.Function):
    """args: x, then per layer (W, b, gamma, beta) x L, then buffers tuple ((rm, rv, nbt) x L), training, dtype."""

    @staticmethod
    def forward(ctx, x, *rest):
        buffers, training, dt = rest[-3], rest[-2], rest[-1]
        params = rest[:-3]
        n_layers = len(params) // 4
        layers = [tuple(p.detach() for p in params[4 * i:4 * i + 4]) + tuple(buffers[i]) for i in range(n_layers)]
        out, saved = mlp_bn_sigmoid_forward(x, layers, training, dt)
        ctx.saved, ctx.layers, ctx.dt, ctx.training = saved, layers, dt, training
        ctx.need_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not on the reference's path")
        dx, grads = mlp_bn_sigmoid_backward(ctx.saved, dout, ctx.layers, ctx.dt, ctx.need_dx)
        flat = [g for layer in grads for g in layer]
        return (dx, *flat, None, None, None)


# ======================================================================================================================
# Model 2 CNN discriminator (MMGAN_MIDI_DES/network_tests.py:147-160)
# ======================================================================================================================
def dcnn_forward(image, w1, b1, w2, b2, wf, bf, dt):
    """image (B,2,H,T) fp32 (any strides) -> logits (B,1) fp32; returns (logits, saved)."""
    b, c, h, t = image.shape
    img = _f32c(image)
    co1, co2 = w1.shape[0], w2.shape[0]
    cols1, oh1, ow1 = ops.im2col(img, planar=True, b=b, h=h, w=t, c=c, kh=4, kw=4, stride=2, pad=1, out_dtype=dt)
    a1 = ops.gemm(cols1, w1.view(co1, -1).t(), bias_n=b1, act=ACT_LEAKY, slope=0.2, compute=dt, out_dtype=dt)
    cols2, oh2, ow2 = ops.im2col(a1, planar=False, b=b, h=oh1, w=ow1, c=co1, kh=4, kw=4, stride=2, pad=1,
                                 out_dtype=dt)
    a2 = ops.gemm(cols2, w2.view(co2, -1).t(), bias_n=b2, act=ACT_LEAKY, slope=0.2, compute=dt, out_dtype=dt)
    flat = ops.permute_pc(a2, b, oh2 * ow2, co2).view(b, -1)          # channel-major flatten (x.view(len(x), -1))
    if flat.shape[1] != wf.shape[1]:
        raise ValueError(f"DiscriminatorCNN.fc expects {wf.shape[1]} features, the {(c, h, t)} roll gives "
                         f"{flat.shape[1]}")
    logits = ops.gemm(flat, wf.t(), bias_n=bf, compute=dt)
    return logits, (cols1, a1, cols2, a2, flat, (b, oh1, ow1, oh2, ow2, co1, co2))


def dcnn_backward(saved, dlogits, w2, wf, dt, w1_for_dx=None, in_hw=None):
    """dlogits (B,1) fp32.  Returns (dw1, db1, dw2, db2, dwf, dbf, dx): dx (B,C,H,T) fp32 = gradient w.r.t. the
    piano-roll input when ``w1_for_dx`` (conv1.weight) and ``in_hw`` = (C, H, T) are given, else None."""
    cols1, a1, cols2, a2, flat, (b, oh1, ow1, oh2, ow2, co1, co2) = saved
    dl = _f32c(dlogits).view(b, 1)
    dwf = ops.gemm(dl.t(), flat, compute=dt)
    dbf = ops.colsum(dl)
    dflat = ops.gemm(dl, wf, compute=dt, out_dtype=dt)                              # (B, co2*P2) channel-major
    da2 = ops.permute_pc(dflat, b, co2, oh2 * ow2).view(b * oh2 * ow2, co2)        # back to channels-last
    dy2 = ops.act_bwd(da2, a2, act=ACT_LEAKY, slope=0.2)
    dw2 = ops.gemm(dy2.t(), cols2, compute=dt).view(w2.shape)
    db2 = ops.colsum(dy2)
    dcols2 = ops.gemm(dy2, w2.view(co2, -1), compute=dt, out_dtype=F32)
    da1 = ops.col2im(dcols2, b=b, h=oh1, w=ow1, c=co1, kh=4, kw=4, stride=2, pad=1, oh=oh2, ow=ow2,
                     out_dtype=dt).view(b * oh1 * ow1, co1)
    dy1 = ops.act_bwd(da1, a1, act=ACT_LEAKY, slope=0.2)
    dw1 = ops.gemm(dy1.t(), cols1, compute=dt).view(co1, -1, 4, 4)
    db1 = ops.colsum(dy1)
    dx = None
    if w1_for_dx is not None:
        c, h, t = in_hw
        dcols1 = ops.gemm(dy1, w1_for_dx.reshape(co1, -1), compute=dt, out_dtype=F32)      # (B*oh1*ow1, C*16)
        dx = ops.col2im(dcols1, b=b, h=h, w=t, c=c, kh=4, kw=4, stride=2, pad=1, oh=oh1, ow=ow1, out_dtype=F32,
                        planar=True)
    return dw1, db1, dw2, db2, dwf, dbf, dx


@_anomaly_guard
class DcnnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, w1, b1, w2, b2, wf, bf, dt):
        logits, saved = dcnn_forward(image, w1.detach(), b1.detach(), w2.detach(), b2.detach(), wf.detach(),
                                     bf.detach(), dt)
        ctx.saved, ctx.dt = saved, dt
        ctx.weights = (w2.detach(), wf.detach(), w1.detach())
        ctx.x_needs_grad = image.requires_grad
        ctx.x_shape, ctx.x_dtype = image.shape, image.dtype
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        w2, wf, w1 = ctx.weights
        # the reference's loop never asks for the input gradient (bridge outputs carry no graph, network_tests.py:189-193);
        # computed on request so that the module is an ordinary autograd citizen
        *grads, dx = dcnn_backward(ctx.saved, dlogits, w2, wf, ctx.dt, w1_for_dx=w1 if ctx.x_needs_grad else None,
                                   in_hw=tuple(ctx.x_shape[1:]))
        dw1 = grads[0].view(ctx.saved[-1][5], -1, 4, 4)
        if dx is not None:
            dx = dx.view(ctx.x_shape).to(ctx.x_dtype)
        return (dx, dw1, *grads[1:], None)


# ======================================================================================================================
# Model 2 MLP discriminator: 3 x [Linear -> LeakyReLU(0.2)] (network_tests.py:126-144); API surface
# ======================================================================================================================
def mlp_leaky_forward(x, layers, dt):
    saved = []
    x = _f32c(x)
    for (w, bias) in layers:
        out = ops.gemm(x, w.t(), bias_n=bias, act=ACT_LEAKY, slope=0.2, compute=dt)
        saved.append((x, out))
        x = out
    return x, saved


def mlp_leaky_backward(saved, dout, layers, dt, need_dx=False):
    grads = [None] * len(layers)
    d = _f32c(dout)
    for li in range(len(layers) - 1, -1, -1):
        x, out = saved[li]
        dy = ops.act_bwd(d, out, act=ACT_LEAKY, slope=0.2)
        grads[li] = (ops.gemm(dy.t(), x, compute=dt), ops.colsum(dy))
        d = ops.gemm(dy, layers[li][0], compute=dt) if (li > 0 or need_dx) else None
    return d, grads


@_anomaly_guard
class MlpLeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *rest):
        dt = rest[-1]
        params = rest[:-1]
        layers = [(params[2 * i].detach(), params[2 * i + 1].detach()) for i in range(len(params) // 2)]
        out, saved = mlp_leaky_forward(x, layers, dt)
        ctx.saved, ctx.layers, ctx.dt, ctx.need_dx = saved, layers, dt, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, grads = mlp_leaky_backward(ctx.saved, dout, ctx.layers, ctx.dt, ctx.need_dx)
        return (dx, *[g for layer in grads for g in layer], None)


# ======================================================================================================================
# SimNN (GAN_DES/SIMNN.py:145-170): Conv(1->32,k3,p1) ReLU pool, Conv(32->64,k3,p1) ReLU pool, flatten (channel-major),
# Linear(.,512) ReLU, Linear(512, n*n+4n).  Not on the training path: generic kernels.
# ======================================================================================================================
def simnn_net_forward(x, w1, b1, w2, b2, wf1, bf1, wf2, bf2, dt):
    """x (B,1,H,W) -> (out (B, n*n+4n) fp32, saved)."""
    b, cin, h, w = x.shape
    x = _f32c(x)
    c1, c2 = w1.shape[0], w2.shape[0]
    cols1, _, _ = ops.im2col(x, planar=True, b=b, h=h, w=w, c=cin, kh=3, kw=3, stride=1, pad=1, out_dtype=dt)
    a1 = ops.gemm(cols1, w1.reshape(c1, -1).t(), bias_n=b1, act=ACT_RELU, compute=dt, out_dtype=dt)
    q1, i1 = ops.maxpool2_fwd(a1, b, h, w, c1)
    h1, w1s = h // 2, w // 2
    cols2, _, _ = ops.im2col(q1, planar=False, b=b, h=h1, w=w1s, c=c1, kh=3, kw=3, stride=1, pad=1, out_dtype=dt)
    a2 = ops.gemm(cols2, w2.reshape(c2, -1).t(), bias_n=b2, act=ACT_RELU, compute=dt, out_dtype=dt)
    q2, i2 = ops.maxpool2_fwd(a2, b, h1, w1s, c2)
    h2, w2s = h1 // 2, w1s // 2
    flat = ops.permute_pc(q2, b, h2 * w2s, c2).view(b, -1)             # x.view(x.size(0), -1) of an NCHW tensor
    if flat.shape[1] != wf1.shape[1]:
        raise ValueError(f"SimNN.fc1 has {wf1.shape[1]} inputs, this input gives {flat.shape[1]}")
    hid = ops.gemm(flat, wf1.t(), bias_n=bf1, act=ACT_RELU, compute=dt)
    out = ops.gemm(hid, wf2.t(), bias_n=bf2, compute=dt)
    return out, (cols1, a1, i1, cols2, a2, i2, flat, hid, (b, cin, h, w, c1, c2))


def simnn_net_backward(saved, dout, w1, w2, wf1, wf2, dt, need_dx=False):
    """Returns (dw1, db1, dw2, db2, dwf1, dbf1, dwf2, dbf2, dx or None)."""
    cols1, a1, i1, cols2, a2, i2, flat, hid, (b, cin, h, w, c1, c2) = saved
    h1, w1s, h2, w2s = h // 2, w // 2, h // 4, w // 4
    d = _f32c(dout)
    dwf2, dbf2 = ops.gemm(d.t(), hid, compute=dt), ops.colsum(d)
    dhid = ops.act_bwd(ops.gemm(d, wf2, compute=dt), hid, act=ACT_RELU)
    dwf1, dbf1 = ops.gemm(dhid.t(), flat, compute=dt), ops.colsum(dhid)
    dflat = ops.gemm(dhid, wf1, compute=dt, out_dtype=dt)                                 # (B, c2*P) channel-major
    dq2 = ops.permute_pc(dflat, b, c2, h2 * w2s).view(b * h2 * w2s, c2)
    da2 = ops.act_bwd(ops.maxpool2_bwd(dq2, i2, b, h1, w1s, c2), a2, act=ACT_RELU)
    dw2, db2 = ops.gemm(da2.t(), cols2, compute=dt).view(w2.shape), ops.colsum(da2)
    dcols2 = ops.gemm(da2, w2.reshape(c2, -1), compute=dt, out_dtype=F32)
    dq1 = ops.col2im(dcols2, b=b, h=h1, w=w1s, c=c1, kh=3, kw=3, stride=1, pad=1, oh=h1, ow=w1s,
                     out_dtype=dt).view(b * h1 * w1s, c1)
    da1 = ops.act_bwd(ops.maxpool2_bwd(dq1, i1, b, h, w, c1), a1, act=ACT_RELU)
    dw1, db1 = ops.gemm(da1.t(), cols1, compute=dt).view(w1.shape), ops.colsum(da1)
    dx = None
    if need_dx:
        dcols1 = ops.gemm(da1, w1.reshape(c1, -1), compute=dt, out_dtype=F32)
        dx = ops.col2im(dcols1, b=b, h=h, w=w, c=cin, kh=3, kw=3, stride=1, pad=1, oh=h, ow=w, out_dtype=F32,
                        planar=True)
    return dw1, db1, dw2, db2, dwf1, dbf1, dwf2, dbf2, dx


@_anomaly_guard
class SimnnNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wf1, bf1, wf2, bf2, dt):
        ps = [p.detach() for p in (w1, b1, w2, b2, wf1, bf1, wf2, bf2)]
        out, saved = simnn_net_forward(x, *ps, dt)
        ctx.saved, ctx.ps, ctx.dt, ctx.need_dx = saved, ps, dt, x.requires_grad
        ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        w1, _, w2, _, wf1, _, wf2, _ = ctx.ps
        *grads, dx = simnn_net_backward(ctx.saved, dout, w1, w2, wf1, wf2, ctx.dt, ctx.need_dx)
        if dx is not None:
            dx = dx.view(ctx.x_shape).to(ctx.x_dtype)
        return (dx, *grads, None)
