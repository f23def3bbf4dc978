"""Tensor-level wrappers over the C ABI (include/gdm.h).

PyTorch is used here only for device memory (caching allocator) and the current HIP stream; every wrapper checks
that its tensors live on a HIP device and raises otherwise -- there is no CPU path.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F32, GdmError, check  # noqa: F401

_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def torch_dtype(dt):
    return _TORCH_DT[dt]


def gdm_dtype(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise GdmError(f"unsupported tensor dtype {t.dtype} (fp32 or bf16 only)")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise GdmError("gan_des_midi_music_gen_amd ops run on a HIP device only (tensor is on "
                           f"{t.device}); there is no CPU fallback")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-entry-point device timing (bench.py's roofline leg): while ``_TIMED["name"]`` names a C-ABI function,
# every call of it is bracketed by two HIP events recorded on the stream the kernel is launched on.
_TIMED = {"name": None, "events": []}


def time_entry_point(name):
    """Start (name) / stop (None) collecting (start, end) event pairs for one C-ABI entry point."""
    _TIMED["name"] = name
    _TIMED["events"] = []


def timed_durations_ms():
    """Average and count of the collected launches (call after a device synchronize)."""
    ev = _TIMED["events"]
    if not ev:
        return 0.0, 0
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)


def _call(name, *args):
    fn = getattr(_lib.load(), name)
    if _TIMED["name"] == name:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _TIMED["events"].append((a, b))
    else:
        rc = fn(*args)
    check(rc, name)


_ws_cache = {}
_ws_retired = []          # superseded buffers a captured hipGraph may still point into (never freed)
_ws_pinned = set()        # keys that were used while a stream capture was in progress
_ws_ns = threading.local()


@contextlib.contextmanager
def workspace_namespace(ns):
    """Scratch buffers handed out inside this context belong to ``ns`` (any hashable) instead of being shared by
    everything that runs on the same stream.  Every hipGraph capture gets a namespace of its own: torch captures all
    graphs on ONE shared side stream, so without it two graphs that are later replayed CONCURRENTLY (a trainer's main
    graph and its generator graph, on different streams) would have the same scratch buffer baked in."""
    prev = getattr(_ws_ns, "ns", None)
    _ws_ns.ns = ns
    try:
        yield
    finally:
        _ws_ns.ns = prev


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream, namespace); safe because every consumer is ordered on the same
    stream (and graphs that may run side by side capture under different namespaces, see workspace_namespace).

    A captured hipGraph has the address of the buffer it saw baked in, so a buffer that was handed out during a capture
    is never released: when a later call (eager, or later in the same capture) needs more room, the old buffer is
    retired -- kept alive, so the graph keeps writing into memory that is still ours -- and a larger one takes its
    place.  (Freeing it, as the first version did, let the allocator hand the same bytes to another tensor of the
    graph.)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, getattr(_ws_ns, "ns", None))
    capturing = torch.cuda.is_current_stream_capturing()
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and (capturing or key in _ws_pinned):
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    if capturing:
        _ws_pinned.add(key)
    return buf


def default_split_k(m, n, k, compute):
    """Split K only when the output has too few tiles to occupy the 256 CUs; aim at ~256 workgroups."""
    tile, kt = (128, 32) if compute == BF16 else (64, 32)
    tiles = ((m + tile - 1) // tile) * ((n + tile - 1) // tile)
    ktiles = (k + kt - 1) // kt
    if tiles >= 128 or ktiles < 32:
        return 1
    # bf16: ~256 workgroups of 128x128 (the fast kernel keeps two k-tiles in flight); exact fp32: 64x64 tiles hold 17 KB
    # of LDS, so ~768 workgroups (three per CU) hide each other's load latency
    target = 256 if compute == BF16 else 768
    return max(1, min(ktiles // 8, (target + tiles - 1) // tiles))


def gemm(a, b, *, bias_n=None, bias_m=None, act=ACT_NONE, slope=0.0, out_dtype=None, compute=F32, split_k=None,
         out=None):
    """act(a @ b + bias): a (M,K), b (K,N) are arbitrary strided 2-D views (``.t()`` is free)."""
    _need_gpu(a, b, bias_n, bias_m, out)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0], (a.shape, b.shape)
    m, k = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=_TORCH_DT[F32 if out_dtype is None else out_dtype], device=a.device)
    assert out.shape == (m, n)
    if split_k is None:
        split_k = default_split_k(m, n, k, compute)
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = split_k * m * n * 4
        ws = workspace(ws_bytes, a.device)
    lib = _lib.load()
    _call("gdm_gemm", _p(a), gdm_dtype(a), a.stride(0), a.stride(1), _p(b), gdm_dtype(b), b.stride(0), b.stride(1),
                       _p(out), gdm_dtype(out), out.stride(0), out.stride(1), m, n, k, _p(bias_n), _p(bias_m), act,
                       float(slope), compute, split_k, _p(ws), ws_bytes, _stream())
    return out


def bce_with_logits(x, target, *, grad_scale=1.0, want_grad=True, fuse_sigmoid_backward=False, loss_out=None,
                    accumulate_loss=False, dx_out=None):
    """Returns (loss (1,) fp32 tensor, dx or None).  x: (n,) fp32 contiguous."""
    _need_gpu(x, loss_out, dx_out)
    x = x.reshape(-1)
    assert x.dtype == torch.float32 and x.is_contiguous()
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=x.device)
    dx = dx_out if dx_out is not None else (torch.empty_like(x) if want_grad else None)
    if dx is not None:
        assert dx.is_contiguous() and dx.numel() == x.numel() and dx.dtype == torch.float32
    _call("gdm_bce_with_logits", _p(x), float(target), x.numel(), float(grad_scale), _p(loss), _p(dx),
                                           1 if fuse_sigmoid_backward else 0, 1 if accumulate_loss else 0, _stream())
    return loss, dx


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, grad_scale=1.0):
    _need_gpu(p, g, m, v)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    _call("gdm_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), int(step), float(lr), float(beta1),
                                    float(beta2), float(eps), float(grad_scale), _stream())


def adam_hyper(device, lr, beta1, beta2, eps, grad_scale=1.0, step=0):
    """8-float device record for adam_step_dev: {step (int bits), lr, beta1, beta2, eps, grad_scale, -, -}."""
    h = torch.zeros(8, dtype=torch.float32)
    h[1], h[2], h[3], h[4], h[5] = lr, beta1, beta2, eps, grad_scale
    h.view(torch.int32)[0] = int(step)
    return h.to(device)


def adam_step_dev(p, g, m, v, hyper):
    """Adam step whose step counter / hyper-parameters live in `hyper` on the device (graph-replayable)."""
    _need_gpu(p, g, m, v, hyper)
    assert hyper.numel() == 8 and hyper.dtype == torch.float32
    _call("gdm_adam_step_dev", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _stream())


def adam_step_dev_pc(p, g_pc, m, v, n, c, pix, shadow_pc, hyper, advance_step=False):
    """Adam on one (n, c, pix) parameter whose gradient g_pc and operand copy shadow_pc are laid out (n, pix, c)."""
    _need_gpu(p, g_pc, m, v, shadow_pc, hyper)
    for t in (p, g_pc, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n * c * pix, (t.shape, n, c, pix)
    assert shadow_pc.is_contiguous() and shadow_pc.numel() == n * c * pix
    _call("gdm_adam_step_dev_pc", _p(p), _p(g_pc), _p(m), _p(v), n, c, pix, _p(shadow_pc), gdm_dtype(shadow_pc),
          _p(hyper), 1 if advance_step else 0, _stream())


SIMNN_ADAM_RECORD_INTS = 1056     # GDM_SIMNN_ADAM_RECORD_INTS (include/gdm.h)


def simnn_adam_step(p_big, g_big_pc, m_big, v_big, n, c, pix, shadow_pc, p_small, g_small, m_small, v_small, conv2_weight,
                    pack, hyper, done):
    """Model 1's whole discriminator optimizer step in one launch (gdm_simnn_adam_step): transposing Adam on the
    (n, c, pix) parameter, plain Adam on the contiguous small range (which holds ``conv2_weight``), re-pack of conv2's
    MFMA images into ``pack``; the device step counter in ``hyper`` is advanced.  ``done``: the optimizer's int32 record
    (SIMNN_ADAM_RECORD_INTS zeros at first; zero it again after rewriting ``hyper`` from the host)."""
    _need_gpu(p_big, g_big_pc, m_big, v_big, shadow_pc, p_small, g_small, m_small, v_small, conv2_weight, pack, hyper, done)
    for t in (p_big, g_big_pc, m_big, v_big):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n * c * pix
    for t in (p_small, g_small, m_small, v_small):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p_small.numel()
    assert shadow_pc.is_contiguous() and shadow_pc.numel() == n * c * pix and conv2_weight.numel() == 4608
    assert hyper.numel() == 8 and done.numel() == SIMNN_ADAM_RECORD_INTS and done.dtype == torch.int32
    dt = gdm_dtype(shadow_pc)
    assert pack.numel() == _lib.load().gdm_simnn_conv2_pack_bytes(dt)
    _call("gdm_simnn_adam_step", _p(p_big), _p(g_big_pc), _p(m_big), _p(v_big), n, c, pix, _p(shadow_pc), _p(p_small),
          _p(g_small), _p(m_small), _p(v_small), p_small.numel(), _p(conv2_weight), _p(pack), dt, _p(hyper), _p(done),
          _stream())


def bn_act_fwd(y, gamma, beta, running_mean, running_var, nbt, *, act, out_dtype=F32, training=True, momentum=0.1,
               eps=1e-5):
    """y (rows, C) fp32 -> (out, save_mean, save_invstd)."""
    _need_gpu(y, gamma, beta, running_mean, running_var, nbt)
    assert y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous()
    rows, c = y.shape
    out = torch.empty((rows, c), dtype=_TORCH_DT[out_dtype], device=y.device)
    mean = torch.empty(c, dtype=torch.float32, device=y.device)
    invstd = torch.empty(c, dtype=torch.float32, device=y.device)
    lib = _lib.load()
    nb = lib.gdm_bn_workspace_bytes(rows, c)
    ws = workspace(nb, y.device)
    _call("gdm_bn_act_fwd", _p(y), rows, c, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(nbt),
                             float(momentum), float(eps), act, _p(out), out_dtype, _p(mean), _p(invstd),
                             1 if training else 0, _p(ws), nb, _stream())
    return out, mean, invstd


def bn_stats(y, running_mean, running_var, nbt, *, momentum=0.1, eps=1e-5):
    """Training-mode batch statistics of y (rows, C) fp32 -> (mean, invstd); running statistics updated."""
    _need_gpu(y, running_mean, running_var, nbt)
    assert y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous()
    rows, c = y.shape
    mean = torch.empty(c, dtype=torch.float32, device=y.device)
    invstd = torch.empty(c, dtype=torch.float32, device=y.device)
    nb = _lib.load().gdm_bn_workspace_bytes(rows, c)
    ws = workspace(nb, y.device)
    _call("gdm_bn_stats", _p(y), rows, c, _p(running_mean), _p(running_var), _p(nbt), float(momentum), float(eps),
          _p(mean), _p(invstd), _p(ws), nb, _stream())
    return mean, invstd


def bn_partials(y):
    """Per-row-chunk Welford triples (chunks, C, 3) of y (rows, C) fp32 -- the first half of training-mode batch norm."""
    _need_gpu(y)
    assert y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous()
    rows, c = y.shape
    chunks = _lib.load().gdm_bn_partial_chunks(rows)
    partials = torch.empty((chunks, c, 3), dtype=torch.float32, device=y.device)
    _call("gdm_bn_partials", _p(y), rows, c, _p(partials), _stream())
    return partials


def bn_apply(y, gamma, beta, mean, invstd, *, act, out_dtype=F32):
    """act((y - mean) * invstd * gamma + beta) with given per-column statistics."""
    _need_gpu(y, gamma, beta, mean, invstd)
    assert y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous()
    rows, c = y.shape
    out = torch.empty((rows, c), dtype=_TORCH_DT[out_dtype], device=y.device)
    _call("gdm_bn_apply", _p(y), rows, c, _p(gamma), _p(beta), _p(mean), _p(invstd), act, _p(out), out_dtype, _stream())
    return out


def bn_finalize(partials, chunks, rows, c, running_mean, running_var, nbt, *, momentum=0.1, eps=1e-5):
    """Merge (chunks, c, 3) Welford partials -> (mean, invstd); running statistics updated."""
    _need_gpu(partials, running_mean, running_var, nbt)
    assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.numel() >= chunks * c * 3
    mean = torch.empty(c, dtype=torch.float32, device=partials.device)
    invstd = torch.empty(c, dtype=torch.float32, device=partials.device)
    _call("gdm_bn_finalize", _p(partials), int(chunks), int(rows), int(c), float(momentum), float(eps), _p(running_mean),
          _p(running_var), _p(nbt), _p(mean), _p(invstd), _stream())
    return mean, invstd


def simnn_gen_pack(w1, w2, w3, out=None):
    """bf16 GEMM images of the generator's conv1 / conv2 / conv3 weights (rebuild when they change)."""
    _need_gpu(w1, w2, w3, out)
    assert w1.dim() == 4 and w1.shape[1:] == (128, 4, 4) and w1.shape[0] <= 128
    assert w2.shape == (128, 64, 4, 4) and w3.shape == (64, 32, 4, 4)
    for w in (w1, w2, w3):
        assert w.is_contiguous() and w.dtype == torch.float32
    nb = _lib.load().gdm_simnn_gen_pack_bytes()
    pack = out if out is not None else torch.empty(nb, dtype=torch.uint8, device=w2.device)
    assert pack.numel() == nb
    _call("gdm_simnn_gen_pack", _p(w1), int(w1.shape[0]), _p(w2), _p(w3), _p(pack), _stream())
    return pack


def simnn_gen_first(noise2d, pack, running_mean, running_var, nbt, *, momentum=0.1, eps=1e-5):
    """Generator layer 1 + its batch statistics in one launch: noise (B, noise_dim) -> (y1 (B*16, 128), mean, invstd)."""
    _need_gpu(noise2d, pack, running_mean, running_var, nbt)
    b, nd = noise2d.shape
    assert noise2d.dtype == torch.float32 and noise2d.is_contiguous() and 2 <= b <= 256 and nd <= 128
    y1 = torch.empty((b * 16, 128), dtype=torch.float32, device=noise2d.device)
    mean = torch.empty(128, dtype=torch.float32, device=noise2d.device)
    invstd = torch.empty(128, dtype=torch.float32, device=noise2d.device)
    _call("gdm_simnn_gen_first", _p(noise2d), b, nd, _p(pack), _p(y1), float(momentum), float(eps), _p(running_mean),
          _p(running_var), _p(nbt), _p(mean), _p(invstd), _stream())
    return y1, mean, invstd


def simnn_gen_convt_bn(layer, yin, mean, invstd, gamma, beta, b, pack):
    """Fused BN+ReLU-on-load + ConvTranspose2d(k4,s2,p1) of generator layer 2 or 3.
    Returns (yout (B*OH*OW, Cout) fp32 raw, partials (chunks, Cout, 3), chunks)."""
    _need_gpu(yin, mean, invstd, gamma, beta, pack)
    cin, cout, ih = (128, 64, 4) if layer == 2 else (64, 32, 8)
    assert yin.shape == (b * ih * ih, cin) and yin.dtype == torch.float32 and yin.is_contiguous()
    chunks = _lib.load().gdm_simnn_gen_convt_chunks(layer, b)
    yout = torch.empty((b * 4 * ih * ih, cout), dtype=torch.float32, device=yin.device)
    partials = torch.empty((chunks, cout, 3), dtype=torch.float32, device=yin.device)
    _call("gdm_simnn_gen_convt_bn", layer, _p(yin), _p(mean), _p(invstd), _p(gamma), _p(beta), b, _p(pack), _p(yout),
          _p(partials), _stream())
    return yout, partials, chunks


def simnn_gen_last(yin, mean, invstd, gamma, beta, w4, b):
    """BN+ReLU on load, ConvTranspose2d(32->1,k5), sigmoid: y3 (B*256, 32) -> (B, 1, 20, 20)."""
    _need_gpu(yin, mean, invstd, gamma, beta, w4)
    assert yin.shape == (b * 256, 32) and yin.is_contiguous() and w4.shape == (32, 1, 5, 5) and w4.is_contiguous()
    out = torch.empty((b, 1, 20, 20), dtype=torch.float32, device=yin.device)
    _call("gdm_simnn_gen_last", _p(yin), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w4), b, _p(out), _stream())
    return out


def bn_act_bwd(dout, out, y, gamma, mean, invstd, *, act):
    _need_gpu(dout, out, y, gamma, mean, invstd)
    rows, c = y.shape
    assert dout.dtype == out.dtype and dout.is_contiguous() and out.is_contiguous() and y.is_contiguous()
    dy = torch.empty_like(y)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    lib = _lib.load()
    nb = lib.gdm_bn_workspace_bytes(rows, c)
    ws = workspace(nb, y.device)
    _call("gdm_bn_act_bwd", _p(dout), _p(out), gdm_dtype(out), _p(y), rows, c, _p(gamma), _p(mean), _p(invstd), act,
                             _p(dy), _p(dgamma), _p(dbeta), _p(ws), nb, _stream())
    return dy, dgamma, dbeta


def bias_act_fwd(x, bias, *, act, slope=0.0, out_dtype=F32):
    _need_gpu(x, bias)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=_TORCH_DT[out_dtype], device=x.device)
    _call("gdm_bias_act_fwd", _p(x), _p(bias), x.shape[0], x.shape[1], act, float(slope), _p(out), out_dtype,
                                       _stream())
    return out


def act_bwd(dout, out, *, act, slope=0.0):
    _need_gpu(dout, out)
    assert dout.dtype == out.dtype and dout.is_contiguous() and out.is_contiguous() and dout.numel() == out.numel()
    dx = torch.empty_like(dout)
    _call("gdm_act_bwd", _p(dout), _p(out), gdm_dtype(out), out.numel(), act, float(slope), _p(dx),
                                  _stream())
    return dx


def colsum(x, out=None):
    _need_gpu(x, out)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    assert out.numel() == c and out.is_contiguous() and out.dtype == torch.float32
    nb = ((rows + 63) // 64 + 1) * c * 4
    ws = workspace(nb, x.device)
    _call("gdm_colsum", _p(x), gdm_dtype(x), rows, c, _p(out), _p(ws), nb, _stream())
    return out


def cast(x, dt):
    _need_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_TORCH_DT[dt], device=x.device)
    _call("gdm_cast", _p(x), gdm_dtype(x), _p(out), dt, x.numel(), _stream())
    return out


class NonFiniteError(RuntimeError):
    """A NaN or Inf entered or left a step (what torch's anomaly mode reports in the reference, network_tests.py:211)."""


def nonfinite_count(tensors, counter=None):
    """counter (1,) int32 device tensor += number of NaN / Inf elements in ``tensors`` (fp32 / bf16 device tensors);
    enqueued on the current stream, no synchronisation.  Returns the counter."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    _need_gpu(*tensors)
    if counter is None:
        counter = torch.zeros(1, dtype=torch.int32, device=tensors[0].device)
    for t in tensors:
        t = t if t.is_contiguous() else t.contiguous()
        _call("gdm_nonfinite_count", _p(t), gdm_dtype(t), t.numel(), _p(counter), _stream())
    return counter


def assert_finite(tensors, what):
    """Synchronising check used under torch.autograd.set_detect_anomaly(True): raises NonFiniteError naming ``what``."""
    n = int(nonfinite_count(tensors).item())
    if n:
        raise NonFiniteError(f"{what}: {n} non-finite value(s) (NaN / Inf)")


# ------------------------------------------------------------------------------------------ model 1 conv trunk
def simnn_code1_width(w1):
    """Last dimension of the int64 tensor that holds conv1's pool / ReLU codes of a (.., w1)-wide pooled map: 8 bytes
    per pixel, rows padded to whole quads of four pixels (include/gdm.h, gdm_simnn_conv1_fwd)."""
    return (w1 + 3) // 4 * 4


def simnn_conv1_fwd(x, w, bias, dt, out=None, x1=None):
    """out = (p1, code1) preallocated (e.g. halves of a 2B batch buffer) or None; code1: (b, h1, simnn_code1_width(w1))
    int64 -- an opaque byte image, only ever handed back to the backward entry points.  ``x1``: a second input tensor
    of the same geometry; the batch is then [x ; x1] in ONE launch (gdm_simnn_conv1_fwd_pair)."""
    _need_gpu(x, w, bias, x1)
    assert x.dim() == 3 and x.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    b, h, wd = x.shape
    bsplit = b
    if x1 is not None:
        assert x1.dtype == torch.float32 and x1.is_contiguous() and x1.shape[1:] == x.shape[1:]
        b += x1.shape[0]
    h1, w1 = (h + 1) // 2, (wd + 1) // 2
    if out is not None:
        p1, code1 = out
        assert p1.shape == (b, h1, w1, 16) and p1.is_contiguous() and p1.dtype == _TORCH_DT[dt]
        assert code1.shape == (b, h1, simnn_code1_width(w1)) and code1.is_contiguous() and code1.dtype == torch.int64
    else:
        p1 = torch.empty((b, h1, w1, 16), dtype=_TORCH_DT[dt], device=x.device)
        code1 = torch.empty((b, h1, simnn_code1_width(w1)), dtype=torch.int64, device=x.device)
    _call("gdm_simnn_conv1_fwd_pair", _p(x), _p(x1), bsplit, _p(w), _p(bias), b, h, wd, _p(p1), _p(code1), dt, _stream())
    return p1, code1


def simnn_conv2_pack(w, dt, out=None):
    """Packed MFMA operand images (forward + flipped backward) of conv2's weight; rebuild when w changes."""
    _need_gpu(w, out)
    assert w.shape == (32, 16, 3, 3) and w.dtype == torch.float32 and w.is_contiguous()
    lib = _lib.load()
    nbytes = lib.gdm_simnn_conv2_pack_bytes(dt)
    pack = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    assert pack.numel() == nbytes and pack.dtype == torch.uint8
    _call("gdm_simnn_conv2_pack", _p(w), dt, _p(pack), _stream())
    return pack


def simnn_conv2_fwd(p1, pack, bias):
    """p1 (B,H1,W1,16) -> p2 (B,H2,W2,32) channels-last, code2 (B,H2,W2,16) uint8 (one byte per channel pair)."""
    _need_gpu(p1, pack, bias)
    assert p1.dim() == 4 and p1.shape[3] == 16 and p1.is_contiguous()
    b, h1, w1, _ = p1.shape
    h2, w2 = h1 // 2, w1 // 2
    p2 = torch.empty((b, h2, w2, 32), dtype=p1.dtype, device=p1.device)
    code2 = torch.empty((b, h2, w2, 16), dtype=torch.uint8, device=p1.device)
    _call("gdm_simnn_conv2_fwd", _p(p1), _p(pack), _p(bias), b, h1, w1, _p(p2), _p(code2), gdm_dtype(p1), _stream())
    return p2, code2


def simnn_conv2_bwd_data(dp2, code2, pack, h1, w1):
    _need_gpu(dp2, code2, pack)
    assert dp2.is_contiguous() and code2.is_contiguous() and dp2.shape[-1] == 32
    b = dp2.shape[0]
    dp1 = torch.empty((b, h1, w1, 16), dtype=dp2.dtype, device=dp2.device)
    _call("gdm_simnn_conv2_bwd_data", _p(dp2), _p(code2), _p(pack), b, h1, w1, _p(dp1), gdm_dtype(dp2), _stream())
    return dp1


def simnn_conv2_bwd_fused(dp2, code2, pack, code1, x0, x1=None, out=None, want_dp1=False):
    """conv2 data gradient with conv1's weight gradient fused in (dp1 stays in registers).

    x0 (B0,H,W) and optionally x1 (B1,H,W) are the inputs of samples [0,B0) and [B0,B0+B1).
    Returns (dw1, db1, dp1 or None)."""
    _need_gpu(dp2, code2, pack, code1, x0, x1)
    assert dp2.is_contiguous() and code2.is_contiguous() and code1.is_contiguous() and x0.is_contiguous()
    b = dp2.shape[0]
    h, wd = x0.shape[1], x0.shape[2]
    h1, w1 = (h + 1) // 2, (wd + 1) // 2
    assert code1.shape == (b, h1, simnn_code1_width(w1)) and code1.dtype == torch.int64, code1.shape
    bsplit = x0.shape[0]
    if x1 is not None:
        assert x1.is_contiguous() and x1.shape[1:] == x0.shape[1:] and bsplit + x1.shape[0] == b
    else:
        assert bsplit == b
    if out is not None:
        dw, db = out
        assert dw.numel() == 64 and db.numel() == 16 and dw.is_contiguous() and db.is_contiguous()
    else:
        dw = torch.empty((16, 1, 2, 2), dtype=torch.float32, device=x0.device)
        db = torch.empty(16, dtype=torch.float32, device=x0.device)
    dp1 = torch.empty((b, h1, w1, 16), dtype=dp2.dtype, device=dp2.device) if want_dp1 else None
    lib = _lib.load()
    nb = lib.gdm_simnn_conv2_bwd_fused_workspace_bytes(b, h1, w1)
    ws = workspace(nb, x0.device)
    _call("gdm_simnn_conv2_bwd_fused", _p(dp2), _p(code2), _p(pack), b, h1, w1, _p(code1), _p(x0), _p(x1), bsplit, h,
          wd, _p(dp1), gdm_dtype(dp2), _p(ws), nb, _stream())
    _call("gdm_simnn_conv2_bwd_fused_finish", b, h1, w1, _p(dw), _p(db), _p(ws), nb, _stream())
    return dw, db, dp1


def simnn_conv2_bwd_weight(dp2, code2, p1, out=None):
    _need_gpu(dp2, code2, p1)
    assert dp2.is_contiguous() and code2.is_contiguous() and p1.is_contiguous() and dp2.dtype == p1.dtype
    b, h1, w1, _ = p1.shape
    if out is not None:
        dw, db = out
        assert dw.numel() == 4608 and db.numel() == 32 and dw.is_contiguous() and db.is_contiguous()
    else:
        dw = torch.empty((32, 16, 3, 3), dtype=torch.float32, device=p1.device)
        db = torch.empty(32, dtype=torch.float32, device=p1.device)
    lib = _lib.load()
    nb = lib.gdm_simnn_conv2_bwd_weight_workspace_bytes(b, h1, w1)
    ws = workspace(nb, p1.device)
    _call("gdm_simnn_conv2_bwd_weight", _p(dp2), _p(code2), _p(p1), b, h1, w1, _p(dw), _p(db), gdm_dtype(p1), _p(ws),
                                         nb, _stream())
    return dw, db


def simnn_conv1_bwd_weight(dp1, code1, x, out=None, accumulate=False):
    _need_gpu(dp1, code1, x)
    assert dp1.is_contiguous() and code1.is_contiguous() and x.is_contiguous()
    b, h, wd = x.shape
    assert dp1.shape[0] == b and code1.shape == (b, (h + 1) // 2, simnn_code1_width((wd + 1) // 2))
    if out is not None:
        dw, db = out
        assert dw.numel() == 64 and db.numel() == 16 and dw.is_contiguous() and db.is_contiguous()
    else:
        assert not accumulate
        dw = torch.empty((16, 1, 2, 2), dtype=torch.float32, device=x.device)
        db = torch.empty(16, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nb = lib.gdm_simnn_conv1_bwd_weight_workspace_bytes(b, h, wd)
    ws = workspace(nb, x.device)
    _call("gdm_simnn_conv1_bwd_weight", _p(dp1), _p(code1), _p(x), b, h, wd, _p(dw), _p(db), gdm_dtype(dp1),
                                         1 if accumulate else 0, _p(ws), nb, _stream())
    return dw, db


def simnn_conv1_bwd_data(dp1, code1, w, h, wd):
    """Gradient w.r.t. the (B,H,W) spectrogram input from dp1 (B,H1,W1,16), conv1's pool/ReLU codes and weight."""
    _need_gpu(dp1, code1, w)
    assert dp1.is_contiguous() and code1.is_contiguous() and w.is_contiguous() and w.numel() == 64
    b = dp1.shape[0]
    assert dp1.shape == (b, (h + 1) // 2, (wd + 1) // 2, 16)
    assert code1.shape == (b, (h + 1) // 2, simnn_code1_width((wd + 1) // 2))
    dx = torch.empty((b, h, wd), dtype=torch.float32, device=dp1.device)
    _call("gdm_simnn_conv1_bwd_data", _p(dp1), _p(code1), _p(w), b, h, wd, _p(dx), gdm_dtype(dp1), _stream())
    return dx


def simnn_head(h1, w2, b2, n0, y0, y1, *, loss_out, accumulate_loss=False, want_grad=True, grad_out=None, dh_dtype=F32):
    """Fused fc2 + sigmoid + BCE(+backward) of model 1's discriminator head.

    h1 (n,128) fp32; rows [0,n0) carry label y0, the rest y1.  grad_out = (dw2, db2, db1) views to fill.
    dh_dtype: dtype of the returned dh1 (BF16 in bf16 mode: it is only ever the operand of fc1's two backward GEMMs).
    Returns (prob (n,), dh1 (n,128) or None, (dw2, db2, db1) or None)."""
    _need_gpu(h1, w2, b2, loss_out)
    assert h1.dim() == 2 and h1.shape[1] == 128 and h1.is_contiguous() and h1.dtype == torch.float32
    n = h1.shape[0]
    prob = torch.empty(n, dtype=torch.float32, device=h1.device)
    dh1 = dw2 = db2 = db1 = None
    if want_grad:
        dh1 = torch.empty(h1.shape, dtype=_TORCH_DT[dh_dtype], device=h1.device)
        if grad_out is not None:
            dw2, db2, db1 = grad_out
            assert dw2.numel() == 128 and db2.numel() == 1 and db1.numel() == 128
        else:
            dw2 = torch.empty((1, 128), dtype=torch.float32, device=h1.device)
            db2 = torch.empty(1, dtype=torch.float32, device=h1.device)
            db1 = torch.empty(128, dtype=torch.float32, device=h1.device)
    nb = _lib.load().gdm_simnn_head_workspace_bytes(n)
    ws = workspace(nb, h1.device)
    _call("gdm_simnn_head", _p(h1), _p(w2), _p(b2), n, n0, float(y0), float(y1), _p(prob), _p(loss_out),
          1 if accumulate_loss else 0, _p(dh1), int(dh_dtype), _p(dw2), _p(db2), _p(db1), _p(ws), nb, _stream())
    return prob, dh1, ((dw2, db2, db1) if want_grad else None)


# ------------------------------------------------------------------------------------------ model 2 fused kernels
def linear_bn_act_max_rows():
    return _lib.load().gdm_linear_bn_act_max_rows()


def linear_bn_act_fwd(x, w, bias, gamma, beta, running_mean, running_var, nbt, *, act, training=True, momentum=0.1,
                      eps=1e-5, save_y=False, groups=1, stat_repeats=1):
    """Linear + BatchNorm1d + activation in one launch (rows per group <= linear_bn_act_max_rows()).

    groups > 1: x holds ``groups`` batches of equal size stacked along dim 0; each is normalised with its own batch
    statistics and the running statistics are updated in that order (mean / invstd come back as (groups, N)).
    stat_repeats: apply each running-statistics update that many times.
    Returns (out, y or None, save_mean, save_invstd)."""
    _need_gpu(x, w, bias, gamma, beta, running_mean, running_var, nbt)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.is_contiguous() and w.is_contiguous()
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    rows, k = x.shape
    assert rows % groups == 0
    m = rows // groups
    n = w.shape[0]
    out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    y = torch.empty((rows, n), dtype=torch.float32, device=x.device) if save_y else None
    shape = (n,) if groups == 1 else (groups, n)
    mean = torch.empty(shape, dtype=torch.float32, device=x.device)
    invstd = torch.empty(shape, dtype=torch.float32, device=x.device)
    _call("gdm_linear_bn_act_fwd", _p(x), _p(w), _p(bias), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
          _p(nbt), float(momentum), float(eps), act, 1 if training else 0, m, n, k, _p(y), _p(out), _p(mean),
          _p(invstd), int(groups), int(stat_repeats), _stream())
    return out, y, mean, invstd


def linear_bn_act_fwd_multi(jobs, *, act, training=True, momentum=0.1, eps=1e-5):
    """1 or 2 independent Linear+BatchNorm1d+activation blocks in ONE launch.  jobs: list of dicts with the positional
    arguments of linear_bn_act_fwd (x, w, bias, gamma, beta, running_mean, running_var, nbt) and optional groups /
    stat_repeats.  Returns [(out, save_mean, save_invstd)] per job (forward only)."""
    assert 1 <= len(jobs) <= 2
    arr = (_lib.LinearBnJob * len(jobs))()
    res = []
    for i, j in enumerate(jobs):
        x, w = j["x"], j["w"]
        _need_gpu(x, w, j["bias"], j["gamma"], j["beta"], j["running_mean"], j["running_var"], j["nbt"])
        assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.is_contiguous() and w.is_contiguous()
        assert x.dtype == torch.float32 and w.dtype == torch.float32
        groups, reps = int(j.get("groups", 1)), int(j.get("stat_repeats", 1))
        rows, k = x.shape
        assert rows % groups == 0
        n = w.shape[0]
        out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
        shape = (n,) if groups == 1 else (groups, n)
        mean = torch.empty(shape, dtype=torch.float32, device=x.device)
        invstd = torch.empty(shape, dtype=torch.float32, device=x.device)
        a = arr[i]
        a.x, a.w, a.bias, a.gamma, a.beta = _p(x), _p(w), _p(j["bias"]), _p(j["gamma"]), _p(j["beta"])
        a.running_mean, a.running_var, a.num_batches_tracked = _p(j["running_mean"]), _p(j["running_var"]), _p(j["nbt"])
        a.y_out, a.out, a.save_mean, a.save_invstd = None, _p(out), _p(mean), _p(invstd)
        a.M, a.N, a.K, a.groups, a.stat_repeats = rows // groups, n, k, groups, reps
        res.append((out, mean, invstd))
    _call("gdm_linear_bn_act_fwd_multi", ctypes.cast(arr, ctypes.c_void_p), len(jobs), float(momentum), float(eps), act,
          1 if training else 0, _stream())
    return res


def concat_cols_multi(pairs, outs=None):
    """[torch.cat((a, b), dim=1) for a, b in pairs] for up to four small fp32 pairs in ONE launch (``outs``: contiguous
    destination tensors, e.g. row blocks of one buffer)."""
    assert 1 <= len(pairs) <= 4 and (outs is None or len(outs) == len(pairs))
    arr = (_lib.ConcatJob * len(pairs))()
    res = []
    for i, (a, b) in enumerate(pairs):
        _need_gpu(a, b)
        assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.is_contiguous() and b.is_contiguous()
        assert a.dtype == torch.float32 and b.dtype == torch.float32
        shape = (a.shape[0], a.shape[1] + b.shape[1])
        out = outs[i] if outs is not None else torch.empty(shape, dtype=torch.float32, device=a.device)
        assert tuple(out.shape) == shape and out.is_contiguous() and out.dtype == torch.float32
        q = arr[i]
        q.a, q.b, q.out, q.M, q.Ka, q.Kb = _p(a), _p(b), _p(out), a.shape[0], a.shape[1], b.shape[1]
        res.append(out)
    _call("gdm_concat_cols_multi", ctypes.cast(arr, ctypes.c_void_p), len(pairs), _stream())
    return res


def dcnn_fused_supported(t):
    return bool(_lib.load().gdm_dcnn_fused_supported(int(t)))


def dcnn_pack(w1, b1, w2, b2, wfc, bfc, t, out=None):
    """bf16 MFMA weight images + permuted fc weight + biases for dcnn_fused; refresh after every weight update."""
    _need_gpu(w1, b1, w2, b2, wfc, bfc, out)
    for a in (w1, b1, w2, b2, wfc, bfc):
        assert a.dtype == torch.float32 and a.is_contiguous()
    nbytes = _lib.load().gdm_dcnn_pack_bytes(int(t))
    pack = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    assert pack.numel() == nbytes
    _call("gdm_dcnn_pack", _p(w1), _p(b1), _p(w2), _p(b2), _p(wfc), _p(bfc), int(t), _p(pack), _stream())
    return pack


def dcnn_fused(xa, planes, t, ya, yb, pack, *, loss_out, accumulate_loss=False, want_grad=True, grad_out=None,
               adam=None):
    """DiscriminatorCNN forward + BCE loss (+ backward) in one persistent kernel.

    xa: (Ba,2,128,T) fp32 contiguous or None (label ya); planes: (p0, p1) each (Bb,128,T) or None (label yb).
    grad_out: 6 tensors (dw1, db1, dw2, db2, dwfc, dbfc) to fill.  Returns (logits (B,), grads or None).
    adam = dict(params=[6 tensors], exp_avg=[6], exp_avg_sq=[6], hyper=(8,) fp32, done=(1,) int32): the optimizer step
    and the refresh of ``pack`` ride the gradient's final summation (gdm_dcnn_fused_adam; one rank only)."""
    _need_gpu(xa, pack, loss_out)
    ba = 0 if xa is None else xa.shape[0]
    p0 = p1 = None
    bb = 0
    if planes is not None:
        p0, p1 = planes
        _need_gpu(p0, p1)
        assert p0.is_contiguous() and p1.is_contiguous() and p0.shape == p1.shape and p0.shape[1:] == (128, t)
        assert p0.dtype == torch.float32 and p1.dtype == torch.float32
        bb = p0.shape[0]
    if xa is not None:
        assert xa.is_contiguous() and xa.dtype == torch.float32 and xa.shape[1:] == (2, 128, t)
    b = ba + bb
    dev = pack.device
    logits = torch.empty(b, dtype=torch.float32, device=dev)
    grads = None
    if want_grad:
        if grad_out is not None:
            grads = list(grad_out)
        else:
            k = 32 * 32 * (((t // 2) - 2) // 2 + 1)
            grads = [torch.empty(s, dtype=torch.float32, device=dev)
                     for s in ((16, 2, 4, 4), (16,), (32, 16, 4, 4), (32,), (1, k), (1,))]
        for g in grads:
            assert g.is_contiguous() and g.dtype == torch.float32
    lib = _lib.load()
    nb = lib.gdm_dcnn_fused_workspace_bytes(b, int(t), 1 if want_grad else 0)
    ws = workspace(nb, dev)
    gp = [_p(g) for g in grads] if grads else [None] * 6
    if adam is not None:
        assert want_grad, "the fused optimizer step needs the gradient"
        rec = _lib.DcnnAdam()
        for field, key in (("param", "params"), ("exp_avg", "exp_avg"), ("exp_avg_sq", "exp_avg_sq")):
            ts = adam[key]
            assert len(ts) == 6
            for i, (tq, g) in enumerate(zip(ts, grads)):
                _need_gpu(tq)
                assert tq.is_contiguous() and tq.dtype == torch.float32 and tq.numel() == g.numel(), (field, i)
                getattr(rec, field)[i] = tq.data_ptr()
        hyper, done = adam["hyper"], adam["done"]
        _need_gpu(hyper, done)
        assert hyper.numel() == 8 and hyper.dtype == torch.float32 and done.numel() == 1 and done.dtype == torch.int32
        rec.hyper, rec.done = hyper.data_ptr(), done.data_ptr()
        _call("gdm_dcnn_fused_adam", _p(xa), ba, _p(p0), _p(p1), b, int(t), float(ya), float(yb), _p(pack), _p(logits),
              _p(loss_out), 1 if accumulate_loss else 0, *gp, ctypes.byref(rec), _p(ws), nb, _stream())
        return logits, grads
    _call("gdm_dcnn_fused", _p(xa), ba, _p(p0), _p(p1), b, int(t), float(ya), float(yb), _p(pack), _p(logits),
          _p(loss_out), 1 if accumulate_loss else 0, 1 if want_grad else 0, *gp, _p(ws), nb, _stream())
    return logits, grads


# ------------------------------------------------------------------------------------------ patch lowering
def im2col(src, *, planar, b, h, w, c, kh, kw, stride, pad, out_dtype):
    _need_gpu(src)
    assert src.is_contiguous()
    oh = (h + 2 * pad - kh) // stride + 1
    ow = (w + 2 * pad - kw) // stride + 1
    cols = torch.empty((b * oh * ow, c * kh * kw), dtype=_TORCH_DT[out_dtype], device=src.device)
    _call("gdm_im2col", _p(src), gdm_dtype(src), 1 if planar else 0, b, h, w, c, kh, kw, stride, pad, oh, ow,
                                 _p(cols), out_dtype, _stream())
    return cols, oh, ow


def col2im(cols, *, b, h, w, c, kh, kw, stride, pad, oh, ow, out_dtype, planar=False, tap_major=False, act=ACT_NONE):
    """tap_major: the columns of ``cols`` are ordered (kh, kw, c) instead of torch's (c, kh, kw); act: fused
    activation on the scattered sum (ACT_NONE / ACT_RELU / ACT_SIGMOID)."""
    assert act in (ACT_NONE, ACT_RELU, ACT_SIGMOID)
    _need_gpu(cols)
    assert cols.is_contiguous() and cols.shape == (b * oh * ow, c * kh * kw)
    shape = (b, c, h, w) if planar else (b, h, w, c)
    dst = torch.empty(shape, dtype=_TORCH_DT[out_dtype], device=cols.device)
    _call("gdm_col2im", _p(cols), gdm_dtype(cols), b, h, w, c, kh, kw, stride, pad, oh, ow, _p(dst),
                                 out_dtype, (1 if planar else 0) | (2 if tap_major else 0) | (act << 4), _stream())
    return dst


def permute_pc(src, b, p, c, out_dtype=None, out=None):
    """(B, P, C) -> (B, C, P) with optional dtype conversion (channels-last <-> channel-major)."""
    _need_gpu(src, out)
    assert src.is_contiguous() and src.numel() == b * p * c
    if out is None:
        out = torch.empty((b, c, p), dtype=src.dtype if out_dtype is None else _TORCH_DT[out_dtype],
                          device=src.device)
    assert out.is_contiguous() and out.numel() == b * p * c
    _call("gdm_permute_pc", _p(src), gdm_dtype(src), b, p, c, _p(out), gdm_dtype(out), _stream())
    return out


def maxpool2_fwd(x, b, h, w, c, want_idx=True):
    """x (B*H*W, C) channels-last -> (out (B*(H//2)*(W//2), C), idx uint8 or None)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.numel() == b * h * w * c
    oh, ow = h // 2, w // 2
    out = torch.empty((b * oh * ow, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((b * oh * ow, c), dtype=torch.uint8, device=x.device) if want_idx else None
    _call("gdm_maxpool2_fwd", _p(x), gdm_dtype(x), b, h, w, c, _p(out), _p(idx), _stream())
    return out, idx


def maxpool2_bwd(dout, idx, b, h, w, c):
    """dout (B*(H//2)*(W//2), C) -> dx (B*H*W, C)."""
    _need_gpu(dout, idx)
    assert dout.is_contiguous() and idx.is_contiguous() and dout.numel() == b * (h // 2) * (w // 2) * c
    dx = torch.empty((b * h * w, c), dtype=dout.dtype, device=dout.device)
    _call("gdm_maxpool2_bwd", _p(dout), gdm_dtype(dout), _p(idx), b, h, w, c, _p(dx), _stream())
    return dx


# ---- DES-matrix prologue kernels (matrix_sim_process.py of both models) -----------------------------------------------
def _des_view(g, s):
    """(B,S,S) / (B,1,S,S) fp32 device tensor -> (tensor, B, sample stride in floats); samples must be dense (S,S)."""
    _need_gpu(g)
    if g.dim() == 4:
        assert g.shape[1] == 1, g.shape
        g = g[:, 0]
    assert g.dim() == 3 and g.shape[1:] == (s, s) and g.dtype == torch.float32, (g.shape, g.dtype)
    if g.stride(2) != 1 or g.stride(1) != s:
        g = g.contiguous()
    return g, g.shape[0], (g.stride(0) if g.shape[0] > 1 else s * s)


def des_scan(g, s, dim, *, threshold=None, note_mod=False, norm_aux=False):
    """Returns dict: thr_mask (B,S) u8 or None, instruments, note_levels (B,dim) i32, zero_mask (B,dim) i64 (bit
    pattern of a u64), aux (B,2,dim) fp32 or None, flags (B) i32 -- all device tensors."""
    g, b, stride = _des_view(g, s)
    dev = g.device
    thr_mask = torch.empty((b, s), dtype=torch.uint8, device=dev) if threshold is not None else None
    inst = torch.empty((b, dim), dtype=torch.int32, device=dev)
    notes = torch.empty((b, dim), dtype=torch.int32, device=dev)
    zmask = torch.empty((b, dim), dtype=torch.int64, device=dev)
    aux = torch.empty((b, 2, dim), dtype=torch.float32, device=dev) if norm_aux else None
    flags = torch.empty(b, dtype=torch.int32, device=dev)
    _call("gdm_des_scan", _p(g), stride, b, s, dim, float(threshold if threshold is not None else 0.0),
          1 if note_mod else 0, 1 if norm_aux else 0, _p(thr_mask), _p(inst), _p(notes), _p(zmask), _p(aux), _p(flags),
          _stream())
    return {"thr_mask": thr_mask, "instruments": inst, "note_levels": notes, "zero_mask": zmask, "aux": aux,
            "flags": flags}


def des_routing(g, s, dim, src_mask, residue_col):
    """src_mask (B,dim) u8, residue_col (B,dim) i32 device tensors -> routing matrices (B,dim,dim) fp64 (device)."""
    g, b, stride = _des_view(g, s)
    _need_gpu(src_mask, residue_col)
    assert src_mask.shape == (b, dim) and src_mask.dtype == torch.uint8 and src_mask.is_contiguous()
    assert residue_col.shape == (b, dim) and residue_col.dtype == torch.int32 and residue_col.is_contiguous()
    out = torch.empty((b, dim, dim), dtype=torch.float64, device=g.device)
    _call("gdm_des_routing", _p(g), stride, b, s, dim, _p(src_mask), _p(residue_col), _p(out), _stream())
    return out


def piano_roll_raster(row_ptr, ev_step, ev_vel, n_files, width):
    """CSR note messages (int32 device tensors) -> (roll, dur) (n_files, 128, width) fp32."""
    _need_gpu(row_ptr, ev_step, ev_vel)
    assert row_ptr.dtype == torch.int32 and row_ptr.numel() == n_files * 128 + 1 and row_ptr.is_contiguous()
    assert ev_step.dtype == torch.int32 and ev_vel.dtype == torch.int32 and ev_step.numel() == ev_vel.numel()
    roll = torch.empty((n_files, 128, width), dtype=torch.float32, device=row_ptr.device)
    dur = torch.empty((n_files, 128, width), dtype=torch.float32, device=row_ptr.device)
    _call("gdm_piano_roll_raster", _p(row_ptr), _p(ev_step), _p(ev_vel), n_files, width, _p(roll), _p(dur), _stream())
    return roll, dur


# ---- mel-spectrogram featuriser kernels (GAN_DES/util.py:37-61) -------------------------------------------------------
def stft_frames(x, hop, n_fft):
    """x (B, L) fp32 -> (B * frames, n_fft) centred, reflect-padded frames; frames = 1 + L // hop."""
    _need_gpu(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    b, l = x.shape
    frames = 1 + l // hop
    out = torch.empty((b * frames, n_fft), dtype=torch.float32, device=x.device)
    _call("gdm_stft_frames", _p(x), b, l, x.stride(0), int(hop), int(n_fft), frames, _p(out), _stream())
    return out, frames


def power_spectrum(c, nfreq, ldp=None):
    """c (rows, 2 * nfreq) = [re | im] -> (rows, ldp) power, zero padded beyond nfreq."""
    _need_gpu(c)
    assert c.dim() == 2 and c.shape[1] == 2 * nfreq and c.dtype == torch.float32 and c.is_contiguous()
    ldp = nfreq if ldp is None else ldp
    p = torch.empty((c.shape[0], ldp), dtype=torch.float32, device=c.device)
    _call("gdm_power_spectrum", _p(c), c.shape[0], int(nfreq), int(ldp), _p(p), _stream())
    return p


def power_to_db(mel, b, frames, top_db=80.0, amin=1e-10):
    """mel (b * frames, n_mels) power -> (b, n_mels, frames) dB with the per-window top_db floor."""
    _need_gpu(mel)
    assert mel.dim() == 2 and mel.shape[0] == b * frames and mel.dtype == torch.float32 and mel.is_contiguous()
    n_mels = mel.shape[1]
    out = torch.empty((b, n_mels, frames), dtype=torch.float32, device=mel.device)
    _call("gdm_power_to_db", _p(mel), int(b), int(frames), int(n_mels), float(-1.0 if top_db is None else top_db),
          float(amin), _p(out), _stream())
    return out
