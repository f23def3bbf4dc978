"""Shared helpers for the parity tests (summaries identical to tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tensor_summary(t, n_samples=16):
    f = t.detach().double().cpu().flatten()
    rng = np.random.RandomState(f.numel() % (2 ** 31 - 1))
    idx = rng.randint(0, f.numel(), size=n_samples)
    return np.concatenate([[f.sum().item(), f.norm().item()], f[idx].numpy()]).astype(np.float64)


def weight_digest(t):
    return np.frombuffer(hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).digest()[:8],
                         dtype=np.uint64)[0]


def assert_summary_close(got, want, rtol, atol, what=""):
    """want/got = [sum, l2, samples...]; the sum is compared against a scale given by the L2 norm."""
    got, want = np.asarray(got), np.asarray(want)
    l2 = max(abs(want[1]), 1e-30)
    assert abs(got[1] - want[1]) <= rtol * l2 + atol, f"{what}: L2 {got[1]} vs {want[1]}"
    assert abs(got[0] - want[0]) <= 50 * rtol * l2 + 50 * atol, f"{what}: sum {got[0]} vs {want[0]}"
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol * 10, atol=atol + rtol * l2 * 0.05, err_msg=what)


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu().flatten()
    b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def record(name, **kv):
    """Append a measured deviation to gpurun_out/parity_r03.jsonl (when that directory exists): the numbers DESIGN.md
    quotes next to the bounds."""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r03.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **kv}) + "\n")


class _RoundOperand(torch.autograd.Function):
    """value rounded to bf16 on the way forward (an operand the kernel reads in bf16), gradient untouched"""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradient(torch.autograd.Function):
    """identity forward; the gradient that flows back through this point is rounded to bf16 (a gradient map the
    kernels store in bf16 / a gradient operand they read in bf16)"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def round_operand(x):
    return _RoundOperand.apply(x)


def round_gradient(x):
    return _RoundGradient.apply(x)
