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
