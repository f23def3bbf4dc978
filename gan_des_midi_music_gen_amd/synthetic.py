"""Seeded synthetic stand-ins for the two data sources of the hot path.

The reference feeds its discriminators from (a) mel-dB spectrogram windows of MAESTRO audio
(GAN_DES/datasets.py:17-91 -> (B,128,216) fp32) and (b) pickled piano-roll windows
(MMGAN_MIDI_DES/datasets.py:73-87 -> piano_roll (B,128,50), durations (B,128,50), beats (B,50)), and (c) from the
DES bridge (matrix_to_wav / matrix_to_midi) which is out of scope.  Neither dataset nor bridge can exist on the
benchmark box, so benchmark and parity inputs are drawn here, always on the CPU generator (so that a CPU run and a
GPU run see identical bits) and then moved to the requested device.
"""
import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def spectrogram_batch(batch, hw=(128, 216), seed=1234, device="cpu"):
    """dB-like mel spectrogram windows: clamp(N(-35, 18^2), -80, 30), shape (B, H, W)."""
    g = _gen(seed)
    x = torch.randn(batch, hw[0], hw[1], generator=g) * 18.0 - 35.0
    return x.clamp_(-80.0, 30.0).to(device)


def simnn_inputs(batch, hw=(128, 216), seed=1234, device="cpu", noise_dim=100):
    """real, fake (B,H,W) and generator noise (B,noise_dim,1,1) for one SIMNN iteration."""
    g = _gen(seed)
    real = (torch.randn(batch, hw[0], hw[1], generator=g) * 18.0 - 35.0).clamp_(-80.0, 30.0)
    fake = (torch.randn(batch, hw[0], hw[1], generator=g) * 18.0 - 35.0).clamp_(-80.0, 30.0)
    noise = torch.randn(batch, noise_dim, 1, 1, generator=g)
    return real.to(device), fake.to(device), noise.to(device)


def _roll(batch, t, g):
    vel = (torch.rand(batch, 128, t, generator=g) < 0.10).float() * torch.randint(1, 128, (batch, 128, t),
                                                                                 generator=g).float()
    dur = (torch.rand(batch, 128, t, generator=g) < 0.05).float() * torch.randint(1, 7, (batch, 128, t),
                                                                                 generator=g).float()
    return vel, dur


def mmgan_inputs(batch, t=50, seed=1234, device="cpu", noise_dim=50, beats_len=50):
    """MAESTRO-shaped batch for one MMGAN iteration.

    Returns a dict: piano_roll, durations (B,128,T); beats (B,beats_len) = cumsum(U(0.3,0.8));
    noise1, noise2, g1_in_a, g1_in_b (B,noise_dim); fake_a, fake_b (B,2,128,T) drawn from the same law as the
    real rolls (they stand in for the two DES-bridge outputs of network_tests.py:294 and 312).
    """
    g = _gen(seed)
    out = {}
    out["piano_roll"], out["durations"] = _roll(batch, t, g)
    out["beats"] = torch.cumsum(torch.rand(batch, beats_len, generator=g) * 0.5 + 0.3, dim=1)
    for k in ("noise1", "noise2", "g1_in_a", "g1_in_b"):
        out[k] = torch.randn(batch, noise_dim, generator=g)
    for k in ("fake_a", "fake_b"):
        v, d = _roll(batch, t, g)
        out[k] = torch.stack([v, d], dim=1).contiguous()
    return {k: v.to(device) for k, v in out.items()}
