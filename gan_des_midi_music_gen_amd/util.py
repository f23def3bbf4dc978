"""Mel-spectrogram featuriser on the device: the drop-in for ``get_melspectrogram_db_tensor`` of the reference's
``GAN_DES/util.py:37-61`` (= ``MMGAN_MIDI_DES/util.py``), which turns a mono window into the (128, 216) dB tensor that
model 1's discriminator consumes (first "next" row).

torchaudio's ``MelSpectrogram`` + ``AmplitudeToDB`` are restated as (all fp32):

    frames (gdm_stft_frames: centred, reflect padded)  x  [w*cos | w*sin] (2048 x 2050, Hann window folded in)
      -> gdm_gemm (exact-fp32 MFMA)  -> re^2 + im^2 (gdm_power_spectrum)  x  HTK filter bank (1025 x 128)
      -> gdm_gemm  -> 10 log10(max(., 1e-10)), floored at (window max - top_db)  (gdm_power_to_db)

The DFT is a GEMM on purpose: 1.8 GFLOP per window on matrix cores is cheaper to get right than a hand-written FFT and
is still thousands of windows per second; it is data preparation, not part of the training iteration.
There is no CPU path (the constant matrices are built on the host once per geometry and cached).
"""
import math

import torch

from . import ops
from .ops import F32

_CONST = {}


def _dft_matrix(n_fft, device):
    """(n_fft, 2 * (n_fft // 2 + 1)) = [w[n] cos(2 pi k n / N) | w[n] sin(2 pi k n / N)], periodic Hann window w."""
    key = ("dft", n_fft, str(device))
    if key not in _CONST:
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
        win = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / n_fft)
        # reduce k*n modulo N in integers first: the angle stays exact for large products
        kn = (torch.outer(n.long(), k.long()) % n_fft).to(torch.float64)
        ang = 2.0 * math.pi * kn / n_fft
        m = torch.cat([torch.cos(ang), torch.sin(ang)], dim=1) * win[:, None]
        _CONST[key] = m.to(torch.float32).to(device).contiguous()
    return _CONST[key]


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") restated: (n_freqs, n_mels) float64."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def _mel_matrix(n_fft, sr, n_mels, fmin, fmax, ldp, device):
    key = ("mel", n_fft, sr, n_mels, float(fmin), float(fmax), ldp, str(device))
    if key not in _CONST:
        nfreq = n_fft // 2 + 1
        fb = torch.zeros((ldp, n_mels), dtype=torch.float64)
        fb[:nfreq] = melscale_fbanks(nfreq, float(fmin), float(fmax), n_mels, sr)
        _CONST[key] = fb.to(torch.float32).to(device).contiguous()
    return _CONST[key]


def melspectrogram_db_batch(waveforms, sr=44100, n_fft=2048, hop=None, n_mels=128, fmin=20, fmax=8300, top_db=80):
    """waveforms (B, L) fp32 on the device -> (B, n_mels, 1 + L // hop) dB (one top_db floor per window)."""
    if not waveforms.is_cuda:
        raise ops.GdmError("melspectrogram_db_batch runs on a HIP device only (no CPU fallback)")
    x = waveforms if waveforms.dtype == torch.float32 else waveforms.float()
    if x.stride(1) != 1:
        x = x.contiguous()
    b = x.shape[0]
    nfreq = n_fft // 2 + 1
    ldp = (nfreq + 3) // 4 * 4                       # K of the mel GEMM padded to whole 16-byte chunks
    frames_m, frames = ops.stft_frames(x, hop, n_fft)
    spec = ops.gemm(frames_m, _dft_matrix(n_fft, x.device), compute=F32)                 # (B*frames, 2*nfreq)
    power = ops.power_spectrum(spec, nfreq, ldp)
    mel = ops.gemm(power, _mel_matrix(n_fft, sr, n_mels, fmin, fmax, ldp, x.device), compute=F32)
    return ops.power_to_db(mel, b, frames, top_db=top_db)


def get_melspectrogram_db_tensor(waveform, sr=44100, n_fft=2048, hop_length=512, n_mels=128, fmin=20, fmax=8300,
                                 top_db=80, mel_length=216):
    """Same signature and result as the reference (util.py:37-61): waveform (L,) -> (n_mels, frames) dB tensor.
    Like there, ``hop_length`` is overridden by ``len(waveform) // (mel_length - 1)`` and the input is cropped to
    ``mel_length * hop`` samples.  A (B, L) batch is accepted too and returns (B, n_mels, frames)."""
    single = waveform.dim() == 1
    x = waveform.unsqueeze(0) if single else waveform
    hop = x.shape[1] // (mel_length - 1)
    x = x[:, : mel_length * hop]
    out = melspectrogram_db_batch(x, sr=sr, n_fft=n_fft, hop=hop, n_mels=n_mels, fmin=fmin, fmax=fmax, top_db=top_db)
    return out[0] if single else out
