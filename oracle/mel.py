"""CPU restatement of the mel-spectrogram featuriser -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``get_melspectrogram_db_tensor`` (reference GAN_DES/util.py:37-61 = MMGAN_MIDI_DES/util.py): the step that
turns a 5-second mono window into the (128, 216) dB tensor model 1's discriminator consumes.  The arithmetic itself
lives in a third-party dependency that is absent here and from /root/reference: **torchaudio, pinned 2.2.1**
(requirements.txt) -- ``transforms.MelSpectrogram`` (= ``Spectrogram`` + ``MelScale``) and ``transforms.AmplitudeToDB``.
This file restates their published algorithm with the defaults the reference call site leaves in place:

  * Spectrogram: ``torch.stft(n_fft=2048, hop_length=hop, win_length=2048, window=hann_window(2048) (periodic),
    center=True, pad_mode="reflect", normalized=False, onesided=True)``, power 2  ->  (1025, 1 + L // hop)
  * MelScale: ``melscale_fbanks(n_freqs=1025, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk")``,
    triangular filters on the HTK scale  m = 2595 log10(1 + f / 700),  mel = fb^T @ power
  * AmplitudeToDB(stype="power", top_db): 10 log10(clamp(x, 1e-10)) (reference value 1.0), then every value is raised to
    at least (max over the spectrogram) - top_db.

PARITY PINNED TO IMPORTABLE RE-IMPLEMENTATIONS, NOT TO TORCHAUDIO ITSELF: torchaudio cannot be imported here (no wheel,
no network) and the reference holds no numeric fixture for this function.  Every stage is tied to an independent,
importable implementation of the same published algorithm (tests/test_mel.py): the STFT to ``torch.stft`` -- the very
function torchaudio's Spectrogram calls -- (1e-12 in float64), the filter bank to
``transformers.audio_utils.mel_filter_bank(norm=None, mel_scale="htk")`` (identical), the dB conversion to its
``power_to_db`` (float32 rounding), and the whole pipeline to the three chained.  What stays unverified is that
torchaudio 2.2.1's own code equals that published algorithm bit for bit.
numpy float64 FFT internally, float32 result.
"""
import numpy as np


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_power(x, n_fft, hop):
    """x (L,) -> power spectrogram (n_fft // 2 + 1, 1 + L // hop), centred frames with reflect padding."""
    x = np.asarray(x, dtype=np.float64)
    pad = n_fft // 2
    if x.shape[0] <= pad:
        raise ValueError(f"reflect padding needs more than {pad} samples, got {x.shape[0]}")
    xp = np.pad(x, (pad, pad), mode="reflect")
    frames = 1 + x.shape[0] // hop
    win = hann_periodic(n_fft)
    idx = np.arange(frames)[:, None] * hop + np.arange(n_fft)[None, :]
    spec = np.fft.rfft(xp[idx] * win[None, :], axis=1)
    return (spec.real ** 2 + spec.imag ** 2).T


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """(n_freqs, n_mels) triangular HTK filter bank, norm=None."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def amplitude_to_db(x, top_db=80.0, amin=1e-10):
    db = 10.0 * np.log10(np.maximum(x, amin))
    if top_db is not None:
        db = np.maximum(db, db.max() - top_db)
    return db


def get_melspectrogram_db_tensor(waveform, sr=44100, n_fft=2048, hop_length=512, n_mels=128, fmin=20, fmax=8300,
                                 top_db=80, mel_length=216):
    """waveform (L,) -> (n_mels, 1 + L' // hop) float32 with hop = L // (mel_length - 1) and L' = min(L, mel_length * hop)
    (the reference overrides ``hop_length`` and crops exactly like this, util.py:40-44)."""
    waveform = np.asarray(waveform)
    hop = waveform.shape[0] // (mel_length - 1)
    waveform = waveform[: mel_length * hop]
    power = stft_power(waveform, n_fft, hop)
    fb = melscale_fbanks(n_fft // 2 + 1, float(fmin), float(fmax), n_mels, sr)
    mel = fb.T @ power
    return amplitude_to_db(mel, top_db).astype(np.float32)
