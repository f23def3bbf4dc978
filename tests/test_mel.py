"""Mel-spectrogram featuriser (first "next" row; reference GAN_DES/util.py:37-61).

PARITY PARTLY PINNED: torchaudio (where the reference's arithmetic lives) is not importable here and the reference
ships no numeric fixture for this function.  Its first and heaviest stage is pinned all the same: torchaudio's
``Spectrogram`` IS ``torch.stft(...).abs().pow(2)`` with a periodic Hann window, and torch.stft is importable -- the
oracle's STFT is compared with it at the reference's geometry (test_oracle_stft_is_torch_stft).  The mel filter bank and
the dB conversion are compared with transformers.audio_utils (an importable, independent implementation documented as
torchaudio-compatible), and so is the whole pipeline assembled from the two libraries
(test_oracle_filter_bank_and_db_are_the_published_ones); torchaudio's own bits remain out of reach.  The GPU tests compare
the HIP path with the oracle on seeded signals.
"""
import numpy as np
import pytest
import torch

from oracle import mel as om

SR = 44100


def _window(seed, n=5 * SR):
    g = np.random.default_rng(seed)
    t = np.arange(n) / SR
    x = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * np.sin(2 * np.pi * 3000.0 * t + 1.0)
    x += 0.02 * g.standard_normal(n)
    env = np.clip(np.sin(2 * np.pi * 0.7 * t + g.uniform(0, 6)) + 0.6, 0.0, 1.0)
    return (x * env).astype(np.float32)


def test_oracle_filter_bank_and_shapes():
    fb = om.melscale_fbanks(1025, 20.0, 8300.0, 128, SR)
    assert fb.shape == (1025, 128) and fb.min() >= 0.0 and fb.max() <= 1.0
    freqs = np.linspace(0, SR // 2, 1025)
    centres = freqs[fb.argmax(0)]
    # (below ~700 Hz the triangles are narrower than a 21.5-Hz FFT bin: neighbouring filters may peak on the same bin)
    assert np.all(np.diff(centres) >= 0) and 20.0 < centres[0] < 80.0 and 7900.0 < centres[-1] < 8300.0
    assert np.all(fb.max(0) > 0)                                          # no empty filter at n_fft = 2048
    # norm=None: neighbouring triangles sum to one between the first and the last filter centre
    m = np.linspace(2595 * np.log10(1 + 20 / 700), 2595 * np.log10(1 + 8300 / 700), 130)
    f_pts = 700 * (10 ** (m / 2595) - 1)
    inner = (freqs >= f_pts[1]) & (freqs <= f_pts[-2])
    np.testing.assert_allclose(fb[inner].sum(1), 1.0, atol=1e-9)
    assert np.all(fb[(freqs < f_pts[0]) | (freqs > f_pts[-1])] == 0.0)
    out = om.get_melspectrogram_db_tensor(_window(0))
    assert out.shape == (128, 216) and out.dtype == np.float32          # the discriminator's (128, 216) input
    assert out.max() - out.min() <= 80.0 + 1e-4                          # top_db floor


@pytest.mark.parametrize("n", [5 * SR, 3 * SR + 123, 4410])
def test_oracle_stft_is_torch_stft(n):
    """torchaudio.transforms.Spectrogram(n_fft=2048, hop_length=hop, power=2) is
    ``torch.stft(x, 2048, hop, 2048, hann_window(2048) [periodic], center=True, pad_mode="reflect", normalized=False,
    onesided=True, return_complex=True).abs().pow(2)`` (torchaudio/functional/functional.py: spectrogram): the stage of
    the featuriser that an importable library pins.  float64 against the oracle's float64 FFT, and torch's float32
    result (what the reference computes in) within float32 rounding of it."""
    x = _window(7, n)
    hop = n // 215                                                        # util.py:43
    want = om.stft_power(x, 2048, hop)
    xt = torch.from_numpy(x)
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        w = torch.hann_window(2048, periodic=True, dtype=dt)
        got = torch.stft(xt.to(dt), n_fft=2048, hop_length=hop, win_length=2048, window=w, center=True,
                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().pow(2)
        assert tuple(got.shape) == want.shape == (1025, 1 + n // hop)
        err = np.abs(got.numpy().astype(np.float64) - want).max() / want.max()
        assert err < tol, (dt, err)
    np.testing.assert_allclose(om.hann_periodic(2048), torch.hann_window(2048, periodic=True, dtype=torch.float64).numpy(),
                               atol=1e-15)


def test_oracle_filter_bank_and_db_are_the_published_ones():
    """The other two stages against an importable independent implementation of the same published algorithm:
    ``transformers.audio_utils.mel_filter_bank(norm=None, mel_scale="htk")`` (documented as torchaudio's
    ``melscale_fbanks``) and ``power_to_db(reference=1, min_value=1e-10, db_range=top_db)`` (= AmplitudeToDB("power",
    top_db)).  Then the whole featuriser, stage by stage from those libraries, against the oracle."""
    au = pytest.importorskip("transformers.audio_utils")
    fb = om.melscale_fbanks(1025, 20.0, 8300.0, 128, SR)
    ref_fb = au.mel_filter_bank(num_frequency_bins=1025, num_mel_filters=128, min_frequency=20.0, max_frequency=8300.0,
                                sampling_rate=SR, norm=None, mel_scale="htk")
    np.testing.assert_allclose(fb, ref_fb, rtol=0, atol=1e-12)
    g = np.random.default_rng(3)
    pw = (np.abs(g.standard_normal((128, 216))) ** 2 * 10.0).astype(np.float32)
    pw[5, 7] = 0.0                                                         # the 1e-10 clamp
    np.testing.assert_allclose(om.amplitude_to_db(pw, 80.0),
                               au.power_to_db(pw.astype(np.float64), reference=1.0, min_value=1e-10, db_range=80.0),
                               atol=2e-5)
    x = _window(11)
    hop = len(x) // 215
    w = torch.hann_window(2048, periodic=True, dtype=torch.float64)
    power = torch.stft(torch.from_numpy(x).double(), n_fft=2048, hop_length=hop, win_length=2048, window=w, center=True,
                       pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().pow(2).numpy()
    want = au.power_to_db(ref_fb.T @ power, reference=1.0, min_value=1e-10, db_range=80.0)[:, :216]
    got = om.get_melspectrogram_db_tensor(x)
    assert got.shape == (128, 216)
    np.testing.assert_allclose(got, want, atol=1e-3)                      # dB; float32 result


def test_oracle_pure_tone_level_and_parseval():
    n = 5 * SR
    t = np.arange(n) / SR
    amp, f0 = 0.5, 1000.0
    x = (amp * np.sin(2 * np.pi * f0 * t)).astype(np.float32)
    hop = n // 215
    p = om.stft_power(x, 2048, hop)
    # Parseval per frame: sum over the one-sided bins (interior doubled) = N * sum((w x)^2)
    win = om.hann_periodic(2048)
    xp = np.pad(x.astype(np.float64), (1024, 1024), mode="reflect")
    f = 100
    seg = xp[f * hop: f * hop + 2048] * win
    lhs = p[0, f] + p[-1, f] + 2.0 * p[1:-1, f].sum()
    np.testing.assert_allclose(lhs, 2048 * np.sum(seg ** 2), rtol=1e-9)
    # a stationary tone: total power in a frame = (amp * sum(w) / 2)^2 spread over the main lobe
    expect = (amp * win.sum() / 2.0) ** 2
    k0 = int(round(f0 * 2048 / SR))
    np.testing.assert_allclose(p[k0 - 3:k0 + 4, f].sum(), expect * 1.5, rtol=0.05)   # Hann: sum of lobe powers = 1.5 peak
    db = om.get_melspectrogram_db_tensor(x)
    fb = om.melscale_fbanks(1025, 20.0, 8300.0, 128, SR)
    band = int(fb[k0].argmax())
    assert int(db[:, 100].argmax()) in (band - 1, band, band + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(0, 5 * SR), (1, 5 * SR), (2, 3 * SR + 123), (3, 4410)])
def test_melspectrogram_matches_oracle(seed, n):
    from gan_des_midi_music_gen_amd import util
    x = _window(seed, n)
    want = om.get_melspectrogram_db_tensor(x)
    got = util.get_melspectrogram_db_tensor(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == want.shape
    # fp32 DFT-by-GEMM vs float64 FFT: bins within 60 dB of the window maximum agree to 0.02 dB, the quiet rest
    # (down to the -80 dB floor, where fp32 round-off of the loud bins leaks in) to 0.5 dB
    loud = want > want.max() - 60.0
    assert np.abs(got - want)[loud].max() < 0.02, np.abs(got - want)[loud].max()
    assert np.abs(got - want).max() < 0.5, np.abs(got - want).max()


@pytest.mark.gpu
def test_melspectrogram_batch_silence_and_errors():
    from gan_des_midi_music_gen_amd import util, ops
    xs = np.stack([_window(s) for s in range(3)] + [np.zeros(5 * SR, np.float32)])
    got = util.get_melspectrogram_db_tensor(torch.from_numpy(xs).cuda()).cpu().numpy()
    assert got.shape == (4, 128, 216)
    for i in range(3):
        want = om.get_melspectrogram_db_tensor(xs[i])
        assert np.abs(got[i] - want)[want > want.max() - 60.0].max() < 0.02
    np.testing.assert_allclose(got[3], -100.0, atol=1e-4)                              # 10 log10(amin) everywhere
    with pytest.raises(ops.GdmError):
        util.get_melspectrogram_db_tensor(torch.zeros(5 * SR))                          # CPU tensor: no fallback
    with pytest.raises(ops.GdmError):
        util.melspectrogram_db_batch(torch.zeros(1, 600).cuda(), hop=100)               # shorter than the reflect pad


def _read_wav(path):
    """What torchaudio.load(path, normalize=True) returns for 16-bit PCM: (channels, frames) float32 in [-1, 1), rate."""
    import wave
    with wave.open(path) as w:
        assert w.getsampwidth() == 2
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
        return (raw.T.astype(np.float32) / 32768.0), w.getframerate()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["simulation.wav", "generation_first5s.wav", "output_0_first5s.wav"])
def test_featuriser_on_the_audio_the_reference_ships(name):
    """The reference's own renderings (tests/golden/wav/, cut by tests/golden/make_wav_fixtures.py) through both of its
    call paths -- get_melspectrogram_db_tensor_from_file (channel mean, whole file, file's sample rate: util.py:89-100)
    and InputSong (channel 0, 5-second windows: GAN_DES/datasets.py:17-52) -- on the device against oracle/mel.py.
    STILL PARITY UNPINNED (torchaudio is absent and the reference holds no spectrogram fixture): this pins the HIP path
    to the restated algorithm on real programme material instead of synthetic tones."""
    import os
    from gan_des_midi_music_gen_amd import util
    wav, sr = _read_wav(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wav", name))
    for x in (wav.mean(axis=0), wav[0][: 5 * sr]):
        x = np.ascontiguousarray(x, dtype=np.float32)
        want = om.get_melspectrogram_db_tensor(x, sr=sr)
        got = util.get_melspectrogram_db_tensor(torch.from_numpy(x).cuda(), sr=sr).cpu().numpy()
        assert got.shape == want.shape == (128, 216)
        loud = want > want.max() - 60.0
        assert np.abs(got - want)[loud].max() < 0.02, np.abs(got - want)[loud].max()
        assert np.abs(got - want).max() < 0.5
