"""Drop-in surface of the piano-roll data path (MMGAN_MIDI_DES/datasets.py) on MI355X.

    generate_piano_roll(midi_input, sequence_length=100, beats_length=50, start=0, end=50)      datasets.py:13-70
    MaestroDatasetMidi(root_dir, sequence_length=100, beats_length=50, device='cpu')            datasets.py:103-123

``generate_piano_roll`` keeps the reference's signature and return value (numpy ``piano_roll (128, W)``, ``durations
(128, W)``, ``beats (beats_length,)``); ``generate_piano_rolls`` is the batched form the training data path wants: a list
of files in, the (F, 128, W) fp32 planes as DEVICE tensors out (one kernel launch for all files, nothing copied back).

What runs where: a Standard MIDI File is a byte stream with running status and variable-length quantities -- parsing is
sequential host work (``read_midi``); merging tracks, tick -> second conversion, the one-second step index and the
point where the reference's event loop stops are vectorised numpy on the host (a few thousand messages per file); the
raster itself -- last-write-wins scatter of velocities, range fill of durations, per (file, note) row -- is the device
kernel (``ops.piano_roll_raster``, csrc/piano_roll.hip).

mido and pretty_midi are not importable here; their behaviour is restated (see oracle/piano_roll.py for the statement
of what is and is not pinned: PARITY UNPINNED).  The reference's control flow is kept as it is, including: the step
index is absolute although the planes are only ``end - start`` wide (a note_on beyond the width ends the event loop:
IndexError inside the reference's bare ``try``), messages at ``sequence_length`` seconds or later end it as well, and
the final slice is ``[:, start:end]`` of the already ``end - start`` wide planes (empty for start >= end - start).
"""
import glob
import os
import struct

import numpy as np
import torch

from . import ops

DEFAULT_TEMPO = 500000
_K_OTHER, _K_ON, _K_OFF, _K_TEMPO, _K_TSIG, _K_EOT = 0, 1, 2, 3, 4, 5
_DATA_BYTES = {0x8: 2, 0x9: 2, 0xA: 2, 0xB: 2, 0xC: 1, 0xD: 1, 0xE: 2}
_SYS_BYTES = {0xF1: 1, 0xF2: 2, 0xF3: 1}


class MidiData:
    """Messages of all tracks as parallel arrays: absolute tick, track index, kind, two data values."""

    def __init__(self, fmt, ticks_per_beat, tick, track, kind, a, b):
        self.format, self.ticks_per_beat = fmt, ticks_per_beat
        self.tick, self.track, self.kind, self.a, self.b = tick, track, kind, a, b


def read_midi(path_or_bytes):
    """Parse a Standard MIDI File (format 0/1; metrical time division) into a MidiData."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    if len(data) < 14 or data[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    hlen, fmt, ntrks, division = struct.unpack(">IHHH", data[4:14])
    if division & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    pos = 8 + hlen
    tick, track, kind, av, bv = [], [], [], [], []
    n_tr = 0
    while pos + 8 <= len(data) and n_tr < ntrks:
        tag = data[pos:pos + 4]
        size = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        pos += 8
        end = min(pos + size, len(data))
        if tag != b"MTrk":
            pos = end
            continue
        now, status = 0, 0
        while pos < end:
            delta = 0
            while True:                                       # variable-length quantity
                byte = data[pos]
                pos += 1
                delta = (delta << 7) | (byte & 0x7F)
                if byte < 0x80:
                    break
            now += delta
            lead = data[pos]
            k, a, b = _K_OTHER, 0, 0
            if lead == 0xFF:                                  # meta event: type, length, body
                mtype = data[pos + 1]
                pos += 2
                n = 0
                while True:
                    byte = data[pos]
                    pos += 1
                    n = (n << 7) | (byte & 0x7F)
                    if byte < 0x80:
                        break
                body = data[pos:pos + n]
                pos += n
                if mtype == 0x51 and n == 3:
                    k, a = _K_TEMPO, int.from_bytes(body, "big")
                elif mtype == 0x58 and n >= 2:
                    k, a, b = _K_TSIG, body[0], 1 << body[1]
                elif mtype == 0x2F:
                    k = _K_EOT
            elif lead in (0xF0, 0xF7):                        # system exclusive: length, body
                pos += 1
                n = 0
                while True:
                    byte = data[pos]
                    pos += 1
                    n = (n << 7) | (byte & 0x7F)
                    if byte < 0x80:
                        break
                pos += n
                status = 0
            else:
                if lead & 0x80:
                    status = lead
                    pos += 1
                elif not status:
                    raise ValueError("MIDI data byte without a running status")
                if status >= 0xF0:
                    pos += _SYS_BYTES.get(status, 0)
                else:
                    nd = _DATA_BYTES[status >> 4]
                    a = data[pos]
                    b = data[pos + 1] if nd == 2 else 0
                    pos += nd
                    if status >> 4 == 0x9:
                        k = _K_ON                             # velocity 0 stays a note_on (as in mido)
                    elif status >> 4 == 0x8:
                        k = _K_OFF
            tick.append(now)
            track.append(n_tr)
            kind.append(k)
            av.append(a)
            bv.append(b)
        pos = end
        n_tr += 1
    return MidiData(fmt, division, np.asarray(tick, dtype=np.int64), np.asarray(track, dtype=np.int32),
                    np.asarray(kind, dtype=np.int8), np.asarray(av, dtype=np.int64), np.asarray(bv, dtype=np.int64))


def message_seconds(md):
    """The stream ``for msg in mido.MidiFile(...)`` yields, as arrays: (delta seconds, kind, a, b) per message, tracks
    merged by absolute tick (stable: track order breaks ties), end_of_track messages dropped with their delta carried
    to the next message, each delta converted with the tempo in force before the message."""
    if md.format == 2:
        raise TypeError("can't merge tracks in type 2 (asynchronous) file")
    order = np.argsort(md.tick, kind="stable")                 # arrays are track-major already
    tick, kind, a, b = md.tick[order], md.kind[order], md.a[order], md.b[order]
    keep = kind != _K_EOT
    tick_k, kind_k, a_k, b_k = tick[keep], kind[keep], a[keep], b[keep]
    # dropping end_of_track and carrying its delta == deltas between the KEPT messages' absolute ticks
    dticks = np.diff(tick_k, prepend=0)
    tempo = np.full(len(tick_k), DEFAULT_TEMPO, dtype=np.int64)
    is_t = np.flatnonzero(kind_k == _K_TEMPO)
    for j, idx in enumerate(is_t):                             # tempo applies to the messages AFTER the set_tempo
        nxt = is_t[j + 1] + 1 if j + 1 < len(is_t) else len(tick_k)
        tempo[idx + 1:nxt] = a_k[idx]
    secs = np.where(dticks > 0, dticks * (tempo * 1e-6 / md.ticks_per_beat), 0.0)     # mido.tick2second's association
    return secs, kind_k, a_k, b_k


def _row_events(md, sequence_length, width):
    """Note messages the reference's loop processes, grouped by note: (row_ptr (129,), step, vel) as int32 arrays."""
    secs, kind, a, b = message_seconds(md)
    step = np.rint(np.cumsum(secs)).astype(np.int64)           # my_time += msg.time; int(round(my_time)) (half to even)
    stop = len(step)
    late = np.flatnonzero(step >= sequence_length)
    if len(late):
        stop = late[0]                                         # `break`
    wide = np.flatnonzero((kind == _K_ON) & (step >= width))
    if len(wide):
        stop = min(stop, wide[0])                              # piano_roll[note, step] raises: the bare except ends the loop
    sel = np.flatnonzero((kind[:stop] == _K_ON) | (kind[:stop] == _K_OFF))
    notes = a[sel]
    by_note = np.argsort(notes, kind="stable")                 # a note's messages stay in file order
    sel = sel[by_note]
    row_ptr = np.zeros(129, dtype=np.int32)
    np.cumsum(np.bincount(notes, minlength=128)[:128], out=row_ptr[1:])
    vel = np.where(kind[sel] == _K_ON, b[sel], -1).astype(np.int32)
    return row_ptr, step[sel].astype(np.int32), vel


def _qpm_to_bpm(qpm, num, den):
    """pretty_midi.utilities.qpm_to_bpm."""
    if den in (1, 2, 4):
        return qpm * den / 4.0
    if den in (8, 16, 32):
        scale = den / 4.0
        if num == 3:
            return scale * qpm
        if num % 3 == 0:
            return scale * qpm / 3.0
        return scale * qpm
    return qpm


def get_beats(md, start_time=0.0):
    """Beat times as pretty_midi.PrettyMIDI.get_beats computes them (restated): 60 / bpm apart from start_time to the end
    of the last note, bpm from the tempo map and the time signature's denominator, re-anchored at tempo and time-signature
    changes."""
    order = np.argsort(md.tick, kind="stable")
    tick, kind, a, b = md.tick[order], md.kind[order], md.a[order], md.b[order]
    t_ticks, t_us = [0], [DEFAULT_TEMPO]
    for tk, us in zip(tick[kind == _K_TEMPO], a[kind == _K_TEMPO]):
        if tk == 0:
            t_us[0] = int(us)
        elif int(us) != t_us[-1]:
            t_ticks.append(int(tk))
            t_us.append(int(us))
    scales = np.asarray(t_us, dtype=np.float64) * 1e-6 / md.ticks_per_beat
    seg_start = np.concatenate([[0.0], np.cumsum(np.diff(t_ticks) * scales[:-1])])

    def tick_time(tk):
        seg = np.searchsorted(t_ticks, tk, side="left") - 1
        seg = np.clip(seg, 0, len(t_ticks) - 1)
        return seg_start[seg] + (tk - np.asarray(t_ticks)[seg]) * scales[seg]

    tempo_times = seg_start
    tempi = 60.0 / (np.asarray(t_us, dtype=np.float64) * 1e-6)
    ts_mask = kind == _K_TSIG
    ts = sorted(zip(tick_time(tick[ts_mask]).tolist(), a[ts_mask].tolist(), b[ts_mask].tolist()))
    ends = tick[(kind == _K_OFF) | ((kind == _K_ON) & (b == 0))]
    end_time = float(tick_time(ends).max()) if len(ends) else 0.0
    beats = [start_time]
    ti = si = 0
    while ti < len(tempo_times) - 1 and beats[-1] > tempo_times[ti + 1]:
        ti += 1
    while si < len(ts) - 1 and beats[-1] >= ts[si + 1][0]:
        si += 1
    while beats[-1] < end_time:
        bpm = _qpm_to_bpm(tempi[ti], ts[si][1], ts[si][2]) if ts else tempi[ti]
        nxt = beats[-1] + 60.0 / bpm
        if ti < len(tempo_times) - 1 and nxt > tempo_times[ti + 1]:
            nxt, left = beats[-1], 1.0
            while ti < len(tempo_times) - 1 and nxt + left * 60.0 / bpm >= tempo_times[ti + 1]:
                part = (tempo_times[ti + 1] - nxt) / (60.0 / bpm)
                nxt += part * 60.0 / bpm
                left -= part
                ti += 1
                bpm = _qpm_to_bpm(tempi[ti], ts[si][1], ts[si][2]) if ts else tempi[ti]
            nxt += left * 60.0 / bpm
        if ts and si < len(ts) - 1 and (nxt > ts[si + 1][0] or np.isclose(nxt, ts[si + 1][0])):
            nxt = ts[si + 1][0]
            si += 1
        beats.append(nxt)
    return np.asarray(beats[:-1])


def _fit_beats(beats, beats_length):
    if len(beats) < beats_length:
        return np.pad(beats, (0, beats_length - len(beats)))
    return beats[:beats_length]


def generate_piano_rolls(midi_inputs, sequence_length=100, beats_length=50, start=0, end=50, device="cuda"):
    """Batched generate_piano_roll: list of paths (or bytes) -> (piano_roll, durations (F,128,W) fp32, beats
    (F,beats_length) fp32) on ``device``; W = what the reference's final slice leaves."""
    if sequence_length is None:
        sequence_length = end + 20
    width = end - start
    if width <= 0:
        raise ValueError("negative dimensions are not allowed")       # np.zeros((128, end - start)) upstream
    ptrs, steps, vels, beats = [np.zeros(1, dtype=np.int32)], [], [], []
    total = 0
    for item in midi_inputs:
        md = item if isinstance(item, MidiData) else read_midi(item)
        rp, st, ve = _row_events(md, sequence_length, width)
        ptrs.append(rp[1:] + total)
        total += int(rp[-1])
        steps.append(st)
        vels.append(ve)
        beats.append(_fit_beats(get_beats(md), beats_length))
    dev = torch.device(device)
    row_ptr = torch.from_numpy(np.concatenate(ptrs)).to(dev)
    ev_step = torch.from_numpy(np.concatenate(steps) if steps else np.zeros(0, np.int32)).to(dev)
    ev_vel = torch.from_numpy(np.concatenate(vels) if vels else np.zeros(0, np.int32)).to(dev)
    roll, dur = ops.piano_roll_raster(row_ptr, ev_step, ev_vel, len(beats), width)
    # `if end < len(piano_roll)` compares with 128 rows; both branches slice the (128, end - start) planes
    sl = slice(start, end) if end < 128 else slice(0, end)
    return roll[:, :, sl], dur[:, :, sl], torch.from_numpy(np.stack(beats)).float().to(dev)


def generate_piano_roll(midi_input, sequence_length=100, beats_length=50, start=0, end=50, device="cuda"):
    """Reference signature; returns numpy (piano_roll, durations, beats) like upstream (float64 planes)."""
    if not isinstance(midi_input, (str, os.PathLike, bytes, bytearray, MidiData)):
        raise ValueError("midi_input must be a file path or a mido.MidiFile object")
    roll, dur, beats = generate_piano_rolls([midi_input], sequence_length, beats_length, start, end, device)
    return (roll[0].double().cpu().numpy(), dur[0].double().cpu().numpy(),
            _fit_beats(get_beats(midi_input if isinstance(midi_input, MidiData) else read_midi(midi_input)),
                       beats_length))


class MaestroDatasetMidi(torch.utils.data.Dataset):
    """Raw-MIDI dataset of the reference (datasets.py:103-123): item = (piano_roll, durations (128,W), beats) fp32
    tensors on ``device``; ``pattern`` replaces the hard-wired Windows glob."""

    def __init__(self, root_dir, sequence_length=100, beats_length=50, device="cuda", pattern="**/*.mid*"):
        self.root_dir, self.sequence_length, self.beats_length, self.device = root_dir, sequence_length, beats_length, device
        self.file_list = sorted(glob.glob(os.path.join(root_dir, pattern), recursive=True))

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        roll, dur, beats = generate_piano_rolls([self.file_list[idx]], self.sequence_length, self.beats_length,
                                                device=self.device)
        return roll[0], dur[0], beats[0]
