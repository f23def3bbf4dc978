"""CPU restatement of generate_piano_roll (MMGAN_MIDI_DES/datasets.py:13-70) -- TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  MIDI file -> (piano_roll (128, W) = last note_on velocity per one-second step, durations (128, W),
first ``beats_length`` beat times), with the reference's control flow kept as it is: the step index is the ABSOLUTE
time step although the arrays are only ``end - start`` wide, the first note_on past that width raises inside the
reference's bare ``try`` and ends the event loop, messages at or beyond ``sequence_length`` seconds end it too, the
final slice is ``[:, start:end]`` of the already ``end - start`` wide arrays.

PARITY UNPINNED: the arithmetic lives in two third-party packages that are absent here and from /root/reference --
mido 1.3.2 (message merge, tick -> second conversion; requirements.txt:72) and pretty_midi (``get_beats``) -- and the
reference holds no numeric fixture for this function (its two unit tests check shapes only, datasets.py:126-146).  This
file restates their published algorithms (oracle/midi_events.py; get_beats below) and is checked on the MIDI files the
reference ships (MMGAN_MIDI_DES/adj_sim_outputs/midi/*.mid, copied as DATA fixtures to tests/golden/midi/) against
hand-computable properties only (tests/test_piano_roll.py).
"""
import numpy as np
from . import midi_events as me


def get_beats(fmt, tpb, tracks, start_time=0.0):
    """pretty_midi.PrettyMIDI.get_beats restated for what the reference's files hold.  Tick -> time follows the tempo
    map; beats advance by 60 / bpm from start_time to the end of the last note, bpm = qpm scaled by the time signature's
    denominator (qpm_to_bpm), tempo / time-signature changes handled as in pretty_midi 0.2.10."""
    # absolute-tick event list over all tracks
    ev = []
    for tr in tracks:
        now = 0
        for (d, kind, a, b) in tr:
            now += d
            ev.append((now, kind, a, b))
    ev.sort(key=lambda m: m[0])
    tempo_ticks, tempi_us = [0], [me.DEFAULT_TEMPO]
    for (t, kind, a, b) in ev:
        if kind == "set_tempo":
            if t == 0:
                tempi_us[0] = a
            elif a != tempi_us[-1]:
                tempo_ticks.append(t)
                tempi_us.append(a)
    # tick -> seconds
    def tick_time(tick):
        s, last_tick, scale = 0.0, 0, tempi_us[0] * 1e-6 / tpb
        for tt, us in zip(tempo_ticks[1:], tempi_us[1:]):
            if tick <= tt:
                break
            s += (tt - last_tick) * scale
            last_tick, scale = tt, us * 1e-6 / tpb
        return s + (tick - last_tick) * scale
    tempo_times = np.array([tick_time(t) for t in tempo_ticks])
    tempi = 60.0 / (np.array(tempi_us) * 1e-6)            # quarter notes per minute
    ts = [(tick_time(t), a, b) for (t, kind, a, b) in ev if kind == "time_signature"]
    ts.sort(key=lambda x: x[0])
    note_ends = [tick_time(t) for (t, kind, a, b) in ev if kind == "note_off" or (kind == "note_on" and b == 0)]
    end_time = max(note_ends) if note_ends else 0.0

    def qpm_to_bpm(qpm, num, den):
        if den == 1:
            return qpm / 4.0
        if den == 2:
            return qpm / 2.0
        if den == 4:
            return qpm
        if den in (8, 16, 32):
            if num == 3:
                return 2.0 * qpm if den == 8 else (4.0 * qpm if den == 16 else 8.0 * qpm)
            if num % 3 == 0:
                return {8: 2.0, 16: 4.0, 32: 8.0}[den] * qpm / 3.0
            return {8: 2.0, 16: 4.0, 32: 8.0}[den] * qpm
        return qpm

    beats = [start_time]
    ti = 0
    while ti < len(tempo_times) - 1 and beats[-1] > tempo_times[ti + 1]:
        ti += 1
    si = 0
    while si < len(ts) - 1 and beats[-1] >= ts[si + 1][0]:
        si += 1

    def bpm_now():
        return qpm_to_bpm(tempi[ti], ts[si][1], ts[si][2]) if ts else tempi[ti]

    while beats[-1] < end_time:
        bpm = bpm_now()
        nxt = beats[-1] + 60.0 / bpm
        if ti < len(tempo_times) - 1 and nxt > tempo_times[ti + 1]:
            nxt, remaining = beats[-1], 1.0
            while ti < len(tempo_times) - 1 and nxt + remaining * 60.0 / bpm >= tempo_times[ti + 1]:
                ratio = (tempo_times[ti + 1] - nxt) / (60.0 / bpm)
                nxt += ratio * 60.0 / bpm
                remaining -= ratio
                ti += 1
                bpm = bpm_now()
            nxt += remaining * 60.0 / bpm
        if ts and si < len(ts) - 1:
            nts = ts[si + 1][0]
            if nxt > nts or np.isclose(nxt, nts):
                nxt = nts
                si += 1
        beats.append(nxt)
    return np.array(beats[:-1])


def generate_piano_roll(path, sequence_length=100, beats_length=50, start=0, end=50):
    if sequence_length is None:
        sequence_length = end + 20
    fmt, tpb, tracks = me.load(path)
    width = end - start
    piano_roll = np.zeros((128, width))
    durations = np.zeros((128, width))
    my_time = 0
    note_on_time = np.zeros(128)
    try:
        for (dt, kind, a, b) in me.merged_seconds(fmt, tpb, tracks):
            my_time += dt
            time_step = int(round(my_time))
            if time_step >= sequence_length:
                break
            if kind == "note_on":
                piano_roll[a, time_step] = b           # IndexError past the window: processing stops (bare except upstream)
                note_on_time[a] = time_step
            elif kind == "note_off":
                off = int(round(note_on_time[a]))
                durations[a, off:time_step] = time_step - off
    except IndexError:
        pass
    if end < len(piano_roll):
        piano_roll, durations = piano_roll[:, start:end], durations[:, start:end]
    else:
        piano_roll, durations = piano_roll[:, :end], durations[:, :end]
    beats = get_beats(fmt, tpb, tracks)
    if len(beats) < beats_length:
        beats = np.pad(beats, (0, beats_length - len(beats)))
    elif len(beats) > beats_length:
        beats = beats[:beats_length]
    return piano_roll, durations, beats
