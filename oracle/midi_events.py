"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Standard MIDI File reader reduced to what generate_piano_roll needs (MMGAN_MIDI_DES/datasets.py:13-70): the merged,
time-ordered message stream with delta times in seconds, as ``for msg in mido.MidiFile(path)`` yields it, and the
tempo / time-signature / note-end data ``pretty_midi.PrettyMIDI(path).get_beats()`` works from.

mido (1.3.2) and pretty_midi are third-party dependencies of the reference that are absent here; this restates their
published behaviour:
  * mido.MidiFile.__iter__: tracks merged by absolute tick (stable sort, track order breaks ties), end_of_track
    messages removed (their delta carried over), delta ticks converted with the tempo in force BEFORE the message
    (tick * tempo * 1e-6 / ticks_per_beat, default tempo 500000), a set_tempo message changing the tempo for what follows;
  * running status, meta events (0xFF type len data), sysex (0xF0 / 0xF7 len data), variable-length quantities.
"""
import struct

DEFAULT_TEMPO = 500000


def _vlq(data, i):
    v = 0
    while True:
        b = data[i]
        i += 1
        v = (v << 7) | (b & 0x7F)
        if not b & 0x80:
            return v, i


_CHANNEL_LEN = {0x8: 2, 0x9: 2, 0xA: 2, 0xB: 2, 0xC: 1, 0xD: 1, 0xE: 2}
_SYSTEM_LEN = {0xF1: 1, 0xF2: 2, 0xF3: 1, 0xF6: 0, 0xF8: 0, 0xFA: 0, 0xFB: 0, 0xFC: 0, 0xFE: 0}


def read_tracks(data):
    """bytes -> (format, ticks_per_beat, [track]); a track is a list of (delta_ticks, kind, a, b):
    kind 'note_on'/'note_off' (a = note, b = velocity), 'set_tempo' (a = microseconds per beat),
    'time_signature' (a = numerator, b = denominator), 'end_of_track', or 'other'."""
    if data[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File (no MThd chunk)")
    hlen, fmt, ntrks, division = struct.unpack(">IHHH", data[4:14])
    if division & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    i = 8 + hlen
    tracks = []
    while i + 8 <= len(data) and len(tracks) < ntrks:
        tag, n = data[i:i + 4], struct.unpack(">I", data[i + 4:i + 8])[0]
        i += 8
        if tag != b"MTrk":
            i += n
            continue
        end, track, status = i + n, [], None
        while i < end:
            delta, i = _vlq(data, i)
            b0 = data[i]
            if b0 == 0xFF:                                   # meta event
                mtype = data[i + 1]
                n2, j = _vlq(data, i + 2)
                body = data[j:j + n2]
                i = j + n2
                if mtype == 0x51 and n2 == 3:
                    track.append((delta, "set_tempo", (body[0] << 16) | (body[1] << 8) | body[2], 0))
                elif mtype == 0x58 and n2 >= 2:
                    track.append((delta, "time_signature", body[0], 2 ** body[1]))
                elif mtype == 0x2F:
                    track.append((delta, "end_of_track", 0, 0))
                else:
                    track.append((delta, "other", 0, 0))
                continue
            if b0 in (0xF0, 0xF7):                           # sysex
                n2, j = _vlq(data, i + 1)
                i = j + n2
                track.append((delta, "other", 0, 0))
                status = None
                continue
            if b0 & 0x80:
                status = b0
                i += 1
            elif status is None:
                raise ValueError("data byte without running status")
            if status >= 0xF0:
                i += _SYSTEM_LEN.get(status, 0)
                track.append((delta, "other", 0, 0))
                continue
            n_data = _CHANNEL_LEN[status >> 4]
            a = data[i]
            b = data[i + 1] if n_data == 2 else 0
            i += n_data
            hi = status >> 4
            if hi == 0x9:
                track.append((delta, "note_on", a, b))       # velocity 0 stays a note_on, as in mido
            elif hi == 0x8:
                track.append((delta, "note_off", a, b))
            else:
                track.append((delta, "other", a, b))
        i = end
        tracks.append(track)
    return fmt, division, tracks


def merged_seconds(fmt, tpb, tracks):
    """The message stream of ``for msg in MidiFile``: list of (delta_seconds, kind, a, b)."""
    if fmt == 2:
        raise TypeError("can't merge tracks in type 2 (asynchronous) file")
    absolute = []
    for tr in tracks:
        now = 0
        for (d, kind, a, b) in tr:
            now += d
            absolute.append((now, kind, a, b))
    absolute.sort(key=lambda m: m[0])                          # stable: track order breaks ties
    out, last, carry, tempo = [], 0, 0, DEFAULT_TEMPO
    for (t, kind, a, b) in absolute:
        delta = t - last
        last = t
        if kind == "end_of_track":
            carry += delta
            continue
        ticks = delta + carry
        carry = 0
        out.append((ticks * (tempo * 1e-6 / tpb) if ticks > 0 else 0, kind, a, b))     # mido.tick2second
        if kind == "set_tempo":
            tempo = a
    out.append((carry * (tempo * 1e-6 / tpb) if carry > 0 else 0, "end_of_track", 0, 0))
    return out


def load(path):
    with open(path, "rb") as f:
        data = f.read()
    fmt, tpb, tracks = read_tracks(data)
    return fmt, tpb, tracks
