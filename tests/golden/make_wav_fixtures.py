#!/usr/bin/env python3
"""tests/golden/wav/*.wav: audio DATA the reference ships (FluidSynth renderings of simulated / generated MIDI), cut to
what the featuriser tests need -- the whole 2-second simulation.wav and the first 5-second window (the reference's
window_size, GAN_DES/datasets.py:23-36, util.py:103-119) of two longer files.  16-bit stereo PCM is kept as it is, so the
tests go through the reference's own steps (int16 / 32768, channel mean or channel 0).  Run in the build container only
(needs /root/reference)."""
import os
import wave

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SRC = (("MMGAN_MIDI_DES/adj_sim_outputs/midi/simulation.wav", "simulation.wav", None),
       ("MMGAN_MIDI_DES/adj_sim_outputs/wav/generation.wav", "generation_first5s.wav", 5),
       ("GAN_DES/adj_sim_outputs/wav/output_0.wav", "output_0_first5s.wav", 5))

os.makedirs(os.path.join(HERE, "wav"), exist_ok=True)
for rel, name, seconds in SRC:
    with wave.open(os.path.join(REF, rel)) as w:
        params = w.getparams()
        n = w.getnframes() if seconds is None else min(w.getnframes(), seconds * w.getframerate())
        data = w.readframes(n)
    with wave.open(os.path.join(HERE, "wav", name), "wb") as o:
        o.setnchannels(params.nchannels)
        o.setsampwidth(params.sampwidth)
        o.setframerate(params.framerate)
        o.writeframes(data)
    print(name, params.nchannels, params.sampwidth, params.framerate, n)
