"""CPU oracle for the GAN training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch fp32 CPU restatement (PyTorch CPU tensor ops, no
nn layers' forward code, own Adam/StepLR/BCE formulas) of the arithmetic on the
hot path of marja-w/gan-des-midi-music-gen:

  * GAN_DES/SIMNN.py:37-142, 256-259, 276-334        (model 1: Generator, Discriminator, step)
  * MMGAN_MIDI_DES/network_tests.py:43-206, 248-329  (model 2: Generator, BeatGenerator,
                                                       Discriminator, DiscriminatorCNN,
                                                       MultiModalGAN, step, StepLR)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / reported baseline -- the product package
(``gan_des_midi_music_gen_amd``) never imports it and has no CPU fallback.

Parity status: the reference ships no numeric known-answer test for this path (SURVEY.md
section 8c), so the oracle is pinned by golden vectors captured from the reference's own classes
imported in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and by
the reference's committed checkpoints' key/shape manifests.  ``tests/test_oracle_golden.py``
checks every function here against those vectors.
"""
