"""MI355X-native GAN training hot path of marja-w/gan-des-midi-music-gen.

Sub-modules mirror the reference's two scripts:
  SIMNN          <- GAN_DES/SIMNN.py            (Generator, Discriminator, SimNN, get_noise, weights_init, train loop)
  network_tests  <- MMGAN_MIDI_DES/network_tests.py (Generator, BeatGenerator, Discriminator, DiscriminatorCNN,
                                                     MultiModalGAN, TestMultiModalGAN.test_training_loop)
  util           <- GAN_DES/util.py             (get_melspectrogram_db_tensor: the mel-dB featuriser)
All arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI declared in include/gdm.h
(libgdm_hip.so, loaded by ``_lib``).  There is no CPU or eager-PyTorch fallback: an op raises if the library is
missing or a tensor is not on a HIP device.
"""
__version__ = "0.1.0"
