"""CPU oracle (test infrastructure only) for model 2 -- MMGAN_MIDI_DES/network_tests.py.

fp32 PyTorch-CPU restatement written from the reference's behaviour:
  get_noise         network_tests.py:43-44
  weights_init      network_tests.py:47-55   (only ever reaches the generators' Linear layers: 73, 108)
  Generator         network_tests.py:58-90   4x [Linear -> BatchNorm1d(train) -> Sigmoid] -> view(B,-1,adj0,adj1)
  BeatGenerator     network_tests.py:93-123  same blocks, input = cat(noise, beats)
  Discriminator     network_tests.py:126-144 3x [Linear -> LeakyReLU(0.2)] (also on the final logit)
  DiscriminatorCNN  network_tests.py:147-160 conv k4 s2 p1 -> LeakyReLU -> conv k4 s2 p1 -> LeakyReLU -> fc
  MultiModalGAN     network_tests.py:163-206 wrapper; the DES bridge (matrix_to_midi, line 189) is replaced by an
                                             injected ``fake_provider`` because it is out of scope
State-dict keys follow the reference (gen.{i}.0.weight = Linear, gen.{i}.1.* = BatchNorm1d, conv1/conv2/fc.*).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .simnn import _BatchNormState, _Weights, batch_norm_eval, batch_norm_train


def get_noise(n_samples, noise_dim, device="cpu"):
    return torch.randn(n_samples, noise_dim, device=device)


def weights_init(m):
    # network_tests.py:47-55
    kind = getattr(m, "kind", None)
    if kind in ("conv2d", "convT2d"):
        nn.init.normal_(m.weight, mean=0, std=1)
    if kind == "bn2d":
        nn.init.xavier_normal_(m.weight)
        nn.init.constant_(m.bias, 0.0)
    if kind == "linear":
        nn.init.xavier_normal_(m.weight)
        nn.init.constant_(m.bias, 0.0)


class _Sigmoid(nn.Module):
    kind = "sigmoid"


def _gen_block(input_dim, output_dim):
    # nn.Sequential(Linear, BatchNorm1d, Sigmoid): indices 0, 1, 2 give the reference's key names
    return nn.Sequential(
        _Weights("linear", (output_dim, input_dim), output_dim, input_dim),
        _BatchNormState("bn1d", output_dim),
        _Sigmoid(),
    )


def _run_gen(gen, x, training):
    for block in gen:
        lin, bn = block[0], block[1]
        x = F.linear(x, lin.weight, lin.bias)
        x = batch_norm_train(x, bn, (0,)) if training else batch_norm_eval(x, bn)
        x = torch.sigmoid(x)
    return x


class Generator(nn.Module):
    def __init__(self, z_dim=10, im_chan=1, hidden_dim=64, input_dim=None, adj_size=None, device="cpu"):
        super().__init__()
        self.z_dim = z_dim
        self.adj_size = adj_size
        self.device = device
        if input_dim is None:
            input_dim = z_dim
        self.input_tensor_dim = input_dim
        self.gen = nn.Sequential(
            _gen_block(z_dim + input_dim, hidden_dim * 4),
            _gen_block(hidden_dim * 4, hidden_dim * 2),
            _gen_block(hidden_dim * 2, hidden_dim),
            _gen_block(hidden_dim, im_chan * adj_size[0] * adj_size[1]),
        )
        self.gen.apply(weights_init)

    def forward(self, noise, input_tensor=None):
        if input_tensor is None:
            # network_tests.py:83-84: drawn on the CPU generator, then moved
            input_tensor = torch.randn(len(noise), self.input_tensor_dim).to(self.device)
        x = torch.cat((noise, input_tensor), dim=1)
        out = _run_gen(self.gen, x, self.training)
        return out.view(len(noise), -1, self.adj_size[0], self.adj_size[1])


class BeatGenerator(nn.Module):
    def __init__(self, z_dim=10, hidden_dim=64, input_dim=None, output_dim=None, device="cpu"):
        super().__init__()
        self.z_dim = z_dim
        self.output_dim = output_dim
        if input_dim is None:
            input_dim = z_dim
        self.input_tensor_dim = input_dim
        self.device = device
        self.gen = nn.Sequential(
            _gen_block(z_dim + input_dim, hidden_dim * 4),
            _gen_block(hidden_dim * 4, hidden_dim * 2),
            _gen_block(hidden_dim * 2, hidden_dim),
            _gen_block(hidden_dim, output_dim),
        )
        self.gen.apply(weights_init)

    def forward(self, noise, input_tensor=None):
        if input_tensor is None:
            input_tensor = torch.randn(len(noise), self.input_tensor_dim).to(self.device)
        x = torch.cat((noise, input_tensor), dim=1)
        return _run_gen(self.gen, x, self.training)


class _LeakyReLU(nn.Module):
    kind = "leaky_relu"


def _disc_block(input_dim, output_dim):
    return nn.Sequential(_Weights("linear", (output_dim, input_dim), output_dim, input_dim), _LeakyReLU())


class Discriminator(nn.Module):
    """MLP discriminator (never instantiated by MultiModalGAN; API surface, network_tests.py:126-144)."""

    def __init__(self, im_chan=1, hidden_dim=16, roll_size=None, device="cpu"):
        super().__init__()
        self.roll_size = roll_size
        self.device = device
        self.disc = nn.Sequential(
            _disc_block(im_chan * roll_size[0] * roll_size[1] * roll_size[2], hidden_dim),
            _disc_block(hidden_dim, hidden_dim * 2),
            _disc_block(hidden_dim * 2, 1),
        )

    def forward(self, image):
        x = image
        for block in self.disc:
            x = F.leaky_relu(F.linear(x, block[0].weight, block[0].bias), 0.2)
        return x


class DiscriminatorCNN(nn.Module):
    def __init__(self, roll_size=(2, 128, 30), hidden_dim=16):
        super().__init__()
        self.conv1 = _Weights("conv2d", (hidden_dim, roll_size[0], 4, 4), hidden_dim, roll_size[0] * 16)
        self.conv2 = _Weights("conv2d", (hidden_dim * 2, hidden_dim, 4, 4), hidden_dim * 2, hidden_dim * 16)
        self.final_size = hidden_dim * 2 * ((roll_size[1] // 4) * (roll_size[2] // 4))
        self.fc = _Weights("linear", (1, self.final_size), 1, self.final_size)

    def forward(self, image):
        x = F.leaky_relu(F.conv2d(image, self.conv1.weight, self.conv1.bias, stride=2, padding=1), 0.2)
        x = F.leaky_relu(F.conv2d(x, self.conv2.weight, self.conv2.bias, stride=2, padding=1), 0.2)
        x = x.reshape(len(x), -1)
        return F.linear(x, self.fc.weight, self.fc.bias)


class MultiModalGAN(nn.Module):
    """network_tests.py:163-206 with the non-differentiable DES bridge injected.

    ``fake_provider(gen_output1, gen_output2, count) -> (tensor (B,2,128,T), failed_sim_count)`` stands in for
    ``matrix_to_midi`` (network_tests.py:189-193); it receives detached generator outputs exactly like the bridge.
    """

    def __init__(self, z_dim=100, hidden_dim=64, adj_size=(28, 28), roll_size=(2, 128, 50), input_dim=50,
                 output_dim=16, instrument=None, start=30, end=80, device="cpu", fake_provider=None):
        super().__init__()
        self.z_dim = z_dim
        self.generator1 = Generator(z_dim, hidden_dim=hidden_dim, adj_size=adj_size, device=device)
        self.generator2 = BeatGenerator(z_dim, hidden_dim=hidden_dim, input_dim=input_dim, output_dim=output_dim,
                                        device=device)
        self.discriminator = DiscriminatorCNN(roll_size=roll_size)
        self.instrument = instrument
        self.start = start
        self.end = end
        self.adj_size = adj_size
        self.device = device
        self.fake_provider = fake_provider

    def forward(self, noise1, noise2, input_tensor, count, make_dot_png=True, g1_input=None):
        # g1_input: the tensor Generator.forward would otherwise draw itself (network_tests.py:83-84); tests pass it
        # explicitly so that the CPU RNG stream is not part of the comparison.
        gen_output1 = self.generator1(noise1, g1_input)
        gen_output2 = self.generator2(noise2, input_tensor)
        sim_output, failed = self.fake_provider(gen_output1.detach(), gen_output2.detach(), count)
        return self.discriminator(sim_output), failed
