"""CPU oracle (test infrastructure only) for model 1 -- GAN_DES/SIMNN.py.

fp32 PyTorch-CPU restatement written from the reference's behaviour:
  get_noise      GAN_DES/SIMNN.py:37-46
  weights_init   GAN_DES/SIMNN.py:49-59
  Generator      GAN_DES/SIMNN.py:62-112   (4x ConvTranspose2d, 3x BatchNorm2d(train)+ReLU, sigmoid)
  Discriminator  GAN_DES/SIMNN.py:115-142  (conv k2 p1 -> ReLU -> pool2 -> conv k3 p1 -> ReLU -> pool2 ->
                                            fc1 -> ReLU -> fc2 -> sigmoid)
Parameter containers keep the reference's state_dict key names (conv1.weight, batch_norm1.running_mean, fc1.bias, ...)
and draw their initial values with the same RNG call sequence as the reference's constructors so that
``torch.manual_seed(s); Discriminator()`` reproduces the reference's weights bit for bit (checked against
checksums in tests/golden/simnn_*.npz).

``input_hw`` is the build's keyword-only extension (SURVEY.md section 8 geometry note); the default (128, 216)
is the reference geometry (fc1.in_features = 32*32*54).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def get_noise(n_samples, noise_dim, device="cpu"):
    # SIMNN.py:46
    return torch.randn(n_samples, noise_dim, 1, 1, device=device)


class _Weights(nn.Module):
    """Bare parameter holder; ``kind`` is what the reference's isinstance() checks look at."""

    def __init__(self, kind, weight_shape, bias_len, fan_in):
        super().__init__()
        self.kind = kind
        self.weight = nn.Parameter(torch.empty(weight_shape))
        # torch's default reset_parameters(): kaiming_uniform_(a=sqrt(5)) for the weight, U(+-1/sqrt(fan_in)) for
        # the bias; the bound is formed with the same double-precision expression so the draws match bit for bit.
        gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
        w_bound = math.sqrt(3.0) * (gain / math.sqrt(fan_in))
        with torch.no_grad():
            self.weight.uniform_(-w_bound, w_bound)
        if bias_len:
            self.bias = nn.Parameter(torch.empty(bias_len))
            b_bound = 1 / math.sqrt(fan_in)
            with torch.no_grad():
                self.bias.uniform_(-b_bound, b_bound)
        else:
            self.register_parameter("bias", None)


class _BatchNormState(nn.Module):
    def __init__(self, kind, channels):
        super().__init__()
        self.kind = kind
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


def batch_norm_train(x, bn, reduce_dims):
    """Training-mode batch norm with running-stat update (momentum 0.1, unbiased running var).

    Restates aten::native_batch_norm(training=True) as used at SIMNN.py:105-108 and network_tests.py:78,113.
    """
    n = 1
    for d in reduce_dims:
        n *= x.shape[d]
    mean = x.mean(dim=reduce_dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=reduce_dims, keepdim=True)  # biased, used for normalisation
    shape = [1] * x.dim()
    shape[1] = -1
    y = (x - mean) / torch.sqrt(var + BN_EPS) * bn.weight.view(shape) + bn.bias.view(shape)
    with torch.no_grad():
        unbiased = var.flatten() * (n / max(n - 1, 1))
        bn.running_mean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.flatten())
        bn.running_var.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased)
        bn.num_batches_tracked += 1
    return y


def batch_norm_eval(x, bn):
    shape = [1] * x.dim()
    shape[1] = -1
    return (x - bn.running_mean.view(shape)) / torch.sqrt(bn.running_var.view(shape) + BN_EPS) * bn.weight.view(
        shape
    ) + bn.bias.view(shape)


def weights_init(m):
    # SIMNN.py:49-59 -- conv / conv-transpose weights ~ N(0, 0.02); BatchNorm2d weight ~ N(0, 0.02) (sic), bias 0.
    kind = getattr(m, "kind", None)
    if kind in ("conv2d", "convT2d"):
        nn.init.normal_(m.weight, mean=0.0, std=0.02)
    if kind == "bn2d":
        nn.init.normal_(m.weight, mean=0.0, std=0.02)
        nn.init.constant_(m.bias, val=0)


class Generator(nn.Module):
    def __init__(self, no_of_channels=1, noise_dim=100, gen_dim=32):
        super().__init__()
        g = gen_dim
        # ConvTranspose2d weight is (in, out, kh, kw); torch computes fan_in from dim 1 (out) * kh * kw.
        self.conv1 = _Weights("convT2d", (noise_dim, g * 4, 4, 4), 0, g * 4 * 16)
        self.conv2 = _Weights("convT2d", (g * 4, g * 2, 4, 4), 0, g * 2 * 16)
        self.conv3 = _Weights("convT2d", (g * 2, g, 4, 4), 0, g * 16)
        self.conv4 = _Weights("convT2d", (g, no_of_channels, 5, 5), 0, no_of_channels * 25)
        self.batch_norm1 = _BatchNormState("bn2d", g * 4)
        self.batch_norm2 = _BatchNormState("bn2d", g * 2)
        self.batch_norm3 = _BatchNormState("bn2d", g)
        # SIMNN.py:89-95 (_initialize_weights): modules() order = self, conv1..4, batch_norm1..3
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            nn.init.normal_(m.weight, 0.0, 0.02)
        for m in (self.batch_norm1, self.batch_norm2, self.batch_norm3):
            nn.init.normal_(m.weight, 1.0, 0.02)
            nn.init.constant_(m.bias, 0)

    def forward(self, input):
        bn = batch_norm_train if self.training else (lambda x, b, _d: batch_norm_eval(x, b))
        x = F.conv_transpose2d(input, self.conv1.weight, None, stride=1, padding=0)
        x = torch.relu(bn(x, self.batch_norm1, (0, 2, 3)))
        x = F.conv_transpose2d(x, self.conv2.weight, None, stride=2, padding=1)
        x = torch.relu(bn(x, self.batch_norm2, (0, 2, 3)))
        x = F.conv_transpose2d(x, self.conv3.weight, None, stride=2, padding=1)
        x = torch.relu(bn(x, self.batch_norm3, (0, 2, 3)))
        x = F.conv_transpose2d(x, self.conv4.weight, None, stride=1, padding=0)
        return torch.sigmoid(x)


def disc_feature_hw(input_hw):
    h, w = input_hw
    h1, w1 = (h + 1) // 2, (w + 1) // 2  # conv k2 p1 -> (h+1, w+1); pool 2 floor
    return h1 // 2, w1 // 2  # conv k3 p1 keeps; pool 2 floor


class Discriminator(nn.Module):
    def __init__(self, no_of_channels=1, disc_dim=32, *, input_hw=(128, 216)):
        super().__init__()
        self.input_hw = tuple(input_hw)
        fh, fw = disc_feature_hw(self.input_hw)
        self.flat = 32 * fh * fw
        self.conv1 = _Weights("conv2d", (16, 1, 2, 2), 16, 1 * 4)
        self.conv2 = _Weights("conv2d", (32, 16, 3, 3), 32, 16 * 9)
        self.fc1 = _Weights("linear", (128, self.flat), 128, self.flat)
        self.fc2 = _Weights("linear", (1, 128), 1, 128)

    def forward(self, input):
        x = input.unsqueeze(1)
        x = F.max_pool2d(torch.relu(F.conv2d(x, self.conv1.weight, self.conv1.bias, stride=1, padding=1)), 2, 2)
        x = F.max_pool2d(torch.relu(F.conv2d(x, self.conv2.weight, self.conv2.bias, stride=1, padding=1)), 2, 2)
        x = x.reshape(-1, self.flat)
        x = torch.relu(F.linear(x, self.fc1.weight, self.fc1.bias))
        return torch.sigmoid(F.linear(x, self.fc2.weight, self.fc2.bias))
