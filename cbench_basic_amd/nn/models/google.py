"""g_a / g_s / h_a / h_s transforms -- same classes, constructor arguments and state-dict keys as
cbench/nn/models/google.py:25-143 -- executed by the hand-written MFMA kernels.

``self.model`` is an nn.Sequential of parameter holders (nn.Conv2d / nn.ConvTranspose2d / GDN /
activation markers) so checkpoints converted with tools/compressai_checkpoint_to_cbench.py
(:139-172, keys ``...latent_inference_modules.x_y.model.N.weight``) load unchanged.  forward()
compiles the sequence into fused layer plans (conv + bias + GDN/ReLU in one launch) on first
use and replays them; it needs the input on the MI355X and has no CPU path.
"""
import torch
import torch.nn as nn

from ..layers.gdn import GDN
from .. import kernels as K


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


def compile_sequential(seq, cin_active=None, cout_actives=None):
    """[conv|deconv] [GDN|ReLU|LeakyReLU]? ... -> list of fused ConvPlans."""
    layers = list(seq)
    plans, i = [], 0
    while i < len(layers):
        m = layers[i]
        if not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            raise TypeError(f"transform must start a fused group with a convolution, got {type(m).__name__}")
        act, gamma, beta = K.ACT_NONE, None, None
        if i + 1 < len(layers):
            n = layers[i + 1]
            if isinstance(n, GDN):
                act = K.ACT_IGDN if n.inverse else K.ACT_GDN
                gamma, beta = n.effective()
                i += 1
            elif isinstance(n, nn.ReLU):
                act = K.ACT_RELU
                i += 1
            elif isinstance(n, nn.LeakyReLU):
                if abs(n.negative_slope - 0.01) > 1e-12:
                    raise NotImplementedError("LeakyReLU slope other than 0.01")
                act = K.ACT_LEAKY_RELU
                i += 1
        tr = isinstance(m, nn.ConvTranspose2d)
        if m.groups != 1 or m.dilation != (1, 1) or m.kernel_size[0] != m.kernel_size[1]:
            raise NotImplementedError("only dense square convolutions are on the hot path")
        plans.append(K.ConvPlan(m.weight.detach(), m.bias.detach() if m.bias is not None else None, m.stride[0], m.padding[0],
                                m.output_padding[0] if tr else 0, tr, act, gamma, beta))
        i += 1
    return plans


class BasicHyperpriorModule(nn.Module):
    """google.py:10-23 -- optional per-channel gains around the transform."""

    def __init__(self):
        super().__init__()
        self._plans = None
        self._plan_key = None

    def _key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def plans(self):
        key = self._key()
        if self._plans is None or key != self._plan_key:
            self._plans = compile_sequential(self.model)
            self._plan_key = key
        return self._plans

    def flops(self, batch, h, w):
        total = 0
        for p in self.plans():
            total += p.flops(batch, h, w)
            h, w = p.out_hw(h, w)
        return total

    def _forward(self, x):
        for p in self.plans():
            x = p(x)
        return x

    def forward(self, input, in_channel_gains=None, out_channel_gains=None, **kwargs):
        if in_channel_gains is not None:
            input = input * in_channel_gains.reshape(1, -1, 1, 1)
        output = self._forward(input)
        if out_channel_gains is not None:
            output = output * out_channel_gains.reshape(1, -1, 1, 1)
        return output


class HyperpriorAnalysisModel(BasicHyperpriorModule):  # google.py:25-43
    def __init__(self, N, M, in_channels=3, **kwargs):
        super().__init__()
        self.in_channels, self.N, self.M = in_channels, N, M
        self.model = nn.Sequential(conv(in_channels, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))


class HyperpriorSynthesisModel(BasicHyperpriorModule):  # google.py:46-64
    def __init__(self, N, M, out_channels=3, **kwargs):
        super().__init__()
        self.out_channels, self.N, self.M = out_channels, N, M
        self.model = nn.Sequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                   deconv(N, N), GDN(N, inverse=True), deconv(N, out_channels))


class HyperpriorHyperAnalysisModel(BasicHyperpriorModule):  # google.py:67-82
    def __init__(self, N, M, **kwargs):
        super().__init__()
        self.N, self.M = N, M
        self.model = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.ReLU(inplace=True), conv(N, N),
                                   nn.ReLU(inplace=True), conv(N, N))


class HyperpriorHyperSynthesisModel(BasicHyperpriorModule):  # google.py:85-101
    def __init__(self, N, M, **kwargs):
        super().__init__()
        self.N, self.M = N, M
        self.model = nn.Sequential(deconv(N, N), nn.ReLU(inplace=True), deconv(N, N), nn.ReLU(inplace=True),
                                   conv(N, M, stride=1, kernel_size=3), nn.ReLU(inplace=True))


class MeanScaleHyperpriorHyperAnalysisModel(BasicHyperpriorModule):  # google.py:104-119
    def __init__(self, N, M, **kwargs):
        super().__init__()
        self.N, self.M = N, M
        self.model = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True),
                                   conv(N, N, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                   conv(N, N, stride=2, kernel_size=5))


class MeanScaleHyperpriorHyperSynthesisModel(BasicHyperpriorModule):  # google.py:122-137
    def __init__(self, N, M, **kwargs):
        super().__init__()
        self.N, self.M = N, M
        self.model = nn.Sequential(deconv(N, M, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                   deconv(M, M * 3 // 2, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                   conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))
