"""Width-switchable (slimmable) conv / GDN parameter holders -- cbench/nn/layers/slimmable_layers.py:71-282.

DynamicConv2d keeps ONE maximal weight tensor and runs W[:co, :ci] (weight slicing, :142-170);
DynamicGDN applies a per-width affine to the re-parametrised (beta, gamma) (:270-274):
    beta  = reparam(beta_scales[idx])  * reparam_b(beta[:C])      + reparam(beta_biases[idx])
    gamma = reparam(gamma_scales[idx]) * reparam_g(gamma[:C, :C]) + reparam(gamma_biases[idx])
On the MI355X a width is a *plan*: the active slice is packed once per (layer, width) into the MFMA
kernel's layout (csrc/conv.hip, cin_active / cout_active) and cached.
"""
from typing import Sequence

import torch
import torch.nn as nn

from .gdn import _NonNegativeParametrizer


class DynamicConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=5, stride=2, padding=None, dilation=1, bias=True, transposed=False,
                 conv_compability=False, **kwargs):
        super().__init__()
        self.max_in_channels = max(in_channels) if isinstance(in_channels, Sequence) else in_channels
        self.channels_list = list(out_channels) if isinstance(out_channels, Sequence) else [out_channels]
        self.max_out_channels = max(self.channels_list)
        self.kernel_size, self.stride, self.transposed = kernel_size, stride, transposed
        self.padding = kernel_size // 2 if padding is None else padding
        if dilation != 1 or conv_compability:
            raise NotImplementedError("dilation / conv_compability variants")
        if transposed:
            self.conv = nn.ConvTranspose2d(self.max_in_channels, self.max_out_channels, kernel_size, stride=stride,
                                           output_padding=stride - 1, bias=bias)
        else:
            self.conv = nn.Conv2d(self.max_in_channels, self.max_out_channels, kernel_size, stride=stride, bias=bias)

    def out_channels_at(self, level):
        return self.channels_list[level] if len(self.channels_list) > 1 else self.channels_list[0]


class DynamicGDN(nn.Module):
    def __init__(self, channels_list, inverse=False, beta_min=1e-6, gamma_init=0.1, **kwargs):
        super().__init__()
        self.channels_list = list(channels_list)
        C = max(self.channels_list)
        self.inverse = bool(inverse)
        self.beta_reparam = _NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(C)))
        self.gamma_reparam = _NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(C)))
        n = len(self.channels_list)
        self.gamma_scales = nn.Parameter(torch.ones(n))
        self.gamma_scales_reparam = _NonNegativeParametrizer()
        self.gamma_biases = nn.Parameter(torch.zeros(n))
        self.gamma_biases_reparam = _NonNegativeParametrizer()
        self.beta_scales = nn.Parameter(torch.ones(n))
        self.beta_scales_reparam = _NonNegativeParametrizer()
        self.beta_biases = nn.Parameter(torch.zeros(n))
        self.beta_biases_reparam = _NonNegativeParametrizer()

    def effective(self, idx):
        """(gamma_eff [C,C], beta_eff [C]) of width index idx, float32 host, reference op order."""
        C = self.channels_list[idx]
        f = lambda t: t.detach().float().cpu()
        rp = lambda m, x: torch.max(x, f(m.lower_bound.bound)) ** 2 - f(m.pedestal)
        with torch.no_grad():
            beta = rp(self.beta_scales_reparam, f(self.beta_scales)[idx]) * rp(self.beta_reparam, f(self.beta)[:C]) \
                + rp(self.beta_biases_reparam, f(self.beta_biases)[idx])
            gamma = rp(self.gamma_scales_reparam, f(self.gamma_scales)[idx]) * rp(self.gamma_reparam, f(self.gamma)[:C, :C]) \
                + rp(self.gamma_biases_reparam, f(self.gamma_biases)[idx])
        return gamma.contiguous(), beta.contiguous()
