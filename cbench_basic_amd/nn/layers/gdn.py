"""GDN parameter holder with CompressAI 1.2.3's re-parametrisation (compressai is an
un-vendored dependency of the reference, requirements.txt:15; in-tree mirror of the forward:
cbench/nn/layers/slimmable_layers.py:270-280, pgm_layers.py:53-66).

    beta_eff  = max(beta,  sqrt(beta_min + 2^-36))^2 - 2^-36
    gamma_eff = max(gamma, sqrt(0        + 2^-36))^2 - 2^-36
    y = x * rsqrt(beta_eff + gamma_eff . x^2)        (inverse: * sqrt)

The module only HOLDS parameters (state-dict compatible with compressai.layers.GDN); the
arithmetic runs fused into the producing conv kernel (csrc/conv.hip epilogue).
"""
import torch
import torch.nn as nn

_REPARAM_OFFSET = 2.0 ** -18


class _NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0):
        super().__init__()
        pedestal = _REPARAM_OFFSET ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = nn.Module()
        self.lower_bound.register_buffer("bound", torch.Tensor([(float(minimum) + pedestal) ** 0.5]))

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = torch.max(x, self.lower_bound.bound)
        return out ** 2 - self.pedestal


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = _NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = _NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def effective(self):
        """(gamma_eff [C,C], beta_eff [C]) float32 on the host, evaluated in the reference's op order."""
        with torch.no_grad():
            b, g = self.beta.detach().float().cpu(), self.gamma.detach().float().cpu()
            bp, bb = self.beta_reparam.pedestal.cpu(), self.beta_reparam.lower_bound.bound.cpu()
            gp, gb = self.gamma_reparam.pedestal.cpu(), self.gamma_reparam.lower_bound.bound.cpu()
            beta = torch.max(b, bb) ** 2 - bp
            gamma = torch.max(g, gb) ** 2 - gp
        return gamma, beta

    def forward(self, x):
        raise RuntimeError("GDN runs fused into the MFMA conv epilogue; call the owning transform model")
