"""Seeded-weight RECIPE shared by tests/golden/make_golden.py (which applies it to the REFERENCE's modules) and by the
tests (which apply it to this repository's modules and to the CPU oracle).  Pure torch; imports nothing from the
reference.  Every tensor is drawn from its own generator seeded by crc32(parameter name) + base seed, so the recipe
depends on parameter NAMES (the state_dict contract), not on their order:

  * GDN parameters (beta / gamma and the per-width affine of DynamicGDN) and the EntropyBottleneck's matrices, factors
    and quantiles keep their constructor values (deterministic constants);
  * EntropyBottleneck biases (uniform(-.5, .5) from the global RNG in the constructor) are redrawn: rand(shape) - 0.5;
  * dim >= 2:  randn(shape) / sqrt(fan_in)      (fan_in = prod(shape[1:]))
  * dim == 1:  randn(shape) * 0.05
  * then ``calib`` = [(name, mul, add_to_odd_channels)]:  p *= mul;  p[1::2] += add   (spreads the latents over the scale
    table and keeps x-hat O(1)).
"""
import zlib

import numpy as np
import torch

_KEEP = (".beta", ".gamma", "_reparam", "scale_beta", "scale_gamma", "bias_beta", "bias_gamma", "beta_scales", "beta_biases",
         "gamma_scales", "gamma_biases")


def named_seed_weights(module, base_seed, calib=()):
    touched = []
    with torch.no_grad():
        params = dict(module.named_parameters())
        for name, p in params.items():
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + int(base_seed)) % (1 << 31))
            if "entropy_bottleneck" in name:
                if ".biases." in name or "._biases" in name or "_bias" in name.rsplit(".", 1)[-1]:
                    p.copy_((torch.rand(p.shape, generator=g) - 0.5).to(p.device))
                    touched.append((name, ",".join(str(d) for d in p.shape)))
                continue
            if any(t in name for t in _KEEP) or not p.is_floating_point():
                continue
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_((torch.randn(p.shape, generator=g) * (1.0 / fan_in ** 0.5)).to(p.device))
            else:
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
            touched.append((name, ",".join(str(d) for d in p.shape)))
        for name, mul, add in calib:
            params[str(name)].mul_(float(mul))
            if float(add):
                params[str(name)][1::2] += float(add)
    return touched


def recipe_input(seed, shape):
    return torch.rand(*[int(s) for s in shape], generator=torch.Generator().manual_seed(int(seed)))
