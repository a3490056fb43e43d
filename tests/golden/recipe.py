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


def seed_name(name):
    """The name a parameter's generator is seeded by: the EntropyBottleneck's ``_biasN`` is seeded as ``biases.N`` (the
    spelling the first fixtures were drawn with), so the values do not depend on which of the two layouts a class uses."""
    head, _, leaf = name.rpartition(".")
    if "entropy_bottleneck" in head and leaf.startswith("_bias") and leaf[5:].isdigit():
        return f"{head}.biases.{leaf[5:]}"
    return name


def named_seed_weights(module, base_seed, calib=()):
    touched = []
    with torch.no_grad():
        params = dict(module.named_parameters())
        for name, p in params.items():
            g = torch.Generator().manual_seed((zlib.crc32(seed_name(name).encode()) + int(base_seed)) % (1 << 31))
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


# ---- GroupedVariableRateCodec op script (fixture grouped_codec.npz)
GROUPED_OPS = [  # (method, args, kwargs) applied in order to a grouped codec of 4 members; see tests/test_cpu_host.py
    ("num_rate_levels", (), {}), ("num_complex_levels", (), {}), ("num_tasks", (), {}), ("__len__", (), {}),
    ("compress", ("x0",), {}), ("set_rate_level", (2,), {}), ("compress", ("x1",), {}), ("decompress", ("b1",), {}),
    ("set_complex_level", (5,), {}), ("set_complex_level", (3,), {"active_only": True}), ("get_current_complex_metrics", (), {}),
    ("set_rate_level", (3,), {}), ("get_current_complex_metrics", (), {}), ("set_task", (1,), {}), ("set_task", (0,), {"active_only": True}),
    ("update_state", (), {}), ("post_training_process", (), {}), ("forward_estimate_bitlen", ("x2",), {}), ("forward", ("x3",), {}),
    ("set_rate_level", (0,), {}), ("forward", ("x4",), {}),
]
GROUPED_VR_CONFIG = {0: (1, 2), 1: (1, 0), 2: (3, 1), 3: (0, 0), 4: (2, 1), 5: (2, 0)}   # outer level -> (member, member's level)


def run_grouped_ops(make_member, Grouped, vr_config):
    """Shared by the generator (reference classes) and the test (this repository's): returns the call log and the values
    returned by every op.  Members: 0 and 2 have rate levels / complexity levels / tasks, 1 only complexity, 3 none."""
    log = []
    members = [make_member(0, log, rate=3, complex=8, tasks=2), make_member(1, log, complex=5), make_member(2, log, rate=2, complex=8, tasks=3),
               make_member(3, log)]
    g = Grouped(members, **({"codec_vr_level_config": vr_config} if vr_config else {}))
    log.append(("init_active", g.active_codec_idx))
    rets = []
    for name, args, kwargs in GROUPED_OPS:
        if name in ("num_rate_levels", "num_complex_levels", "num_tasks"):
            r = getattr(g, name)
        elif name == "__len__":
            r = len(g)
        elif name == "forward":
            r = g(*args, **kwargs)
        else:
            r = getattr(g, name)(*args, **kwargs)
        rets.append(repr(r))
        log.append(("active", g.active_codec_idx))
    return [repr(e) for e in log], rets
