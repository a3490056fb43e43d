"""CPU ORACLE of the codec graph -- test infrastructure only (tests/, smoke(), bench cpu_baseline).

Restates, with plain PyTorch-CPU fp32 ops + the C rANS oracle, what the reference computes for the
hyperprior latent graph:
    encode  latent_graph.py:1232-1264  (x -> g_a -> y -> h_a -> z ; z coder ; h_s ; y coder ; merge)
    decode  latent_graph.py:1266-1295
    g_a/g_s/h_a/h_s   nn/models/google.py:25-101 (compressai conv/deconv/GDN, upstream semantics)
    z coder           compressai_coder.py:230-245  (EntropyBottleneck, upstream semantics)
    y coder           compressai_coder.py:377-393  (GaussianConditional, upstream semantics)
    framing           compressai_coder.py:63-84, utils/bytes_ops.py:19-51

Parity status: the rANS layer / pmf quantisation are PINNED (see rans64_oracle.c); the
CompressAI-defined arithmetic (EB/GC tables, GDN re-parametrisation) is "parity unpinned":
compressai==1.2.3 is not vendored in /root/reference and no reference test touches it
(SURVEY 8c), so this file restates its published algorithm.

It works from a ``state_dict`` (same keys as the reference: ``latent_inference_modules.x_y.model.0.weight`` ...)
so the product and the oracle can be fed identical weights.
"""
import io
import math
import struct

import numpy as np
import torch
import torch.nn.functional as F

from . import rans_oracle as ro

PEDESTAL = (2.0 ** -18) ** 2


def scale_table(lo=0.11, hi=256, levels=64):
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def _reparam(x, minimum):
    bound = (minimum + PEDESTAL) ** 0.5
    return torch.max(x, torch.tensor([bound])) ** 2 - torch.tensor([PEDESTAL])


def gdn(x, gamma_raw, beta_raw, inverse):
    C = x.shape[1]
    beta = _reparam(beta_raw, 1e-6)
    gamma = _reparam(gamma_raw, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x * x, gamma, beta)
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


def run_sequential(sd, prefix, spec, x):
    """spec: list of ("conv"|"deconv", stride, k) | ("gdn", inverse) | ("relu",) | ("leaky",)."""
    for i, item in enumerate(spec):
        key = f"{prefix}.{i}."
        if item[0] == "conv":
            x = F.conv2d(x, sd[key + "weight"], sd[key + "bias"], stride=item[1], padding=item[2] // 2)
        elif item[0] == "deconv":
            x = F.conv_transpose2d(x, sd[key + "weight"], sd[key + "bias"], stride=item[1], padding=item[2] // 2,
                                   output_padding=item[1] - 1)
        elif item[0] == "gdn":
            x = gdn(x, sd[key + "gamma"], sd[key + "beta"], item[1])
        elif item[0] == "relu":
            x = F.relu(x)
        elif item[0] == "leaky":
            x = F.leaky_relu(x)
    return x


G_A = [("conv", 2, 5), ("gdn", False), ("conv", 2, 5), ("gdn", False), ("conv", 2, 5), ("gdn", False), ("conv", 2, 5)]
G_S = [("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5)]
H_A = [("conv", 1, 3), ("relu",), ("conv", 2, 5), ("relu",), ("conv", 2, 5)]
H_S = [("deconv", 2, 5), ("relu",), ("deconv", 2, 5), ("relu",), ("conv", 1, 3), ("relu",)]
MS_H_A = [("conv", 1, 3), ("leaky",), ("conv", 2, 5), ("leaky",), ("conv", 2, 5)]
MS_H_S = [("deconv", 2, 5), ("leaky",), ("deconv", 2, 5), ("leaky",), ("conv", 1, 3)]


def pmf_rows_to_cdf(pmf, tail, lengths, max_length):
    cdf = np.zeros((len(lengths), max_length + 2), dtype=np.int32)
    for i in range(len(lengths)):
        prob = np.concatenate([pmf[i, : lengths[i]], tail[i]]).astype(np.float32)
        q = ro.pmf_to_quantized_cdf(prob.tolist(), 16)
        cdf[i, : len(q)] = q
    return cdf


def eb_tables(sd, prefix, n_filters=4):
    """EntropyBottleneck.update() upstream semantics."""
    q = sd[prefix + "quantiles"].float()
    med = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
    start = med - minima
    length = maxima + minima + 1
    L = int(length.max())
    samples = torch.arange(L)[None, :] + start[:, None, None]

    def logits(v):
        for i in range(n_filters + 1):
            v = torch.matmul(F.softplus(sd[f"{prefix}matrices.{i}"].float()), v) + sd[f"{prefix}biases.{i}"].float()
            if i < n_filters:
                v = v + torch.tanh(sd[f"{prefix}factors.{i}"].float()) * torch.tanh(v)
        return v

    lower, upper = logits(samples - 0.5), logits(samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = pmf_rows_to_cdf(pmf.numpy(), tail.numpy(), length.numpy(), L)
    return cdf, (length + 2).numpy().astype(np.int32), (-minima).numpy().astype(np.int32), med


def gc_tables(table, tail_mass=1e-9):
    """GaussianConditional.update() upstream semantics."""
    from scipy.stats import norm
    mult = -norm.ppf(tail_mass / 2)
    center = torch.ceil(table * mult).int()
    length = 2 * center + 1
    L = int(length.max())
    samples = torch.abs(torch.arange(L).int() - center[:, None]).float()
    sc = table.unsqueeze(1).float()
    phi = lambda v: 0.5 * torch.erfc(-(2 ** -0.5) * v)
    upper, lower = phi((0.5 - samples) / sc), phi((-0.5 - samples) / sc)
    cdf = pmf_rows_to_cdf((upper - lower).numpy(), (2 * lower[:, :1]).numpy(), length.numpy(), L)
    return cdf, (length + 2).numpy().astype(np.int32), (-center).numpy().astype(np.int32)


def gc_indexes(scales, table, bound=0.11):
    s = torch.max(scales, torch.tensor([bound]))
    idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


def write_body(shape, strings):
    out = [struct.pack(">3I", shape[0], shape[1], len(strings))]
    for s in strings:
        out.append(struct.pack(">I", len(s)))
        out.append(s)
    return b"".join(out)


def read_body(data):
    h, w, n = struct.unpack(">3I", data[:12])
    cur, out = 12, []
    for _ in range(n):
        (L,) = struct.unpack(">I", data[cur:cur + 4])
        out.append(data[cur + 4:cur + 4 + L])
        cur += 4 + L
    return out, (h, w)


def _coder(cdf, sizes, offsets):
    e, d = ro.Rans64Encoder(16, True, 4), ro.Rans64Decoder(16, True, 4)
    e.init_cdf_params(cdf, sizes, offsets)
    d.init_cdf_params(cdf, sizes, offsets)
    return e, d


class HyperpriorOracle:
    """Plain hyperprior graph (configs/lossy_graph_scalable_exp_hp.py:182-215) on the CPU."""

    def __init__(self, state_dict, prefix=""):
        self.sd = {k[len(prefix):]: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith(prefix)}
        self.table = scale_table()
        self.eb = eb_tables(self.sd, "latent_node_entropy_coders.z.entropy_bottleneck.")
        self.gc = gc_tables(self.table)
        self.z_enc, self.z_dec = _coder(*self.eb[:3])
        self.y_enc, self.y_dec = _coder(*self.gc)

    def g_a(self, x): return run_sequential(self.sd, "latent_inference_modules.x_y.model", G_A, x)
    def h_a(self, y): return run_sequential(self.sd, "latent_inference_modules.y_z.model", H_A, y)
    def h_s(self, z): return run_sequential(self.sd, "latent_generative_modules.z_y.model", H_S, z)
    def g_s(self, y): return run_sequential(self.sd, "latent_generative_modules.y_x.model", G_S, y)

    def analyse(self, x):
        """Integer symbols / indexes the entropy stage sees (for symbol-level parity)."""
        y = self.g_a(x)
        z = self.h_a(y)
        med = self.eb[3].reshape(1, -1, 1, 1)
        z_sym = torch.round(z - med)
        z_hat = z_sym + med
        scales = self.h_s(z_hat)[..., : y.shape[-2], : y.shape[-1]]
        return dict(y=y, z=z, z_sym=z_sym.int(), z_hat=z_hat, scales=scales, y_idx=gc_indexes(scales, self.table),
                    y_sym=torch.round(y).int())

    def compress(self, x):
        a = self.analyse(x)
        B, C = a["z"].shape[:2]
        z_idx = torch.arange(C, dtype=torch.int32).reshape(1, C, 1, 1).expand_as(a["z_sym"])
        z_strings = [self.z_enc.encode_with_indexes(a["z_sym"][b].numpy(), z_idx[b].numpy()) for b in range(B)]
        y_strings = [self.y_enc.encode_with_indexes(a["y_sym"][b].numpy(), a["y_idx"][b].numpy()) for b in range(B)]
        bz = write_body(a["z"].shape[-2:], z_strings)
        by = write_body(a["y"].shape[-2:], y_strings)
        return struct.pack("I", len(bz)) + bz + by  # merge_bytes(num_segments=2)

    def decompress(self, data):
        (nz,) = struct.unpack("I", data[:4])
        bz, by = data[4:4 + nz], data[4 + nz:]
        z_strings, zshape = read_body(bz)
        C = self.eb[3].numel()
        z_idx = torch.arange(C, dtype=torch.int32).reshape(C, 1, 1).expand(C, *zshape).contiguous().numpy()
        z_sym = torch.stack([torch.from_numpy(self.z_dec.decode_with_indexes(s, z_idx)) for s in z_strings])
        z_hat = z_sym.float() + self.eb[3].reshape(1, -1, 1, 1)
        y_strings, yshape = read_body(by)
        scales = self.h_s(z_hat)[..., : yshape[0], : yshape[1]]
        y_idx = gc_indexes(scales, self.table)
        y_hat = torch.stack([torch.from_numpy(self.y_dec.decode_with_indexes(s, y_idx[b].numpy())) for b, s in enumerate(y_strings)]).float()
        return self.g_s(y_hat)


def psnr(a, b, max_val=1.0):
    """benchmark/metrics/pytorch_distortion.py:12-15 per image."""
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return 20 * math.log10(max_val) - 10 * torch.log10(mse.double())
