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
            v = torch.matmul(F.softplus(sd[f"{prefix}_matrix{i}"].float()), v) + sd[f"{prefix}_bias{i}"].float()
            if i < n_filters:
                v = v + torch.tanh(sd[f"{prefix}_factor{i}"].float()) * torch.tanh(v)
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
        self.last = a
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

    def forward_entropies(self, x):
        """Eval forward()'s rate terms (compressai_coder.py:203-228,352-375): nats per image for y and z."""
        a = self.analyse(x)
        return dict(y=float(gc_entropy(a["y"], a["scales"])),
                    z=float(eb_entropy(self.sd, "latent_node_entropy_coders.z.entropy_bottleneck.", a["z"])))


def gc_entropy(y, scales, bound=0.11, lik_bound=1e-9):
    """GaussianConditional forward's -sum log likelihood / batch (upstream _likelihood, compressai_coder.py:352-375)."""
    v = torch.abs(torch.round(y))
    s = torch.max(scales, torch.tensor([bound]))
    phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
    lik = torch.max(phi((0.5 - v) / s) - phi((-0.5 - v) / s), torch.tensor([lik_bound]))
    return (-torch.log(lik)).sum() / y.shape[0]


def eb_entropy(sd, prefix, z, lik_bound=1e-9, n_filters=4):
    """EntropyBottleneck forward's -sum log likelihood / batch (upstream, compressai_coder.py:203-228)."""
    med = sd[prefix + "quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    zq = torch.round(z - med) + med
    C = z.shape[1]
    v = zq.permute(1, 0, 2, 3).reshape(C, 1, -1)

    def logits(t):
        for i in range(n_filters + 1):
            t = torch.matmul(F.softplus(sd[f"{prefix}_matrix{i}"]), t) + sd[f"{prefix}_bias{i}"]
            if i < n_filters:
                t = t + torch.tanh(sd[f"{prefix}_factor{i}"]) * torch.tanh(t)
        return t

    lower, upper = logits(v - 0.5), logits(v + 0.5)
    sign = -torch.sign(lower + upper)
    lik = torch.max(torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower)), torch.tensor([lik_bound]))
    return (-torch.log(lik)).sum() / z.shape[0]


def psnr(a, b, max_val=1.0):
    """benchmark/metrics/pytorch_distortion.py:12-15 per image."""
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return 20 * math.log10(max_val) - 10 * torch.log10(mse.double())


# ============================================================================================
# Topo-group AR graphs (cfg-3: configs/lossy_latent_graph_topogroup.py:203-244) and the BaSIC
# slimmable graph (cfg-4: configs/presets/lossy_latent_graph_scalable_ar_models.py:73-197)
# ============================================================================================
def run_slimmable(sd, prefix, spec, x, level, out_lists):
    """SlimmableConv2dPGMModel._forward_slimmable (pgm_layers.py:781-845) in eval mode: DynamicConv2d with
    W[:co,:ci] (slimmable_layers.py:157-183) and DynamicGDN with the per-width affine (:258-282).
    spec items: ("conv"|"deconv", stride, k) | ("gdn", inverse) | ("leaky",); out_lists: per conv its channels_list."""
    ci, conv_i = x.shape[1], 0
    for i, item in enumerate(spec):
        key = f"{prefix}.{i}."
        if item[0] in ("conv", "deconv"):
            co = out_lists[conv_i][level]
            conv_i += 1
            w, b = sd[key + "conv.weight"], sd[key + "conv.bias"]
            if item[0] == "conv":
                x = F.conv2d(x, w[:co, :ci].contiguous(), b[:co], stride=item[1], padding=item[2] // 2)
            else:
                x = F.conv_transpose2d(x, w[:ci, :co].contiguous(), b[:co], stride=item[1], padding=item[2] // 2, output_padding=item[1] - 1)
            ci = co
        elif item[0] == "gdn":
            C = x.shape[1]
            rp = lambda v, minimum=0.0: torch.max(v, torch.tensor([(minimum + PEDESTAL) ** 0.5])) ** 2 - torch.tensor([PEDESTAL])
            beta = rp(sd[key + "beta_scales"][level]) * rp(sd[key + "beta"][:C], 1e-6) + rp(sd[key + "beta_biases"][level])
            gamma = rp(sd[key + "gamma_scales"][level]) * rp(sd[key + "gamma"][:C, :C]) + rp(sd[key + "gamma_biases"][level])
            norm = F.conv2d(x * x, gamma.reshape(C, C, 1, 1), beta)
            x = x * (torch.sqrt(norm) if item[1] else torch.rsqrt(norm))
        elif item[0] == "leaky":
            x = F.leaky_relu(x)
        elif item[0] == "relu":
            x = F.relu(x)
    return x


class TopoGroupCodecOracle:
    """Hyperprior transforms + EntropyBottleneck z + topo-group AR Gaussian y-coder (batch 1 streams)."""

    def __init__(self, state_dict, method="checkerboard", channels=192, channel_groups=1, expand_bottleneck=True, use_param_merger=True,
                 prefix=""):
        from .pgm_oracle import TopoGroupGaussianOracle
        self.sd = {k[len(prefix):]: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith(prefix)}
        self.eb = eb_tables(self.sd, "latent_node_entropy_coders.z.entropy_bottleneck.")
        self.z_enc, self.z_dec = _coder(*self.eb[:3])
        ysd = {k[len("latent_node_entropy_coders.y."):]: v for k, v in self.sd.items() if k.startswith("latent_node_entropy_coders.y.")}
        self.y = TopoGroupGaussianOracle(ysd, channels, channel_groups, method, expand_bottleneck, use_param_merger,
                                         context_model=any(k.startswith("topo_group_context_model.") for k in ysd))

    def g_a(self, x): return run_sequential(self.sd, "latent_inference_modules.x_y.model", G_A, x)
    def h_a(self, y): return run_sequential(self.sd, "latent_inference_modules.y_z.model", H_A, y)
    def h_s(self, z): return run_sequential(self.sd, "latent_generative_modules.z_y.model", H_S, z)
    def g_s(self, y): return run_sequential(self.sd, "latent_generative_modules.y_x.model", G_S, y)

    def _z(self, z):
        med = self.eb[3].reshape(1, -1, 1, 1)
        sym = torch.round(z - med)
        C = z.shape[1]
        idx = torch.arange(C, dtype=torch.int32).reshape(1, C, 1, 1).expand_as(sym)
        strings = [self.z_enc.encode_with_indexes(sym[b].int().numpy(), idx[b].numpy()) for b in range(z.shape[0])]
        return write_body(z.shape[-2:], strings), sym + med

    def compress(self, x):
        """A batch is coded the reference's way: one z string per image (write_body), ONE y stream for the whole batch."""
        y = self.g_a(x)
        bz, z_hat = self._z(self.h_a(y))
        prior = self.h_s(z_hat)
        by, sym, idx, buf = self.y.encode(y, prior)
        self.last = dict(y=y, prior=prior, y_sym=sym, y_idx=idx, y_hat=buf)
        return struct.pack("I", len(bz)) + bz + by

    def decompress(self, data):
        (nz,) = struct.unpack("I", data[:4])
        bz, by = data[4:4 + nz], data[4 + nz:]
        z_strings, zshape = read_body(bz)
        C = self.eb[3].numel()
        z_idx = torch.arange(C, dtype=torch.int32).reshape(C, 1, 1).expand(C, *zshape).contiguous().numpy()
        z_sym = torch.stack([torch.from_numpy(self.z_dec.decode_with_indexes(s, z_idx)) for s in z_strings])
        z_hat = z_sym.float() + self.eb[3].reshape(1, -1, 1, 1)
        prior = self.h_s(z_hat)
        y_hat = self.y.decode(by, prior, (prior.shape[0], self.y.C, prior.shape[2], prior.shape[3]))
        return self.g_s(y_hat)

    def forward_entropies(self, x):
        """Eval forward()'s rate terms (latent_graph.py:1168-1178): nats per image for y and z."""
        y = self.g_a(x)
        z = self.h_a(y)
        med = self.eb[3].reshape(1, -1, 1, 1)
        prior = self.h_s(torch.round(z - med) + med)
        return dict(y=float(self.y.forward_entropy(y, prior)),
                    z=float(eb_entropy(self.sd, "latent_node_entropy_coders.z.entropy_bottleneck.", z)))


SL_G_A = [("conv", 2, 5), ("gdn", False), ("conv", 2, 5), ("gdn", False), ("conv", 2, 5), ("gdn", False), ("conv", 2, 5)]
SL_G_S = [("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5), ("gdn", True), ("deconv", 2, 5)]
SL_MS_H_A = [("conv", 1, 3), ("leaky",), ("conv", 2, 5), ("leaky",), ("conv", 2, 5)]
SL_MS_H_S = [("deconv", 2, 5), ("leaky",), ("deconv", 2, 5), ("leaky",), ("conv", 1, 3)]


class BasicCodecOracle(TopoGroupCodecOracle):
    """BaSIC graph: slimmable transforms at explicit width levels (level = argmax of the controller one-hot)."""

    def __init__(self, state_dict, widths, M=192, prefix=""):
        super().__init__(state_dict, method="scanline", channels=M, prefix=prefix)
        self.widths, self.M = list(widths), M
        self.levels = dict(xy=len(widths) - 1, yz=len(widths) - 1, zy=len(widths) - 1, yx=len(widths) - 1)

    def set_levels(self, xy, yz, zy, yx):
        self.levels = dict(xy=xy, yz=yz, zy=zy, yx=yx)

    def g_a(self, x):
        w, n = self.widths, len(self.widths)
        return run_slimmable(self.sd, "latent_inference_modules.x_y.pgm_model", SL_G_A, x, self.levels["xy"], [w, w, w, [self.M] * n])

    def h_a(self, y):
        w, n = self.widths, len(self.widths)
        return run_slimmable(self.sd, "latent_inference_modules.y_z.pgm_model", SL_MS_H_A, y, self.levels["yz"], [w, w, [self.M] * n])

    def h_s(self, z):
        w, n = self.widths, len(self.widths)
        return run_slimmable(self.sd, "latent_generative_modules.z_y.pgm_model", SL_MS_H_S, z, self.levels["zy"],
                             [w, [c * 3 // 2 for c in w], [2 * self.M] * n])

    def g_s(self, y):
        w, n = self.widths, len(self.widths)
        return run_slimmable(self.sd, "latent_generative_modules.y_x.pgm_model", SL_G_S, y, self.levels["yx"], [w, w, w, [3] * n])
