"""z / y latent coders of the plain hyperprior graph -- API of
cbench/modules/prior_model/prior_coder/compressai_coder.py:87-397 (CompressAIEntropyBottleneckPriorCoder,
CompressAISlimmableEntropyBottleneckPriorCoder, CompressAIGaussianConditionalCoder).

The reference delegates the arithmetic to CompressAI 1.2.3 (``EntropyBottleneck``,
``GaussianConditional``, ``compressai.ans``), which is NOT vendored in the reference tree
(requirements.txt:15).  Its published algorithm is restated here:
  * table construction (``update()``) runs once, on the host, in float32 with the upstream
    operation order, and the integer CDFs are uploaded to HBM;
  * the per-image work -- quantise, table index, rANS -- is hand-written HIP
    (csrc/entropy.hip, csrc/rans.hip), one rANS stream per image, bypass always on,
    16-bit precision (csrc/rans/rans_interface.cpp:50-53 is the in-tree fork of compressai.ans).
Framing is the reference's ``write_body``/``read_body`` (compressai_coder.py:63-84), big-endian.
"""
import math
import struct
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....base import HotPathModule
from ....nn import kernels as K
from .... import ans as _ans

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """compressai_coder.py:28-30."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def write_body(shape, out_strings: List[List[bytes]], segments=1) -> bytes:
    """compressai_coder.py:75-84: >III (H, W, n) then per string >I length(s) and payload(s)."""
    parts = [struct.pack(">3I", shape[0], shape[1], len(out_strings))]
    for s in out_strings:
        assert len(s) == segments
        parts.append(struct.pack(">%dI" % segments, *[len(seg) for seg in s]))
        parts.extend(s)
    return b"".join(parts)


class PendingBody:
    """A coder's byte string whose GPU work is enqueued but not yet synchronised (``encode(..., lazy=True)``):
    the caller keeps launching kernels and calls ``result()`` when it needs the bytes."""

    def __init__(self, tables, handle, shape_hw):
        self._tables, self._handle, self._shape = tables, handle, tuple(int(v) for v in shape_hw)

    def _end(self):
        if self._handle is not None:
            self._host, self._off = self._tables.encode_batch_end(self._handle)  # the one host synchronisation
            self._handle = None
        return self._host, self._off

    def result(self) -> bytes:
        host, off = self._end()
        return K.frame_streams(host, off, self._shape)

    def nbytes(self) -> int:
        return K.frame_streams_size(self._end()[1])

    def write_into(self, address: int, capacity: int) -> int:
        """Frame the streams directly into caller memory (the codec's final bytes object): one copy instead of three."""
        host, off = self._end()
        return K.frame_streams_into(host, off, self._shape, address, capacity)


def read_body(data: bytes, segments=1):
    """compressai_coder.py:63-72."""
    h, w, n = struct.unpack(">3I", data[:12])
    cur, out = 12, []
    for _ in range(n):
        lens = struct.unpack(">%dI" % segments, data[cur:cur + 4 * segments])
        cur += 4 * segments
        batch = []
        for L in lens:
            batch.append(data[cur:cur + L])
            cur += L
        out.append(batch)
    return out, (h, w)


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length, precision=16):
    """EntropyModel._pmf_to_cdf (upstream): per row quantise [pmf[:len], tail] to a 2^16 CDF."""
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, : pmf_length[i]], tail_mass[i]]).astype(np.float32)
        q = _ans.pmf_to_quantized_cdf(prob.tolist(), precision)
        cdf[i, : len(q)] = q
    return cdf


class EntropyBottleneck(nn.Module):
    """Parameter holder + table builder of the factorised prior (Balle et al. 2018).

    Parameter names: ``_matrixN / _biasN / _factorN / quantiles`` -- the layout the reference's own checkpoint converter
    writes under ``latent_node_entropy_coders.z.entropy_bottleneck`` (tools/compressai_checkpoint_to_cbench.py:16-25,
    139-152: "nn.ParameterList to nn.Parameters") and the registration order of the pinned compressai 1.2.3
    (requirements.txt:15).  ``state_dict()`` emits these names; ``load_state_dict`` also accepts the ParameterList spelling
    of later compressai releases (``matrices.N / biases.N / factors.N`` and the zoo files' ``_matrices.N`` ...)."""

    _LIST_NAMES = (("matrices.", "_matrix"), ("biases.", "_bias"), ("factors.", "_factor"),
                   ("_matrices.", "_matrix"), ("_biases.", "_bias"), ("_factors.", "_factor"))

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3)):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.likelihood_lower_bound = nn.Module()   # LowerBound(likelihood_bound) of EntropyModel (upstream)
        self.likelihood_lower_bound.register_buffer("bound", torch.Tensor([1e-9]))
        # EntropyModel's registered buffers (upstream): empty until update(); resized on load like
        # compressai.models.utils.update_registered_buffers does
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    @property
    def matrices(self):
        return [getattr(self, f"_matrix{i:d}") for i in range(len(self.filters) + 1)]

    @property
    def biases(self):
        return [getattr(self, f"_bias{i:d}") for i in range(len(self.filters) + 1)]

    @property
    def factors(self):
        return [getattr(self, f"_factor{i:d}") for i in range(len(self.filters))]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for old, new in self._LIST_NAMES:                            # ParameterList spellings -> _matrixN / _biasN / _factorN
            for key in [k for k in state_dict if k.startswith(prefix + old) and k[len(prefix + old):].isdigit()]:
                state_dict[prefix + new + key[len(prefix + old):]] = state_dict.pop(key)
        for name in ("_offset", "_quantized_cdf", "_cdf_length"):   # table buffers take the checkpoint's size
            v = state_dict.get(prefix + name)
            if v is not None and v.shape != getattr(self, name).shape:
                setattr(self, name, torch.zeros(v.shape, dtype=torch.int32, device=getattr(self, name).device))
        # weights without their tables (a strict=False load of a file that carries no buffers): tables built from the OLD weights
        # must not survive, or the next update(force=False) keeps them and z is coded with another model than the one loaded
        loads_weights = any(k.startswith(prefix) and k[len(prefix):].lstrip("_").startswith(("matrix", "bias", "factor", "quantiles"))
                            for k in state_dict)
        if loads_weights and prefix + "_offset" not in state_dict:
            self.reset_tables()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def reset_tables(self):
        """Forget the coding tables (the next ``update()`` rebuilds them): to be called by whatever changes the weights after a
        first ``update_state()`` -- loaded tables are otherwise kept, as upstream's ``update(force=False)`` does."""
        for name in ("_offset", "_quantized_cdf", "_cdf_length"):
            setattr(self, name, torch.zeros((0,), dtype=torch.int32, device=getattr(self, name).device))

    def _logits_cumulative(self, inputs):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(self.matrices[i].detach().float().cpu()), logits)
            logits = logits + self.biases[i].detach().float().cpu()
            if i < len(self.filters):
                logits = logits + torch.tanh(self.factors[i].detach().float().cpu()) * torch.tanh(logits)
        return logits

    def medians(self):
        return self.quantiles.detach()[:, 0, 1].float()

    def likelihood_coefficients(self):
        """[C, 58] pre-activated parameters of the cumulative-logit network for basic_eb_nll_per_image_dev:
        per layer softplus(matrix) row-major, bias, tanh(factor)."""
        assert self.filters == (3, 3, 3, 3), "the likelihood kernel is specialised for filters (3,3,3,3)"
        with torch.no_grad():
            parts = []
            for i in range(5):
                parts.append(F.softplus(self.matrices[i].detach().float().cpu()).reshape(self.channels, -1))
                parts.append(self.biases[i].detach().float().cpu().reshape(self.channels, -1))
                if i < 4:
                    parts.append(torch.tanh(self.factors[i].detach().float().cpu()).reshape(self.channels, -1))
            coef = torch.cat(parts, 1)
        assert coef.shape[1] == 58
        return coef.contiguous()

    def update(self, force=False):
        """EntropyBottleneck.update() (upstream): tables already present -- from an earlier update or a loaded checkpoint -- are
        kept unless ``force`` (compressai_coder.py:131-151 calls it with force=False).  Returns True when rebuilt."""
        if self._offset.numel() > 0 and not force:
            return False
        with torch.no_grad():
            q = self.quantiles.detach().float().cpu()
            medians = q[:, 0, 1]
            minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
            offset = -minima
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = int(pmf_length.max().item())
            samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
            lower = self._logits_cumulative(samples - 0.5)
            upper = self._logits_cumulative(samples + 0.5)
            sign = -torch.sign(lower + upper)
            pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
            tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
        dev = self.quantiles.device
        self._offset = offset.to(dev)
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        return True

    def host_tables(self):
        """(cdf int32 [C, L], cdf_length, offset) of the current table buffers."""
        return (self._quantized_cdf.cpu().numpy().astype(np.int32), self._cdf_length.cpu().numpy().astype(np.int32).reshape(-1),
                self._offset.cpu().numpy().astype(np.int32).reshape(-1))

    def build_tables(self):
        self.update(force=True)
        return self.host_tables()


class CompressAIEntropyBottleneckPriorCoder(HotPathModule):
    """compressai_coder.py:87-248."""

    def __init__(self, entropy_bottleneck_channels=256, eps=1e-7, use_inner_aux_opt=False, use_bit_rate_loss=True,
                 freeze_params=False, training_output_straight_through=False, **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)
        self.eps = eps
        self._tables = None
        self._medians_dev = None

    def _ready(self):
        if self._tables is None:
            raise RuntimeError("Not Initialized! Should call self.update_state() before coding!")
        if self._medians_dev is None or self._medians_dev.device != self.device:
            self._medians_dev = self.entropy_bottleneck.medians().to(self.device).contiguous()

    def update_state(self, *args, force=False, **kwargs) -> None:  # :247-248 -> self.update(force) -> EntropyBottleneck.update(force)
        self.entropy_bottleneck.update(force=force)
        cdf, lengths, offsets = self.entropy_bottleneck.host_tables()
        self._tables = K.RansTables(cdfs=cdf, cdf_sizes=lengths, offsets=offsets, precision=16, bypass=True, bypass_precision=4)
        self._cdf_host = (cdf, lengths, offsets)
        self._medians_dev = None
        self._coef_dev = None

    def forward(self, input, *args, channel_gains=None, channel_gains_inv=None, **kwargs):
        """Eval-mode forward (:203-228): dequantised latent round(z - median) + median, and the rate estimate
        metric_dict["prior_entropy"] = -sum log(likelihood) / batch (nats)."""
        self._ready()
        if channel_gains is not None:
            input = input * channel_gains.reshape(1, -1, 1, 1)
        if getattr(self, "rate_proxy", "round") == "noise":   # upstream quantize(..., "noise"): the train-mode proxy
            zhat = input + (torch.rand_like(input) - 0.5)
        else:
            _, _, zhat = K.eb_quantize_index(input, self._medians_dev)
        if getattr(self, "_coef_dev", None) is None or self._coef_dev.device != self.device:
            self._coef_dev = self.entropy_bottleneck.likelihood_coefficients().to(self.device)
        self.update_cache("metric_dict", prior_entropy=K.eb_nll_per_image(zhat, self._coef_dev, 1e-9).mean())
        if channel_gains_inv is not None:
            zhat = zhat * channel_gains_inv.reshape(1, -1, 1, 1)
        return zhat

    supports_lazy_encode = True
    accepts_buffer = True  # decode() reads the stream through the buffer protocol (bytes or memoryview)

    def encode(self, input, *args, channel_gains=None, channel_gains_inv=None, lazy=False, **kwargs) -> bytes:  # :230-236
        self._ready()
        if channel_gains is not None:
            input = input * channel_gains.reshape(1, -1, 1, 1)
        sym, idx, _ = K.eb_quantize_index(input, self._medians_dev)
        n = sym[0].numel()
        body = PendingBody(self._tables, self._tables.encode_batch_begin(sym.reshape(-1), idx.reshape(-1), n), input.shape[-2:])
        return body if lazy else body.result()

    def decode(self, byte_string, *args, channel_gains=None, channel_gains_inv=None, **kwargs):  # :238-245
        self._ready()
        h, w, B = K.frame_header(byte_string)
        shape, C = (h, w), self.entropy_bottleneck.channels
        idx = torch.arange(C, device=self.device, dtype=torch.int32).reshape(1, C, 1, 1).expand(B, C, *shape).contiguous()
        sym = self._tables.decode_batch_from_frame(byte_string, idx.reshape(-1), C * shape[0] * shape[1])
        zhat = K.eb_dequantize(sym.reshape(B, C, *shape), self._medians_dev)
        if channel_gains_inv is not None:
            zhat = zhat * channel_gains_inv.reshape(1, -1, 1, 1)
        return zhat


class CompressAISlimmableEntropyBottleneckPriorCoder(HotPathModule):
    """compressai_coder.py:251-338: one EntropyBottleneck per slimmable width."""

    def __init__(self, entropy_bottleneck_channels_list=[256], **kwargs):
        super().__init__()
        self.entropy_bottleneck_channels_list = list(entropy_bottleneck_channels_list)
        self.entropy_bottlenecks = nn.ModuleList(
            [CompressAIEntropyBottleneckPriorCoder(c, **kwargs) for c in self.entropy_bottleneck_channels_list])

    def _pick(self, channels, slim_level):
        return self.entropy_bottlenecks[self.entropy_bottleneck_channels_list.index(channels) if slim_level is None else slim_level]

    def forward(self, input, *args, slim_level=None, **kwargs):
        return self._pick(input.shape[1], slim_level)(input, *args, **kwargs)

    def encode(self, input, *args, slim_level=None, **kwargs) -> bytes:
        return self._pick(input.shape[1], slim_level).encode(input, *args, **kwargs)

    def decode(self, byte_string, *args, slim_level=None, **kwargs):
        if slim_level is None:
            # the reference dereferences an undefined tensor here (compressai_coder.py:333); require the level
            raise ValueError("slim_level is required to decode a slimmable entropy bottleneck stream")
        return self.entropy_bottlenecks[slim_level].decode(byte_string, *args, **kwargs)

    def update_state(self, *args, **kwargs) -> None:
        for eb in self.entropy_bottlenecks:
            eb.update_state(*args, **kwargs)


def gaussian_conditional_tables(scale_table: torch.Tensor, tail_mass=1e-9):
    """GaussianConditional.update() (upstream; an in-tree restatement sits commented out at
    pgm_coder.py:2095-2135).  Returns (cdf int32 [T, L], cdf_length, offset)."""
    from scipy.stats import norm
    st = scale_table.float().cpu()
    multiplier = -norm.ppf(tail_mass / 2)
    pmf_center = torch.ceil(st * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(torch.max(pmf_length).item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    scale = st.unsqueeze(1).float()

    def phi(v):
        return 0.5 * torch.erfc(-(2 ** -0.5) * v)

    upper = phi((0.5 - samples) / scale)
    lower = phi((-0.5 - samples) / scale)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return cdf, (pmf_length + 2).numpy().astype(np.int32), (-pmf_center).numpy().astype(np.int32)


class GaussianConditional(nn.Module):
    """State holder of the y-coder's ``gaussian_conditional`` (compressai_coder.py:345: ``GaussianConditional(None)``): the
    buffers that travel in the reference's checkpoints -- EntropyModel's ``_offset / _quantized_cdf / _cdf_length`` plus
    ``scale_table`` and ``scale_bound`` and the two LowerBound children -- empty until ``update_state()`` and resized to the
    checkpoint's on load exactly as compressai_coder.py:298-319 does with update_registered_buffers."""

    def __init__(self, scale_bound=0.11, tail_mass=1e-9, likelihood_bound=1e-9):
        super().__init__()
        self.tail_mass = float(tail_mass)
        self.likelihood_lower_bound = nn.Module()
        self.likelihood_lower_bound.register_buffer("bound", torch.Tensor([float(likelihood_bound)]))
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self.lower_bound_scale = nn.Module()
        self.lower_bound_scale.register_buffer("bound", torch.Tensor([float(scale_bound)]))
        self.register_buffer("scale_table", torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for name in ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"):
            v = state_dict.get(prefix + name)
            if v is not None and v.shape != getattr(self, name).shape:
                cur = getattr(self, name)
                setattr(self, name, torch.zeros(v.shape, dtype=cur.dtype, device=cur.device))
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def update_scale_table(self, scale_table, force=False):
        """GaussianConditional.update_scale_table (upstream): tables already present (an earlier update or a loaded
        checkpoint) are kept unless ``force``.  Returns True when the tables were rebuilt."""
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.scale_table.device
        self.scale_table = scale_table.float().to(dev)
        cdf, lengths, offsets = gaussian_conditional_tables(self.scale_table, self.tail_mass)
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._cdf_length = torch.from_numpy(lengths).to(dev)
        self._offset = torch.from_numpy(offsets).to(dev)
        return True

    def host_tables(self):
        return (self._quantized_cdf.cpu().numpy().astype(np.int32), self._cdf_length.cpu().numpy().astype(np.int32).reshape(-1),
                self._offset.cpu().numpy().astype(np.int32).reshape(-1))


class CompressAIGaussianConditionalCoder(HotPathModule):
    """compressai_coder.py:341-397: zero-mean Gaussian with hyperprior scales."""

    def __init__(self, use_bit_rate_loss=True, training_output_straight_through=False, scale_bound=0.11, **kwargs):
        super().__init__()
        self.gaussian_conditional = GaussianConditional(scale_bound=scale_bound)
        self.scale_bound = float(scale_bound)
        self._tables = None
        self._scale_table_dev = None

    @property
    def scale_table(self):
        return self.gaussian_conditional.scale_table

    def update_state(self, *args, force=False, **kwargs) -> None:  # :395-397 -> update_scale_table(get_scale_table())
        gc = self.gaussian_conditional
        gc.update_scale_table(get_scale_table(), force=force)
        cdf, lengths, offsets = gc.host_tables()
        self._tables = K.RansTables(cdfs=cdf, cdf_sizes=lengths, offsets=offsets, precision=16, bypass=True, bypass_precision=4)
        self._cdf_host = (cdf, lengths, offsets)
        self._scale_table_dev = None

    def _ready(self):
        if self._tables is None:
            raise RuntimeError("Not Initialized! Should call self.update_state() before coding!")
        if self._scale_table_dev is None or self._scale_table_dev.device != self.device:
            self._scale_table_dev = self.scale_table.to(self.device).contiguous()

    @staticmethod
    def _crop(prior, h, w):
        return prior[..., :h, :w].contiguous()

    def forward(self, y, *args, prior=None, channel_gains=None, channel_gains_inv=None, **kwargs):
        self._ready()
        if channel_gains is not None:
            y = y * channel_gains.reshape(1, -1, 1, 1)
        scales = self._crop(prior, *y.shape[-2:])
        if getattr(self, "rate_proxy", "round") == "noise":   # upstream quantize(..., "noise"): the train-mode proxy
            yhat = y + (torch.rand_like(y) - 0.5)
        else:
            _, _, yhat = K.gc_quantize_index(y, scales, self._scale_table_dev, self.scale_bound)
        self.update_cache("metric_dict", prior_entropy=K.gauss_nll_per_image(yhat, scales, False, self.scale_bound, 1e-9).mean())
        if channel_gains_inv is not None:
            yhat = yhat * channel_gains_inv.reshape(1, -1, 1, 1)
        return yhat

    supports_lazy_encode = True
    accepts_buffer = True  # decode() reads the stream through the buffer protocol (bytes or memoryview)

    def encode(self, y, *args, prior=None, channel_gains=None, channel_gains_inv=None, lazy=False, **kwargs):  # :377-385
        self._ready()
        if channel_gains is not None:
            y = y * channel_gains.reshape(1, -1, 1, 1)
        sym, idx, _ = K.gc_quantize_index(y, self._crop(prior, *y.shape[-2:]), self._scale_table_dev, self.scale_bound,
                                          want_yhat=False)
        body = PendingBody(self._tables, self._tables.encode_batch_begin(sym.reshape(-1), idx.reshape(-1), sym[0].numel()), y.shape[-2:])
        return body if lazy else body.result()

    def decode(self, byte_string, *args, prior=None, channel_gains=None, channel_gains_inv=None, **kwargs):  # :387-393
        self._ready()
        h, w, _ = K.frame_header(byte_string)
        shape = (h, w)
        scales = self._crop(prior, *shape)
        _, idx, _ = K.gc_quantize_index(scales, scales, self._scale_table_dev, self.scale_bound, want_yhat=False)
        sym = self._tables.decode_batch_from_frame(byte_string, idx.reshape(-1), idx[0].numel())
        yhat = K.i32_to_f32(sym.reshape(idx.shape))
        if channel_gains_inv is not None:
            yhat = yhat * channel_gains_inv.reshape(1, -1, 1, 1)
        return yhat
