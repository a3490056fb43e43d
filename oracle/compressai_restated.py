"""Restatement of the CompressAI 1.2.3 pieces the reference imports (requirements.txt:15) --
CPU ORACLE / fixture-generation infrastructure only.

compressai is NOT vendored under /root/reference and is not installed here, so these classes
restate its published algorithm (PyTorch-CPU fp32).  They serve two purposes:
  * stand-ins registered as ``compressai.*`` when tests/golden/make_golden.py imports the
    reference's own Python (latent graph driver, PGM coders, masked conv) to generate vectors;
  * the arithmetic behind oracle/codec_oracle.py.
Parity status of THIS file: unpinned (no reference test or fixture exercises compressai).
The entropy coder underneath (``BufferedRansEncoder``/``RansDecoder``) is the reference's own
csrc/rans build (oracle/_ref) when present, else the C oracle.
"""
import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import rans_oracle as ro


class LowerBoundFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        return g, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return torch.max(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum, self.reparam_offset = float(minimum), float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((self.minimum + self.reparam_offset ** 2) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        C = x.shape[1]
        norm = F.conv2d(x ** 2, self.gamma_reparam(self.gamma).reshape(C, C, 1, 1), self.beta_reparam(self.beta))
        return x * (torch.sqrt(norm) if self.inverse else torch.rsqrt(norm))


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, output_padding=stride - 1,
                              padding=kernel_size // 2)


class MaskedConv2d(nn.Conv2d):
    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0

    def forward(self, x):
        self.weight.data *= self.mask
        return super().forward(x)


def _ref_rans():
    return ro.load_ref()[1]


class BufferedRansEncoder:
    def __init__(self):
        r = _ref_rans()
        self._impl = r.BufferedRansEncoder() if r is not None else None
        self._buf = []

    def encode_with_indexes(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        if self._impl is not None:
            return self._impl.encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets)
        self._buf.append((symbols, indexes, cdfs, cdf_sizes, offsets))

    def flush(self):
        if self._impl is not None:
            return self._impl.flush()
        sym = np.concatenate([np.asarray(b[0], np.int32) for b in self._buf])
        idx = np.concatenate([np.asarray(b[1], np.int32) for b in self._buf])
        e = ro.Rans64Encoder(16, True, 4)
        cd = self._buf[0][2]
        L = max(len(c) for c in cd)
        arr = np.zeros((len(cd), L), np.int32)
        for i, c in enumerate(cd):
            arr[i, : len(c)] = c
        e.init_cdf_params(arr, self._buf[0][3], self._buf[0][4])
        self._buf = []
        return e.encode_with_indexes(sym, idx)


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        b = BufferedRansEncoder()
        b.encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets)
        return b.flush()


class RansDecoder:
    def __init__(self):
        r = _ref_rans()
        self._impl = r.RansDecoder() if r is not None else None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdf_sizes, offsets):
        if self._impl is not None:
            return self._impl.decode_with_indexes(encoded, indexes, cdfs, cdf_sizes, offsets)
        d = ro.Rans64Decoder(16, True, 4)
        L = max(len(c) for c in cdfs)
        arr = np.zeros((len(cdfs), L), np.int32)
        for i, c in enumerate(cdfs):
            arr[i, : len(c)] = c
        d.init_cdf_params(arr, cdf_sizes, offsets)
        return d.decode_with_indexes(encoded, np.asarray(indexes, np.int32)).tolist()


def pmf_to_quantized_cdf(pmf, precision=16):
    return torch.IntTensor(ro.pmf_to_quantized_cdf(pmf.tolist(), precision))


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def quantize(self, inputs, mode, means=None):
        if mode == "noise":
            return inputs + torch.empty_like(inputs).uniform_(-0.5, 0.5)
        outputs = inputs.clone()
        if means is not None:
            outputs -= means
        outputs = torch.round(outputs)
        if mode == "dequantize":
            if means is not None:
                outputs += means
            return outputs
        return outputs.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.type(dtype)
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    def compress(self, inputs, indexes, means=None):
        symbols = self.quantize(inputs, "symbols", means)
        strings = []
        for i in range(symbols.size(0)):
            strings.append(RansEncoder().encode_with_indexes(
                symbols[i].reshape(-1).int().tolist(), indexes[i].reshape(-1).int().tolist(), self._quantized_cdf.tolist(),
                self._cdf_length.reshape(-1).int().tolist(), self._offset.reshape(-1).int().tolist()))
        return strings

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        cdf = self._quantized_cdf
        outputs = cdf.new_empty(indexes.size())
        for i, s in enumerate(strings):
            values = RansDecoder().decode_with_indexes(s, indexes[i].reshape(-1).int().tolist(), cdf.tolist(),
                                                       self._cdf_length.reshape(-1).int().tolist(),
                                                       self._offset.reshape(-1).int().tolist())
            outputs[i] = torch.tensor(values, dtype=outputs.dtype).reshape(outputs[i].size())
        return self.dequantize(outputs, means, dtype)


class EntropyBottleneck(EntropyModel):
    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale, self.tail_mass = float(init_scale), float(tail_mass)
        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        # nn.Parameters named _matrixN / _biasN / _factorN: the layout the reference's converter writes for its pinned
        # compressai (tools/compressai_checkpoint_to_cbench.py:16-25 "nn.ParameterList to nn.Parameters")
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, filters[i + 1], filters[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, filters[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, filters[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def loss(self):
        """Auxiliary loss of the published algorithm: the learned quantiles should sit where the cumulative's logits equal
        (-target, 0, target).  Only the reference's train-mode forward reads it (compressai_coder.py:126-128,186-198)."""
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def _logits_cumulative(self, inputs, stop_gradient=True):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, f"_matrix{i:d}").detach()), logits) + getattr(self, f"_bias{i:d}").detach()
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i:d}").detach()) * torch.tanh(logits)
        return logits

    def _likelihood(self, inputs, stop_gradient=False):
        lower = self._logits_cumulative(inputs - 0.5)
        upper = self._logits_cumulative(inputs + 0.5)
        sign = -torch.sign(lower + upper)
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower)), lower, upper

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        with torch.no_grad():
            medians = self.quantiles[:, 0, 1]
            minima = torch.clamp(torch.ceil(medians - self.quantiles[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - medians).int(), min=0)
            self._offset = -minima
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = pmf_length.max().item()
            samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
            pmf, lower, upper = self._likelihood(samples, stop_gradient=True)
            pmf = pmf[:, 0, :]
            tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
            self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
            self._cdf_length = pmf_length + 2
        return True

    def forward(self, x, training=None):
        if training is None:
            training = self.training
        perm = np.arange(len(x.shape))
        perm[0], perm[1] = perm[1], perm[0]
        inv_perm = np.arange(len(x.shape))[np.argsort(perm)]
        x = x.permute(*perm).contiguous()
        shape = x.size()
        values = x.reshape(x.size(0), 1, -1)
        outputs = self.quantize(values, "noise" if training else "dequantize", self._get_medians())
        likelihood, _, _ = self._likelihood(outputs)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        outputs = outputs.reshape(shape).permute(*inv_perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*inv_perm).contiguous()
        return outputs, likelihood

    @staticmethod
    def _build_indexes(size):
        dims = len(size)
        N, C = size[0], size[1]
        view_dims = np.ones((dims,), dtype=np.int64)
        view_dims[1] = -1
        return torch.arange(C).view(*view_dims).int().repeat(N, 1, *size[2:])

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        medians = self._get_medians().detach()
        for _ in range(len(x.size()) - 2):
            medians = medians.unsqueeze(-1) if medians.dim() < len(x.size()) else medians
        medians = self._get_medians().detach().reshape(1, -1, *([1] * (len(x.size()) - 2))).expand(x.size(0), -1, *([-1] * (len(x.size()) - 2)))
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), *size)
        indexes = self._build_indexes(output_size)
        medians = self._get_medians().detach().reshape(1, -1, *([1] * len(size))).expand(len(strings), -1, *([-1] * len(size)))
        return super().decompress(strings, indexes, medians.dtype, medians)


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    @staticmethod
    def _standardized_cumulative(inputs):
        return 0.5 * torch.erfc(-(2 ** -0.5) * inputs)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.Tensor(tuple(float(s) for s in scale_table))
        self.update()
        return True

    def update(self):
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = self.scale_table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def _likelihood(self, inputs, scales, means=None):
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        return self._standardized_cumulative((0.5 - values) / scales) - self._standardized_cumulative((-0.5 - values) / scales)

    def forward(self, inputs, scales, means=None, training=None):
        if training is None:
            training = self.training
        outputs = self.quantize(inputs, "noise" if training else "dequantize", means)
        likelihood = self._likelihood(outputs, scales, means)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes


def update_registered_buffers(module, module_name, buffer_names, state_dict, policy="resize_if_empty", dtype=torch.int):
    for name in buffer_names:
        key = f"{module_name}.{name}"
        if key in state_dict:
            getattr(module, name).resize_(state_dict[key].size())
