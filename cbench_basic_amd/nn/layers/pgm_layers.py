"""Slimmable transforms of BaSIC -- cbench/nn/layers/pgm_layers.py:714-927,1054-1075
(SlimmableConv2dPGMModel and its Hyperprior / MeanScale-hyper subclasses).

forward(input, pgm=one_hot): level = argmax(one_hot) (pgm_layers.py:783); every DynamicConv2d then uses
out_channels_list[level] and whatever input width arrives (slimmable_layers.py:157-183).  Here each level is
compiled once into fused MFMA layer plans over the ACTIVE weight slice and cached.
"""
import torch
import torch.nn as nn

from .. import kernels as K
from .slimmable_layers import DynamicConv2d, DynamicGDN


class SlimmableConv2dPGMModel(nn.Module):
    def __init__(self, *args, in_channels=256, in_groups=1, out_channels=256, mid_channels_list=(64, 128, 256), **kwargs):
        super().__init__()
        self.in_channels, self.in_groups, self.out_channels = in_channels, in_groups, out_channels
        self.mid_channels_list = list(mid_channels_list)
        self.out_channels_list = list(out_channels) if isinstance(out_channels, (list, tuple)) else [out_channels] * len(self.mid_channels_list)
        self.default_pgm = nn.Parameter(torch.zeros(1, 1, len(self.mid_channels_list)))  # state-dict compat (BasePGMLayer :111)
        self.default_agg_pgm = nn.Parameter(torch.zeros(1, len(self.mid_channels_list)))        # (BasePGMLayer :117)
        self.pgm_model = self.build_pgm_model()
        self._plans = {}
        self._key = None
        self.total_ops = 0.0

    def build_pgm_model(self) -> nn.Module:
        raise NotImplementedError()

    def num_levels(self):
        return len(self.mid_channels_list)

    def _compile(self, level):
        layers, plans, i, cin = list(self.pgm_model), [], 0, self.in_channels
        while i < len(layers):
            m = layers[i]
            assert isinstance(m, DynamicConv2d)
            cout = m.out_channels_at(level)
            act, gamma, beta = K.ACT_NONE, None, None
            if i + 1 < len(layers):
                n = layers[i + 1]
                if isinstance(n, DynamicGDN):
                    assert n.channels_list[level] == cout
                    act = K.ACT_IGDN if n.inverse else K.ACT_GDN
                    gamma, beta = n.effective(level)
                    i += 1
                elif isinstance(n, nn.LeakyReLU):
                    act = K.ACT_LEAKY_RELU
                    i += 1
                elif isinstance(n, nn.ReLU):
                    act = K.ACT_RELU
                    i += 1
            c = m.conv
            plans.append(K.ConvPlan(c.weight.detach(), c.bias.detach() if c.bias is not None else None, m.stride, m.padding,
                                    m.stride - 1 if m.transposed else 0, m.transposed, act, gamma, beta, cin_active=cin, cout_active=cout))
            cin = cout
            i += 1
        return plans

    def plans(self, level):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key != self._key:
            self._plans, self._key = {}, key
        if level not in self._plans:
            self._plans[level] = self._compile(level)
        return self._plans[level]

    @staticmethod
    def level_of(pgm, n_levels):
        if pgm is None:
            return n_levels - 1
        if isinstance(pgm, int):
            return pgm
        return int(pgm.reshape(-1, pgm.shape[-1])[0].argmax().item())

    def flops(self, level, batch, h, w):
        total = 0
        for p in self.plans(level):
            total += p.flops(batch, h, w)
            h, w = p.out_hw(h, w)
        return total

    def reference_ops(self, level, batch, h, w):
        """The reference's complexity counter for one forward at this width level: multiply-accumulates plus one per
        output element for a bias, summed over DynamicConv2d (count_dynamic_convNd, slimmable_layers.py:186-206) and
        DynamicGDN (count_gdn: a C x C 1x1 convolution with bias, :284-293); activations count nothing.  This is the
        "FLOPs" the greedy complexity search ranks controller settings by (pgm_layers.py:816-825)."""
        total, cin = 0, self.in_channels
        for m in self.pgm_model:
            if isinstance(m, DynamicConv2d):
                cout, k, s, p = m.out_channels_at(level), m.kernel_size, m.stride, m.padding
                if m.transposed:
                    h, w = (h - 1) * s - 2 * p + k + (s - 1), (w - 1) * s - 2 * p + k + (s - 1)
                else:
                    h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
                total += batch * cout * h * w * (cin * k * k + (1 if m.conv.bias is not None else 0))
                cin = cout
            elif isinstance(m, DynamicGDN):
                total += batch * cin * h * w * (cin + 1)
        return total

    def get_current_flops(self, input=None):
        """total_ops of the last forward (pgm_layers.py:761-763)."""
        return self.total_ops

    def forward(self, input, *args, pgm=None, input_mask=None, **kwargs):
        if input_mask is not None:
            raise NotImplementedError("input_mask")
        if pgm is not None and not isinstance(pgm, int) and pgm.dim() == 4 and pgm.shape[0] != 1:
            raise NotImplementedError("per-sample slim levels")
        level = self.level_of(pgm, self.num_levels())
        self.total_ops = float(self.reference_ops(level, input.shape[0], input.shape[2], input.shape[3]))
        x = input
        for p in self.plans(level):
            x = p(x)
        return x


class HyperpriorAnalysisSlimmableConv2dPGMModel(SlimmableConv2dPGMModel):  # pgm_layers.py:904-914
    def build_pgm_model(self):
        m, o = self.mid_channels_list, self.out_channels_list
        return nn.Sequential(DynamicConv2d(self.in_channels, m), DynamicGDN(m), DynamicConv2d(max(m), m), DynamicGDN(m),
                             DynamicConv2d(max(m), m), DynamicGDN(m), DynamicConv2d(max(m), o))


class HyperpriorSynthesisSlimmableConv2dPGMModel(SlimmableConv2dPGMModel):  # pgm_layers.py:917-927
    def build_pgm_model(self):
        m, o = self.mid_channels_list, self.out_channels_list
        return nn.Sequential(DynamicConv2d(self.in_channels, m, transposed=True), DynamicGDN(m, inverse=True),
                             DynamicConv2d(max(m), m, transposed=True), DynamicGDN(m, inverse=True),
                             DynamicConv2d(max(m), m, transposed=True), DynamicGDN(m, inverse=True),
                             DynamicConv2d(max(m), o, transposed=True))


class MeanScaleHyperpriorHyperAnalysisSlimmableConv2dPGMModel(SlimmableConv2dPGMModel):  # pgm_layers.py:1054-1063
    def build_pgm_model(self):
        m, o = self.mid_channels_list, self.out_channels_list
        return nn.Sequential(DynamicConv2d(self.in_channels, m, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True),
                             DynamicConv2d(max(m), m), nn.LeakyReLU(inplace=True), DynamicConv2d(max(m), o))


class MeanScaleHyperpriorHyperSynthesisSlimmableConv2dPGMModel(SlimmableConv2dPGMModel):  # pgm_layers.py:1065-1075
    def build_pgm_model(self):
        m, o = self.mid_channels_list, self.out_channels_list
        assert max(o) == max(m) * 2
        l2 = [c * 3 // 2 for c in m]
        return nn.Sequential(DynamicConv2d(self.in_channels, m, transposed=True), nn.LeakyReLU(inplace=True),
                             DynamicConv2d(max(m), l2, transposed=True), nn.LeakyReLU(inplace=True),
                             DynamicConv2d(max(l2), o, stride=1, kernel_size=3))
