"""Topologically-grouped autoregressive Gaussian y-coder --
GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder (cbench/modules/prior_model/prior_coder/
pgm_coder.py:983) with its bases TopoGroupPGMPriorCoder (:863-981), GaussianPGMPriorCoderImpl
(:718-821), NNTrainablePGMPriorCoder (:218-616) and the optional
TopoGroupDynamicMaskConv2dContextModel (cbench/nn/layers/masked_conv.py:231-305).

Coding loop (identical order of operations to the reference):
  for each topo group g = 0, 1, ...                                   pgm_coder.py:921-941 / :958-978
      params = context model( decoded-so-far buffer, prior )           at the positions of g only
      idx    = argmin_j |scale - table[j]| ;  mu = mean                :802-821, torch_ans.py:279-282
      encode: sym = round(y - mu); buffer = sym + mu        decode: sym <- rANS; buffer = sym + mu
  one rANS stream per image over the group-major symbol order          :943-947, :951,:971
Where the reference evaluates the full H x W context model for EVERY group and gathers with boolean
masks (nonzero + sync per group), this implementation precomputes, per latent shape, the static
per-group element / position lists and launches the masked-conv kernels on those positions only.

Stream format: for batch size 1 (how the reference tests, configs/*: batch_size=1) the bytes are the
reference's.  For batch size > 1 the reference codes ALL images into one serial stream
(data[mask] spans the batch); ``batch_stream_mode="per_image"`` (default for B > 1) instead emits one
independent stream per image -- framed as <I n> <n x I length> streams -- which is what lets 256 images
be coded concurrently; ``"reference"`` reproduces the single stream.
"""
import contextlib
import struct
import threading
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from ....base import HotPathModule
from ....nn import kernels as K
from .... import _lib
from .compressai_coder import get_scale_table
from .torch_ans import gaussian_ans_params


def default_topo_groups(method: str, channel_groups: int, h: int, w: int) -> np.ndarray:
    """_get_default_pgm (pgm_coder.py:1416-1491): int64 [G_c, H, W] topo-group ids."""
    G = channel_groups
    t = np.zeros((G, h, w), dtype=np.int64)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    checker = ((yy + xx) % 2).astype(np.int64)  # 1 on (even,odd) and (odd,even)
    if method == "none":
        pass
    elif method == "scanline":
        t[:] = (yy * w + xx)[None]
    elif method == "zigzag":
        t[:] = (yy + xx)[None]
    elif method == "checkerboard":
        t[:] = checker[None]
    elif method == "half-checkerboard":
        t[:] = 1
        t[:, 1::2, 1::2] = 0
    elif method == "halfinv-checkerboard":
        t[:, 1::2, 1::2] = 1
    elif method == "quarter-checkerboard":
        t[:] = 1
        t[:, 1::4, 3::4] = 0
        t[:, 3::4, 1::4] = 0
    elif method == "interlace-checkerboard":
        for i in range(G):
            t[i] = (1 - checker) if i % 2 == 0 else checker
    elif method == "raster2x2":
        t[:, 0::2, 1::2] = 1
        t[:, 1::2, 0::2] = 2
        t[:, 1::2, 1::2] = 3
    elif method == "channelwise":
        for i in range(G):
            t[i] = i
    elif method == "channelwise-checkerboard":
        for i in range(G):
            t[i] = i * 2 + checker
    elif method == "channelwise-scanline":
        for i in range(G):
            t[i] = yy * w + xx + i * h * w
    elif method == "channelwise-g10":
        splits = [1] * 9 + [G - 9]
        s = 0
        for i, n in enumerate(splits):
            t[s:s + n] = i
            s += n
    elif method == "elic":
        splits = [1, 1, 2, 4, G - 8]
        s = 0
        for i, n in enumerate(splits):
            t[s:s + n] = i * 2 + checker[None]
            s += n
    else:
        raise NotImplementedError(f"Unknown default_topo_group_method {method}")
    return t


class TopoGroupDynamicMaskConv2d(nn.Conv2d):
    """Parameter holder of cbench/nn/layers/masked_conv.py:69-100 (weights only; the masked
    convolution itself is csrc/mconv.hip)."""

    def __init__(self, in_channels, out_channels, kernel_size, dynamic_channel_groups=1, allow_same_topogroup_conv=False, **kwargs):
        kwargs.pop("groups", None)
        for k in ("allow_continuous_topo_groups", "continuous_topo_groups_training_use_uniform_noise",
                  "continuous_topo_groups_smooth_func", "detach_context_model"):
            kwargs.pop(k, None)
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.dynamic_channel_groups = dynamic_channel_groups
        self.allow_same_topogroup_conv = allow_same_topogroup_conv

    def forward(self, *a, **k):
        raise RuntimeError("masked convolutions run through MaskedConvPlan on the coding positions")


class TopoGroupDynamicMaskConv2dContextModel(nn.Module):
    """masked_conv.py:231-305 parameter layout (context_prediction, param_merger_in, param_merger_out)."""

    def __init__(self, in_channels=192, out_channels=384, kernel_size=5, use_param_merger=True, param_merger_in_channels=None,
                 param_merger_mid_channels_list=None, param_merger_kernel_size=1, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.padding = kernel_size // 2
        self.use_param_merger = use_param_merger
        self.context_prediction = TopoGroupDynamicMaskConv2d(in_channels, out_channels, kernel_size, padding=self.padding)
        if use_param_merger:
            if param_merger_kernel_size != 1:
                raise NotImplementedError("param merger kernels other than 1x1")
            self.param_merger_out_channels = out_channels
            self.param_merger_in_channels = out_channels * 2 if param_merger_in_channels is None else param_merger_in_channels
            self.param_merger_mid_channels_list = [out_channels * 5 // 3, out_channels * 4 // 3] \
                if param_merger_mid_channels_list is None else param_merger_mid_channels_list
            mids = self.param_merger_mid_channels_list
            self.param_merger_in = TopoGroupDynamicMaskConv2d(self.param_merger_in_channels, mids[0], 1, allow_same_topogroup_conv=True)
            layers = []
            for i in range(len(mids) - 1):
                layers += [nn.LeakyReLU(inplace=True), TopoGroupDynamicMaskConv2d(mids[i], mids[i + 1], 1, allow_same_topogroup_conv=True)]
            layers += [nn.LeakyReLU(inplace=True), TopoGroupDynamicMaskConv2d(mids[-1], out_channels, 1, allow_same_topogroup_conv=True)]
            self.param_merger_out = nn.Sequential(*layers)


_CAPTURE_LOCK = threading.Lock()


@contextlib.contextmanager
def _capture(device, graph):
    """HIP-graph capture that is safe beside other host threads (stream workers with their own codec replicas): one capture at
    a time in the process, in thread-local capture mode -- in the default global mode a hipMalloc / hipFree issued by ANY other
    thread while this one captures fails or invalidates the capture.  ``capture_begin`` / ``capture_end`` are called directly on a
    private side stream: the ``torch.cuda.graph`` context manager would first synchronise the whole DEVICE and empty the caching
    allocator (torch 2.10), i.e. drain every other worker's stream and free their cached blocks; here only THIS thread's stream
    is drained."""
    with _CAPTURE_LOCK:
        cur = torch.cuda.current_stream(device)
        cur.synchronize()
        side = torch.cuda.Stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            graph.capture_begin(capture_error_mode="thread_local")
            try:
                yield
            finally:
                graph.capture_end()
        cur.wait_stream(side)


class _GroupPlan:
    """Static coding schedule of one (H, W) latent shape: per topo group the per-image element list
    (ascending flat index = boolean-mask order, pgm_coder.py:898-900) and the spatial positions."""

    def __init__(self, topo: np.ndarray, channels: int, device):
        G, H, W = topo.shape
        self.h, self.w, self.hw = H, W, H * W
        per = channels // G
        full = np.repeat(topo, per, axis=0).reshape(-1)          # post-repeat, :1770
        self.topo_dev = torch.from_numpy(topo.astype(np.int32)).to(device).contiguous()
        self.topo_cat_dev = torch.cat([self.topo_dev, torch.full_like(self.topo_dev, -1)], 0).contiguous()
        # first step that visits a position (all channel groups): when the id-less (-1) merger groups are evaluated
        self.first_dev = torch.from_numpy(topo.min(axis=0).astype(np.int32)).to(device).contiguous()
        self.groups = []
        base = 0
        for g in range(int(topo.max()) + 1):
            elems = np.nonzero(full == g)[0].astype(np.int32)
            pos = np.nonzero((topo == g).any(axis=0).reshape(-1))[0].astype(np.int32)
            self.groups.append(dict(elems=torch.from_numpy(elems).to(device), n=int(elems.size), base=base, pos_np=pos))
            base += int(elems.size)
        self.per_image = base
        self._pos_cache = {}
        # plane layout of the merger's PRIVATE hidden activations: positions ordered by the first step that codes them, so
        # the positions of a step are contiguous (whole cache lines per gather / store; row-major planes give a checkerboard
        # step every other float).  None when that is the row-major order anyway.
        order = np.argsort(topo.min(axis=0).reshape(-1), kind="stable")
        perm = np.empty(H * W, dtype=np.int32)
        perm[order] = np.arange(H * W, dtype=np.int32)
        ident = np.array_equal(perm, np.arange(H * W))
        self.hidden_perm_dev = None if ident else torch.from_numpy(perm).to(device).contiguous()            # position -> slot
        self.hidden_order_dev = None if ident else torch.from_numpy(order.astype(np.int64)).to(device)      # slot -> position

    def positions(self, g, batch, device):
        key = (g, batch)
        if key not in self._pos_cache:
            p = self.groups[g]["pos_np"]
            allp = (np.arange(batch, dtype=np.int64)[:, None] * self.hw + p[None, :]).reshape(-1).astype(np.int32)
            self._pos_cache[key] = torch.from_numpy(allp).to(device)
        return self._pos_cache[key]


class _Kernel:
    """(weight, bias) standing in for a convolution module when the kernel comes with the call."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


class GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(HotPathModule):
    accepts_buffer = True  # decode() reads the stream through the buffer protocol (bytes or memoryview)

    def __init__(self, *args, in_channels=256, channel_groups=1, default_topo_group_method="none", default_num_topo_groups=-1,
                 topo_group_context_model: Optional[TopoGroupDynamicMaskConv2dContextModel] = None, kernel_size=5,
                 use_param_merger=True, use_joint_ar_model_impl=False, param_merger_expand_bottleneck=False,
                 use_autoregressive_encode=True, use_bypass_coding=True, freq_precision=16, bypass_precision=4,
                 lower_bound_scale=0.11, quantizer_params=None, fixed_input_shape=None,
                 force_input_prior_shape_aligned=True, batch_stream_mode="auto", topo_group_predictor=None,
                 pgm_include_dynamic_kernel=False, pgm_include_dynamic_kernel_full=False, pgm_dynamic_kernel_enable_tiling=False,
                 pgm_dynamic_kernel_add_self=False, training_no_quantize_for_likelihood=False, **kwargs):
        super().__init__()
        # pgm_coder.py:225,376-387,413-416: the rate estimate is taken on the residual y - mu under the zero-mean density
        # (eval: round(y - mu); train-mode proxy: y - mu + fresh uniform noise) instead of on the quantised latent.  The
        # BaSIC presets set it (lossy_latent_graph_scalable_ar_models.py:121,261-330).
        self.training_no_quantize_for_likelihood = bool(training_no_quantize_for_likelihood)
        # "round" = the eval-mode forward (the latents the codec really codes); "noise" = the reference's TRAIN-mode proxies
        # (additive uniform noise, torch_ans.py:127-136), set by the complexity search when it is asked for the reference's
        # loss (latent_graph.py:1322, complexity_level_greedy_search_loss_mode="train")
        self.rate_proxy = "round"
        # Dynamic-kernel PGMs (pgm_coder.py:996-1001,1314-1339,1941-1955): the pgm handed to encode / decode / forward is a
        # tuple (topo groups, context-conv weight [1, 2C, C, k, k], bias [1, 2C]) -- the structure AND the kernel of the
        # context model come with the call; with pgm_dynamic_kernel_add_self the module's own kernel is added to it.  The
        # "full" variant (kernels for the merger layers too) leaves its kernels installed on the layers after the call in
        # the reference (set_dynamic_kernel, :1947-1950) and the tiled variant is a per-position kernel (an unfold-matmul
        # with a spatial axis, masked_conv.py:199-201): neither is offered.
        if pgm_include_dynamic_kernel_full or pgm_dynamic_kernel_enable_tiling:
            raise NotImplementedError("pgm_include_dynamic_kernel_full / pgm_dynamic_kernel_enable_tiling")
        self.pgm_include_dynamic_kernel = bool(pgm_include_dynamic_kernel)
        self.pgm_dynamic_kernel_add_self = bool(pgm_dynamic_kernel_add_self)
        if self.pgm_include_dynamic_kernel and (topo_group_context_model is not None or use_joint_ar_model_impl):
            raise NotImplementedError("dynamic kernels need the built-in context convolution")
        self._dyn_layers = {}
        # use_joint_ar_model_impl (pgm_coder.py:1975-2070): raster-scan coding with a plain 1x1 entropy_parameters network
        # on cat(prior, ctx) and "chunk" parameters (scales, then means).  Raster order IS the scanline schedule, so the
        # coder runs on the same kernels: the layers are built from re-ordered views of the weights (_build_layers).
        self.use_joint_ar_model_impl = bool(use_joint_ar_model_impl)
        if self.use_joint_ar_model_impl:
            if channel_groups != 1 or topo_group_context_model is not None or not use_param_merger:
                raise ValueError("use_joint_ar_model_impl needs channel_groups == 1, the built-in context model and the param merger")
            default_topo_group_method = "scanline"
        if not use_autoregressive_encode:
            raise NotImplementedError("use_autoregressive_encode=False")
        # Quantiser (torch_ans.py:16-50,105-121,163-178): "uniform" params [offset, -, step] -> y' = (y - offset) / step,
        # "uniform_scale" params [step] -> y' = y / step.  With pgm_input_dequantized=False (the only mode offered, the
        # reference's default) the whole coder -- context model, parameters, symbols -- lives in the y' domain and only the
        # returned latent is mapped back (y' * step + offset), so the quantiser is an affine map around the coding path.
        if kwargs.get("pgm_input_dequantized", False):
            raise NotImplementedError("pgm_input_dequantized=True")
        quantizer_type = kwargs.get("quantizer_type", "uniform")
        if quantizer_type not in ("uniform", "uniform_scale"):
            raise NotImplementedError(f"quantizer_type {quantizer_type}")
        self.quantizer_type = quantizer_type
        if quantizer_params is None:
            quantizer_params = [0.0, float(1 << (kwargs.get("data_precision", 8) - 1)), 1.0] if quantizer_type == "uniform" else [1.0]
        assert len(quantizer_params) == (3 if quantizer_type == "uniform" else 1)
        self.in_channels = in_channels
        self.channel_groups = channel_groups
        self.default_topo_group_method = default_topo_group_method
        self.kernel_size = kernel_size
        self.padding = (kernel_size // 2, kernel_size // 2)
        self.use_param_merger = use_param_merger
        self.param_merger_expand_bottleneck = param_merger_expand_bottleneck
        self.freq_precision, self.use_bypass_coding, self.bypass_precision = freq_precision, use_bypass_coding, bypass_precision
        # LowerBound(lower_bound_scale) of GaussianPGMPriorCoderImpl (pgm_coder.py:720-721): a module holding the buffer ``bound``
        self.lower_bound_scale = nn.Module()
        self.lower_bound_scale.register_buffer("bound", torch.Tensor([float(lower_bound_scale)]))
        self._lower_bound_scale = float(lower_bound_scale)
        self.register_buffer("quantizer_params", torch.as_tensor(quantizer_params, dtype=torch.float32), persistent=False)
        self._quantizer_host = tuple(float(v) for v in quantizer_params)   # read per call: no device round trip on the hot path
        self.fixed_input_shape = fixed_input_shape
        self.force_input_prior_shape_aligned = force_input_prior_shape_aligned
        self.batch_stream_mode = batch_stream_mode
        self.eps = kwargs.get("eps", 1e-7)
        self.estimate_rate = False
        if default_topo_group_method in ("channelwise-g10", "elic"):  # pgm_coder.py:1164-1171
            self.channel_groups = in_channels // 16
            assert self.channel_groups >= (9 if default_topo_group_method == "channelwise-g10" else 8)
        if in_channels % self.channel_groups:
            raise ValueError("in_channels must divide into channel_groups")
        self.out_channels = in_channels * 2  # mean and scale ("split_interleave")
        G, C2 = self.channel_groups, self.out_channels
        self.topo_group_context_model = topo_group_context_model
        if topo_group_context_model is None:
            # same parameter names as the reference (pgm_coder.py:1185-1239)
            self.conv_kernel_weight = nn.Parameter(torch.zeros(C2, in_channels, kernel_size, kernel_size))
            self.conv_kernel_bias = nn.Parameter(torch.zeros(C2))
            self.context_prediction = TopoGroupDynamicMaskConv2d(in_channels, C2, kernel_size, padding=self.padding,
                                                                dynamic_channel_groups=G)
            if use_param_merger and self.use_joint_ar_model_impl:  # (:1204-1212)
                self.entropy_parameters = nn.Sequential(
                    nn.Conv2d(C2 * 2, C2 * 5 // 3, 1), nn.LeakyReLU(inplace=True),
                    nn.Conv2d(C2 * 5 // 3, C2 * 4 // 3, 1), nn.LeakyReLU(inplace=True), nn.Conv2d(C2 * 4 // 3, C2, 1))
            elif use_param_merger:
                bott = C2 * 4 if param_merger_expand_bottleneck else C2 * 2
                mk = lambda i, o: TopoGroupDynamicMaskConv2d(i, o, 1, dynamic_channel_groups=G * 2, allow_same_topogroup_conv=True)
                self.param_merger = nn.Sequential(mk(C2 * 2, bott), nn.LeakyReLU(inplace=True), mk(bott, bott),
                                                  nn.LeakyReLU(inplace=True), mk(bott, C2 * 2))
        # learned topo groups (pgm_coder.py:1095-1103): at inference the reference reads the predictor's CACHED output
        # (buffer ``topo_group_predictor_cache``, integer ids or logits [1, G_c * L, h, w]); the cache travels in the
        # state_dict, so a tensor (or any module / callable producing one) is enough here
        self.topo_group_predictor = topo_group_predictor if isinstance(topo_group_predictor, nn.Module) else None
        if topo_group_predictor is not None:
            with torch.no_grad():
                cache = topo_group_predictor() if callable(topo_group_predictor) else topo_group_predictor
            self.register_buffer("topo_group_predictor_cache", torch.as_tensor(cache).detach().clone())
        self.scale_table = get_scale_table()
        self._tables = None
        self._layers = None
        self._layer_key = None
        self._plans = {}
        self._scale_table_dev = None
        self._graphs = {}

    # ------------------------------------------------------------------ state
    def update_state(self, *args, **kwargs) -> None:
        """TorchANSPriorCoder.update_state (torch_ans.py:237-251)."""
        freqs, nsym, offsets = gaussian_ans_params(self.scale_table, self.freq_precision, self._lower_bound_scale)
        self._ans_params = (freqs, nsym, offsets)
        self._tables = K.RansTables(freqs=freqs, nsym=nsym, offsets=offsets, precision=self.freq_precision,
                                    bypass=self.use_bypass_coding, bypass_precision=self.bypass_precision)
        self._layers = None
        self._scale_table_dev = None
        self._graphs = {}

    def _ready(self):
        if self._tables is None:
            raise AssertionError("Not Initialized! Should call self.update_state() before coding!")
        if self._scale_table_dev is None or self._scale_table_dev.device != self.device:
            self._scale_table_dev = self.scale_table.to(self.device).contiguous()
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._layers is None or key != self._layer_key:
            self._layers = self._build_layers()
            self._layer_key = key
            self._graphs = {}

    def _build_layers(self, ctx_kernel=None):
        """MaskedConvPlans of the context conv and the merger layers, with the activation that FOLLOWS a
        layer fused into it.  ctx_kernel = (weight, bias) replaces the context convolution's own parameters (dynamic-kernel
        PGMs)."""
        G, C2 = self.channel_groups, self.out_channels
        L = dict()
        cm = self.topo_group_context_model
        if cm is None:
            cp = self.context_prediction
            if ctx_kernel is not None:
                cp = _Kernel(*ctx_kernel)
            L["ctx"] = K.MaskedConvPlan(cp.weight, cp.bias, G, G, False)
            if self.use_param_merger and self.use_joint_ar_model_impl:
                e = self.entropy_parameters
                C = self.in_channels
                # first layer: the reference feeds cat(prior, ctx) (:1619), the workspace holds cat(ctx, prior)
                w0 = torch.cat([e[0].weight[:, C2:], e[0].weight[:, :C2]], 1)
                # last layer: "chunk" output (scales 0..C-1, means C..2C-1) -> the interleaved (mean, scale) pairs the
                # Gaussian kernels read
                order = torch.stack([torch.arange(C, 2 * C), torch.arange(0, C)], 1).reshape(-1)
                L["m"] = [
                    (K.MaskedConvPlan(w0, e[0].bias, 1, 1, True, K.ACT_LEAKY_RELU), "pgm", "pgm"),
                    (K.MaskedConvPlan(e[2].weight, e[2].bias, 1, 1, True, K.ACT_LEAKY_RELU), "pgm", "pgm"),
                    (K.MaskedConvPlan(e[4].weight[order], e[4].bias[order], 1, 1, True, K.ACT_NONE), "pgm", "pgm"),
                ]
                L["dense"] = [(w0, e[0].bias, True, 1), (e[2].weight, e[2].bias, True, 1), (e[4].weight[order], e[4].bias[order], False, 1)]
            elif self.use_param_merger:
                m = self.param_merger
                L["m"] = [
                    (K.MaskedConvPlan(m[0].weight, m[0].bias, 2 * G, 2 * G, True, K.ACT_LEAKY_RELU), "cat", "cat"),
                    (K.MaskedConvPlan(m[2].weight, m[2].bias, 2 * G, 2 * G, True, K.ACT_LEAKY_RELU), "cat", "cat"),
                    # only the first G of the 2G output groups survive (pgm_coder.py:1631-1632)
                    (K.MaskedConvPlan(m[4].weight[:C2], m[4].bias[:C2], 2 * G, G, True, K.ACT_NONE), "cat", "pgm"),
                ]
                if G == 1:
                    # one channel group: two input / output groups (topo id of the position, and -1 for the prior half).  An
                    # output row of the -1 group sees only inputs of the -1 group (mask: topo_in <= topo_out); rows of the
                    # other group see everything.  As dense layers: the masked block of the weights is zero.
                    def masked(conv):
                        w = conv.weight.detach().clone()
                        w[w.shape[0] // 2:, : w.shape[1] // 2] = 0
                        return w
                    # (weight, bias, LeakyReLU after, input channel groups of the masked layer = its canonical summation blocks)
                    L["dense"] = [(masked(m[0]), m[0].bias, True, 2), (masked(m[2]), m[2].bias, True, 2), (m[4].weight[:C2], m[4].bias[:C2], False, 2)]
        else:
            cp = cm.context_prediction
            L["ctx"] = K.MaskedConvPlan(cp.weight, cp.bias, G, G, False)
            if cm.use_param_merger:
                convs = [cm.param_merger_in] + [l for l in cm.param_merger_out if isinstance(l, TopoGroupDynamicMaskConv2d)]
                # param_merger_in: 2G input groups (pgm, -1), channel_group_mask keeps the first G output groups
                # (masked_conv.py:289-292); the following layers see pgm on both sides (:293-296)
                L["m"] = []
                for i, c in enumerate(convs):
                    act = K.ACT_LEAKY_RELU if i + 1 < len(convs) else K.ACT_NONE
                    gi = 2 * G if i == 0 else G
                    L["m"].append((K.MaskedConvPlan(c.weight, c.bias, gi, G, True, act), "cat" if i == 0 else "pgm", "pgm"))
                if G == 1:
                    L["dense"] = [(c.weight, c.bias, i + 1 < len(convs), 2 if i == 0 else 1) for i, c in enumerate(convs)]
        L["ctx_raw"] = (cp.weight, cp.bias)
        return L

    def _split_dynamic_pgm(self, pgm):
        """(topo-group part of the pgm, (weight, bias) of the context convolution or None) -- _preprocess_pgm,
        pgm_coder.py:1300-1339."""
        if not self.pgm_include_dynamic_kernel:
            if isinstance(pgm, (tuple, list)):
                raise ValueError("a (topo groups, kernel weight, kernel bias) pgm needs pgm_include_dynamic_kernel=True")
            return pgm, None
        if pgm is None:       # the reference then codes with its own kernel (:1302-1312)
            return None, None
        topo, w, b = pgm
        cp = self.context_prediction
        w, b = torch.as_tensor(w).detach().to(self.device, torch.float32), torch.as_tensor(b).detach().to(self.device, torch.float32)
        if w.shape[0] != 1 or b.shape[0] != 1:
            raise NotImplementedError("per-sample dynamic kernels")
        if w.numel() != cp.weight.numel() or b.numel() != cp.bias.numel():
            raise NotImplementedError("spatially varying dynamic kernels (a kernel per position)")
        w, b = w.reshape(cp.weight.shape), b.reshape(cp.bias.shape)
        if self.pgm_dynamic_kernel_add_self:
            w, b = w + cp.weight.detach(), b + cp.bias.detach()
        return topo, (w.contiguous(), b.contiguous())

    def _enter_dynamic(self, kernel):
        """Swaps in the layer plans built around a call's own context kernel; returns what _leave_dynamic restores.  The
        plans are kept for the last few kernels seen (by content), so a kernel reused over many calls is packed once."""
        self._ready()
        if kernel is None:
            return None
        import zlib
        key = (zlib.crc32(kernel[0].cpu().numpy().tobytes()), zlib.crc32(kernel[1].cpu().numpy().tobytes()), self._layer_key)
        entry = self._dyn_layers.get(key)
        if entry is None:
            if len(self._dyn_layers) >= 4:
                self._dyn_layers.pop(next(iter(self._dyn_layers)))
            entry = self._dyn_layers[key] = (self._build_layers(ctx_kernel=kernel), {})
        saved = (self._layers, self._graphs)
        self._layers, self._graphs = entry
        return saved

    def _leave_dynamic(self, saved):
        if saved is not None:
            self._layers, self._graphs = saved

    def _topo_from_pgm(self, pgm, h, w) -> np.ndarray:
        """_preprocess_pgm in coding mode (pgm_coder.py:1340-1380, fast_mode=True): logits -> argmax over the last L
        of each channel group's G_c * L channels; a map larger than the latent is trimmed, a smaller one is tiled
        from the top-left corner (F.fold of whole patches: positions beyond the last whole patch stay group 0)."""
        G = self.channel_groups
        t = pgm.detach()
        if t.dim() != 4:
            raise ValueError("pgm must be [1, G_c (* L), h, w]")
        if t.shape[0] != 1:
            raise ValueError("_topo_from_pgm takes ONE map; a batch-sized pgm (per-sample topo groups) is split by _plans_for()")
        if torch.is_floating_point(t):
            if t.shape[1] % G:
                raise ValueError("logits channels must be a multiple of channel_groups")
            t = t.reshape(1, G, t.shape[1] // G, t.shape[2], t.shape[3]).argmax(2)
        if t.shape[1] != G:
            raise ValueError(f"pgm has {t.shape[1]} channel groups, the coder {G}")
        patch = t[0].cpu().numpy().astype(np.int64)[:, :h, :w]
        ph, pw = patch.shape[1:]
        if ph == h and pw == w:
            return patch
        topo = np.zeros((G, h, w), dtype=np.int64)
        for i in range(h // ph):
            for j in range(w // pw):
                topo[:, i * ph:(i + 1) * ph, j * pw:(j + 1) * pw] = patch
        return topo

    def _plan(self, h, w, pgm=None):
        if pgm is None and hasattr(self, "topo_group_predictor_cache"):
            pgm = self.topo_group_predictor_cache              # eval path of _get_pgm (:1551)
        if pgm is None:
            key = (h, w, str(self.device))
            if key not in self._plans:
                topo = default_topo_groups(self.default_topo_group_method, self.channel_groups, h, w)
                self._plans[key] = _GroupPlan(topo, self.in_channels, self.device)
                self._plans[key].key = ("default",)
            return self._plans[key]
        ident = None
        if pgm is getattr(self, "topo_group_predictor_cache", None):   # the module's own buffer: no D2H per call
            ident = ("cache", pgm._version, h, w, str(self.device))
            if ident in self._plans:
                return self._plans[ident]
        topo = self._topo_from_pgm(pgm, h, w)
        key = (h, w, str(self.device), topo.tobytes())
        if key not in self._plans:
            self._plans[key] = _GroupPlan(topo, self.in_channels, self.device)
            self._plans[key].key = ("pgm", sum(1 for k in self._plans if len(k) == 4))
        if ident is not None:
            self._plans[ident] = self._plans[key]
        return self._plans[key]

    def _plans_for(self, h, w, pgm, batch):
        """Per-sample topo groups (pgm_coder.py:1340-1380 keep a pgm's batch dimension; the group masks then differ from image to
        image, :885-890): the list of the images' plans when ``pgm`` carries the batch's leading dimension and the maps really
        differ, else the ONE plan of the call.  Images are independent in the AR loop, so each is coded with its own schedule;
        what the reference fixes is the ORDER of the single stream of a batch -- group-major over all images (data[mask]) --
        which _encode_impl / _decode_impl keep."""
        if torch.is_tensor(pgm) and pgm.dim() == 4 and pgm.shape[0] > 1:
            if pgm.shape[0] != batch:
                raise ValueError(f"pgm has batch {pgm.shape[0]}, the input {batch}")
            plans = [self._plan(h, w, pgm[b:b + 1]) for b in range(batch)]
            return plans[0] if all(pl is plans[0] for pl in plans) else plans
        return self._plan(h, w, pgm)

    # ------------------------------------------------------------------ context model at one group's positions
    def _alloc(self, B, H, W, prior, plan):
        dev, C2 = self.device, self.out_channels
        ws = dict()
        ws["ybuf"] = torch.zeros((B, self.in_channels, H, W), device=dev)
        merger = self._layers.get("m")
        if merger is None:
            ws["ctx"] = torch.zeros((B, C2, H, W), device=dev)
            ws["params"] = torch.empty((B, C2, H, W), device=dev)
        else:
            # cat(ctx, prior) is read by the first merger layer alone: like the hidden activations it lives in the
            # step-contiguous plane order (see _GroupPlan.hidden_order_dev) -- the context layer writes it that way, the prior
            # half is permuted here, once per call
            cat = torch.empty((B, 2 * C2, H, W), device=dev)
            cat[:, :C2].zero_()
            if prior is None:
                cat[:, C2:].zero_()
            elif plan.hidden_order_dev is None:
                cat[:, C2:].copy_(prior)
            else:
                cat[:, C2:].copy_(torch.index_select(prior.reshape(B, C2, H * W), 2, plan.hidden_order_dev).reshape(B, C2, H, W))
            ws["cat"] = cat
            ws["hidden"] = [torch.empty((B, pl.cout, H, W), device=dev) for pl, _, _ in merger]
            ws["params"] = ws["hidden"][-1]
        return ws

    def _context(self, ws, plan, g, B, prior):
        """Parameters of coding step g.  The workspace persists over the steps of one encode / decode call, so each
        masked layer only evaluates the (channel group, position) pairs that belong to step g (csrc/mconv.hip, step
        rule); the reference recomputes every group at every step (pgm_coder.py:922-924)."""
        return self._context_at(ws, plan, plan.positions(g, B, self.device), prior, step=g)

    def _context_at(self, ws, plan, pos, prior, step=None):
        topo = dict(pgm=plan.topo_dev, cat=plan.topo_cat_dev)
        merger = self._layers.get("m")
        sk = dict(step=step, first_step=plan.first_dev) if step is not None else {}
        if merger is None:
            self._layers["ctx"](ws["ybuf"], topo["pgm"], topo["pgm"], pos, ws["ctx"], **sk)
            # no merger: params = ctx + prior (pgm_coder.py:1634-1636); elementwise add on the full map is
            # cheap plumbing and only the group's positions are read afterwards
            ws["params"] = ws["ctx"] + prior if prior is not None else ws["ctx"]
            return ws["params"]
        hp, last = plan.hidden_perm_dev, len(merger) - 1   # cat and the hidden planes in step-contiguous order; the parameters row-major
        self._layers["ctx"](ws["ybuf"], topo["pgm"], topo["pgm"], pos, ws["cat"], out_offset=0, out_perm=hp, **sk)
        x = ws["cat"]
        for i, ((pl, tin, tout), out) in enumerate(zip(merger, ws["hidden"])):
            pl(x, topo[tin], topo[tout], pos, out, in_perm=hp, out_perm=hp if i < last else None, **sk)
            x = out
        return ws["params"]

    # ------------------------------------------------------------------ coding
    def _check_prior(self, shape, prior):
        if prior is not None and self.force_input_prior_shape_aligned:
            assert tuple(shape[2:]) == tuple(prior.shape[2:]), \
                "Input and prior shape not aligned! Consider setting force_input_prior_shape_aligned = False, which may add a little overhead to the bitstream to save the input shape!"
        if prior is not None and not self.force_input_prior_shape_aligned:
            prior = prior[..., : shape[2], : shape[3]]
        return None if prior is None else prior.contiguous()

    def _per_image(self, B):
        mode = self.batch_stream_mode
        if mode == "auto":
            return B > 1
        return mode == "per_image"

    # Many-group patterns (scanline: H*W groups) are launch-bound: ~5 tiny kernels per group.  Their whole
    # per-group launch sequence is captured once per (batch, H, W) into a HIP graph and replayed.
    GRAPH_MIN_GROUPS = 8

    # The scan-line schedule (one coding step per spatial position) with one channel group can run as ONE persistent launch
    # (csrc/scanline.hip) instead of ~6 dependent launches per step.  Both paths sum in the canonical block order of
    # csrc/mconv.hip and code IDENTICAL integers (tests/test_gpu_scanline.py demands equality), so which one serves a call is a
    # matter of speed alone: the encoder and the decoder of a stream need not agree, and nothing about it is recorded in the
    # stream.  The two attributes below are tuning knobs, not part of the format.
    use_persistent_scanline = True
    # batches up to this one take the lane-per-output persistent kernels (measured, scripts/scanline_probe.py, C = 192: batch 1 x2.1
    # in the loop, batch 8 break-even with the per-step path); larger ones -- up to ScanlinePlan.batched_max(): 64 images for layers of
    # BaSIC's shape -- take the batched persistent kernel (round 4: the batch as the N dimension of MFMA tiles, weights in
    # registers); what neither serves goes to the per-step path
    persistent_scanline_max_batch = 4

    def _scanline_plan(self, plan, prior, batch=1, decode=False, width=None):
        """The ScanlinePlan serving this call, or None (then the per-step path codes the same integers): the configuration
        must be the scan-line schedule with dense merger layers that fit the chip's LDS, the batch one a persistent kernel
        serves, and the launch must fit the device (the decoder launch adds one wavefront per image stream and needs the
        table set's fast search image)."""
        if not self.use_persistent_scanline or self.channel_groups != 1 or self.default_topo_group_method != "scanline" \
                or plan.key != ("default",) or "dense" not in self._layers or (batch > 1 and not self._per_image(batch)):
            return None
        C2 = self.out_channels
        pc = 0 if prior is None else prior.shape[1]
        if self._layers["dense"][0][0].shape[1] != C2 + pc:
            return None   # first merger layer expects cat(ctx, prior) of another width (e.g. coding without a prior)
        sp = self._layers.get("scanline")
        if sp is None or sp[1] != pc:
            cw, cb = self._layers["ctx_raw"]
            try:
                sp = (K.ScanlinePlan(cw.detach(), cb.detach() if cb is not None else None,
                                     [(w.detach(), b.detach() if b is not None else None, a, gi) for w, b, a, gi in self._layers["dense"]], pc), pc)
            except ValueError:   # the layers' weights exceed the LDS of the chip (a property of the configuration)
                sp = (None, pc)
            self._layers["scanline"] = sp
        sl = sp[0]
        if sl is None:
            return None
        if batch > self.persistent_scanline_max_batch and (width is None or batch > sl.batched_max(width, decode)):
            return None
        if not (sl.can_decode(self._tables, batch) if decode else sl.can_encode(batch)):
            return None
        return sl

    def _run_encode(self, y, prior, pgm=None):
        self._ready()
        plan = self._plans_for(y.shape[2], y.shape[3], pgm, y.shape[0])
        if isinstance(plan, list):   # per-sample topo groups: every image with its own schedule (symbols in ITS coding order)
            outs = [self._run_encode(y[b:b + 1].contiguous(), None if prior is None else prior[b:b + 1].contiguous(), pgm[b:b + 1])
                    for b in range(y.shape[0])]
            return (torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs]), torch.cat([o[2].clone() for o in outs]), plan)
        sl = self._scanline_plan(plan, prior, y.shape[0], width=y.shape[3])
        if sl is not None:
            sym, idx, ybuf = sl.encode(y, prior, self._scale_table_dev)
            sl.check()   # a launch whose grid never became resident gave up on its barriers: fail loudly, never code garbage
            return sym, idx, ybuf, plan
        if len(plan.groups) < self.GRAPH_MIN_GROUPS or not getattr(self, "use_hip_graphs", True):
            return self._run_encode_impl(y, prior, plan)
        key = ("enc", tuple(y.shape), prior is not None, plan.key)
        entry = self._graphs.get(key)
        if entry is None:
            sy = torch.empty_like(y)
            sp = torch.empty_like(prior) if prior is not None else None
            sy.copy_(y)
            if sp is not None:
                sp.copy_(prior)
            self._run_encode_impl(sy, sp, plan)    # eager warm-up: builds position lists, sets kernel attributes
            graph = torch.cuda.CUDAGraph()
            with _capture(self.device, graph):
                out = self._run_encode_impl(sy, sp, plan)
            entry = self._graphs[key] = (graph, sy, sp, out)
        graph, sy, sp, out = entry
        sy.copy_(y)
        if sp is not None:
            sp.copy_(prior)
        graph.replay()
        return out

    def _run_encode_impl(self, y, prior, plan):
        B, C, H, W = y.shape
        ws = self._alloc(B, H, W, prior, plan)
        n = plan.per_image
        sym = torch.empty((B, n), device=self.device, dtype=torch.int32)
        idx = torch.empty((B, n), device=self.device, dtype=torch.int32)
        L = _lib.lib()
        for g, grp in enumerate(plan.groups):
            if grp["n"] == 0:
                continue
            params = self._context(ws, plan, g, B, prior)
            _lib.check(L.basic_pgm_gauss_encode_group_dev(
                y.data_ptr(), params.data_ptr(), B, C, H * W, grp["elems"].data_ptr(), grp["n"], self._scale_table_dev.data_ptr(),
                self._scale_table_dev.numel(), sym.data_ptr(), idx.data_ptr(), n, grp["base"], ws["ybuf"].data_ptr(),
                K._stream()))
        return sym, idx, ws["ybuf"], plan

    def _quantizer(self, quantizer_params):
        """(offset, step) of this call, or None for the identity (torch_ans.py:105-121: a call's quantizer_params override
        the module's)."""
        qp = self._quantizer_host if quantizer_params is None else \
            [float(v) for v in torch.as_tensor(quantizer_params, dtype=torch.float32).reshape(-1).tolist()]
        off, step = (qp[0], qp[2]) if self.quantizer_type == "uniform" else (0.0, qp[0])
        return None if (off == 0.0 and step == 1.0) else (off, step)

    def _to_coding_domain(self, input, q):     # _data_preprocess(transform=True, quantize=False), same float32 operations
        if q is None:
            return input
        return (input - q[0]) / q[1] if self.quantizer_type == "uniform" else input / q[1]

    def _from_coding_domain(self, out, q):     # _data_postprocess
        if q is None:
            return out
        return out * q[1] + q[0] if self.quantizer_type == "uniform" else out * q[1]

    def forward(self, input, prior=None, pgm=None, quantizer_params=None, **kwargs):
        pgm, kernel = self._split_dynamic_pgm(pgm)
        q = self._quantizer(quantizer_params)
        saved = self._enter_dynamic(kernel)
        try:
            return self._from_coding_domain(self._forward_impl(self._to_coding_domain(input, q), prior=prior, pgm=pgm, **kwargs), q)
        finally:
            self._leave_dynamic(saved)

    def encode(self, input, *args, prior=None, pgm=None, quantizer_params=None, **kwargs) -> bytes:
        pgm, kernel = self._split_dynamic_pgm(pgm)
        q = self._quantizer(quantizer_params)
        saved = self._enter_dynamic(kernel)
        try:
            return self._encode_impl(self._to_coding_domain(input, q), *args, prior=prior, pgm=pgm, **kwargs)
        finally:
            self._leave_dynamic(saved)

    def decode(self, byte_string: bytes, *args, prior=None, pgm=None, quantizer_params=None, **kwargs):
        pgm, kernel = self._split_dynamic_pgm(pgm)
        q = self._quantizer(quantizer_params)
        saved = self._enter_dynamic(kernel)
        try:
            return self._from_coding_domain(self._decode_impl(byte_string, *args, prior=prior, pgm=pgm, **kwargs), q)
        finally:
            self._leave_dynamic(saved)

    def _forward_impl(self, input, prior=None, pgm=None, quantizer_params=None, **kwargs):
        """Eval-mode forward (pgm_coder.py:391-539): returns the dequantised latent round(y).  With
        ``self.estimate_rate = True`` it also evaluates the reference's rate estimate: ONE full-map pass of the
        context model on round(y) (:421-429), likelihood cdf(q+.5) - cdf(q-.5) under N(mu, max(sigma, 0.11)),
        metric_dict["prior_entropy"] = -sum log(max(p, eps)) / batch (:374-389,:520-522)."""
        self._ready()
        input = input.contiguous()
        prior = self._check_prior(input.shape, prior)
        noise = getattr(self, "rate_proxy", "round") == "noise"
        q = input + (torch.rand_like(input) - 0.5) if noise else torch.round(input)
        if getattr(self, "estimate_rate", False):
            B, C, H, W = input.shape
            plan = self._plans_for(H, W, pgm, B)   # logits are taken at their argmax here too (the coding-mode groups)
            if isinstance(plan, list):   # per-sample topo groups: image by image, the mean of their rates
                saved, nlls = self.estimate_rate, []
                for b in range(B):
                    self._forward_impl(input[b:b + 1], prior=None if prior is None else prior[b:b + 1], pgm=pgm[b:b + 1], **kwargs)
                    nlls.append(self.get_raw_cache("metric_dict")["prior_entropy"])
                self.update_cache("metric_dict", prior_entropy=torch.stack([torch.as_tensor(v, device=self.device).float() for v in nlls]).mean())
                return q
            ws = self._alloc(B, H, W, prior, plan)
            ws["ybuf"] = q
            allpos = torch.arange(B * H * W, device=self.device, dtype=torch.int32)
            params = self._context_at(ws, plan, allpos, prior)
            if self.training_no_quantize_for_likelihood:
                # the residual: eval round(y - mu); train-mode proxy y - mu + a SECOND noise draw (pgm_coder.py:383)
                v, mode = (input + (torch.rand_like(input) - 0.5), True) if noise else (input, "round_residual")
            else:
                v, mode = q, True
            nll = K.gauss_nll_per_image(v, params, mode, self._lower_bound_scale, self.eps)
            self.update_cache("metric_dict", prior_entropy=nll.mean())
        return q

    def _encode_impl(self, input, *args, prior=None, pgm=None, quantizer_params=None, **kwargs) -> bytes:
        self._ready()
        input = input.contiguous()
        prior = self._check_prior(input.shape, prior)
        B = input.shape[0]
        sym, idx, _, plan = self._run_encode(input, prior, pgm)
        plans = plan if isinstance(plan, list) else None
        n = (plans[0] if plans else plan).per_image
        if self._per_image(B):
            host, off = self._tables.encode_batch_end(self._tables.encode_batch_begin(sym.reshape(-1), idx.reshape(-1), n))
            lens = (np.diff(off) * 4).astype("<u4")
            body = b"".join([struct.pack("<I", B), lens.tobytes(), memoryview(host[: int(off[-1])])])   # one copy of the words
        else:
            # reference order for a batch: group-major over ALL images (data[mask] spans the batch, :898-900)
            if plans:   # per-sample groups: group g of image 0, of image 1, ...; then group g + 1 (images with fewer groups drop out)
                parts_s, parts_i = [], []
                for g in range(max(len(pl.groups) for pl in plans)):
                    for b, pl in enumerate(plans):
                        if g < len(pl.groups):
                            grp = pl.groups[g]
                            parts_s.append(sym[b, grp["base"]: grp["base"] + grp["n"]])
                            parts_i.append(idx[b, grp["base"]: grp["base"] + grp["n"]])
                sym, idx = torch.cat(parts_s), torch.cat(parts_i)
            elif B > 1:
                parts_s, parts_i = [], []
                for grp in plan.groups:
                    parts_s.append(sym[:, grp["base"]: grp["base"] + grp["n"]].reshape(-1))
                    parts_i.append(idx[:, grp["base"]: grp["base"] + grp["n"]].reshape(-1))
                sym, idx = torch.cat(parts_s), torch.cat(parts_i)
            body = self._tables.encode_batch_to_bytes(sym.reshape(-1).contiguous(), idx.reshape(-1).contiguous(), B * n)[0]
        head = b""
        if self.fixed_input_shape is None and not (self.force_input_prior_shape_aligned and prior is not None):
            spatial = input.shape[2:]                                    # pgm_coder.py:581-596
            head = struct.pack("B", len(spatial) + 1) + struct.pack("<H", B) + b"".join(struct.pack("<H", d) for d in spatial)
        return head + body

    def _decode_impl(self, byte_string: bytes, *args, prior=None, pgm=None, quantizer_params=None, **kwargs):
        self._ready()
        ptr = 0
        if self.fixed_input_shape is not None:
            B, spatial = self.fixed_input_shape[0], tuple(self.fixed_input_shape[1:])
        elif self.force_input_prior_shape_aligned and prior is not None:
            B, spatial = prior.shape[0], tuple(prior.shape[2:])
        else:
            nd = byte_string[0]
            dims = struct.unpack("<%dH" % nd, byte_string[1:1 + 2 * nd])
            ptr = 1 + 2 * nd
            B, spatial = dims[0], tuple(dims[1:])
        H, W = spatial
        prior = self._check_prior((B, self.in_channels, H, W), prior)
        body = byte_string[ptr:]
        plan = self._plans_for(H, W, pgm, B)
        if isinstance(plan, list):
            return self._decode_per_sample(body, prior, B, H, W, plan)
        n, C = plan.per_image, self.in_channels
        per_image = self._per_image(B)
        if per_image:
            (nb,) = struct.unpack("<I", body[:4])
            assert nb == B
            lens = np.frombuffer(body, dtype="<u4", count=B, offset=4).astype(np.int64)
            payload = 4 + 4 * B
        else:
            lens, payload = np.array([len(body)], dtype=np.int64), 0
        if (lens < 8).any() or (lens % 4).any() or payload + int(lens.sum()) > len(body):
            raise ValueError("rANS stream must hold >= 2 whole 32-bit words")
        woff = np.concatenate([[0], np.cumsum(lens // 4)]).astype(np.int64)
        dev = self.device
        # the streams lie back to back in the body: one copy into the pinned staging buffer, DMA from there
        stage = self._tables._stage_in(int(woff[-1]))
        stage.numpy()[:] = np.frombuffer(body, dtype=np.int32, count=int(woff[-1]), offset=payload)
        words_np = stage.numpy()
        sl = self._scanline_plan(plan, prior, B, decode=True, width=W)
        if sl is not None:   # persistent scan-line launch (see _run_encode); one stream per image
            d_words = stage.to(dev, non_blocking=True)
            self._tables._pin_in_event = torch.cuda.Event()
            self._tables._pin_in_event.record(torch.cuda.current_stream(dev))
            d_woff = torch.from_numpy(woff).to(dev)
            _, _, ybuf = sl.decode(self._tables, d_words, d_woff, prior, B, H, W, self._scale_table_dev)
            sl.check()
            return ybuf
        use_graph = len(plan.groups) >= self.GRAPH_MIN_GROUPS and getattr(self, "use_hip_graphs", True) and per_image
        if not use_graph:
            d_words = stage.to(dev, non_blocking=True)
            self._tables._pin_in_event = torch.cuda.Event()
            self._tables._pin_in_event.record(torch.cuda.current_stream(dev))
            d_woff = torch.from_numpy(woff).to(dev)
            return self._run_decode_impl(d_words, d_woff, prior, B, H, W, per_image, plan)
        # static buffers: per-image streams never exceed the encoder's slot bound plus slack
        cap = B * (3 * n + 4)
        key = ("dec", B, H, W, prior is not None, plan.key)
        entry = self._graphs.get(key)
        if entry is None:
            sw = torch.zeros((cap,), device=dev, dtype=torch.int32)
            so = torch.zeros((B + 1,), device=dev, dtype=torch.int64)
            sp = torch.empty_like(prior) if prior is not None else None
            sw[: words_np.size].copy_(stage)
            so.copy_(torch.from_numpy(woff))
            if sp is not None:
                sp.copy_(prior)
            self._run_decode_impl(sw, so, sp, B, H, W, True, plan)   # eager warm-up
            graph = torch.cuda.CUDAGraph()
            with _capture(self.device, graph):
                out = self._run_decode_impl(sw, so, sp, B, H, W, True, plan)
            entry = self._graphs[key] = (graph, sw, so, sp, out)
        graph, sw, so, sp, out = entry
        if words_np.size > cap:
            raise ValueError("encoded stream larger than the decoder's static buffer")
        sw[: words_np.size].copy_(stage, non_blocking=True)
        self._tables._pin_in_event = torch.cuda.Event()
        self._tables._pin_in_event.record(torch.cuda.current_stream(dev))
        so.copy_(torch.from_numpy(woff))
        if sp is not None:
            sp.copy_(prior)
        graph.replay()
        return out.clone()

    def _decode_per_sample(self, body, prior, B, H, W, plans):
        """Decoding with per-sample topo groups.  One stream per image: every image decodes alone with its own schedule.  ONE stream
        for the batch (the reference's layout, data[mask] over all images): the images advance in lock step, group by group --
        parameters and table rows of group g for every image that has one, their symbols off the shared stream, then scatter."""
        dev, C = self.device, self.in_channels
        n = plans[0].per_image
        if self._per_image(B):
            (nb,) = struct.unpack("<I", body[:4])
            assert nb == B
            lens = np.frombuffer(body, dtype="<u4", count=B, offset=4).astype(np.int64)
            off = 4 + 4 * B + np.concatenate([[0], np.cumsum(lens)])
            outs = []
            for b in range(B):
                one = bytes(body[int(off[b]): int(off[b + 1])])
                if len(one) < 8 or len(one) % 4:
                    raise ValueError("rANS stream must hold >= 2 whole 32-bit words")
                outs.append(self._decode_single_stream(one, None if prior is None else prior[b:b + 1].contiguous(), 1, H, W, plans[b]).clone())
            return torch.cat(outs)
        if len(body) < 8 or len(body) % 4:
            raise ValueError("rANS stream must hold >= 2 whole 32-bit words")
        words = torch.from_numpy(np.frombuffer(bytes(body), dtype=np.int32).copy()).to(dev)
        woff = torch.tensor([0, len(body) // 4], device=dev, dtype=torch.int64)
        state = torch.zeros((1,), device=dev, dtype=torch.int64)
        pos = torch.full((1,), -1, device=dev, dtype=torch.int64)
        pri = [None if prior is None else prior[b:b + 1].contiguous() for b in range(B)]
        wss = [self._alloc(1, H, W, pri[b], plans[b]) for b in range(B)]
        sym = torch.empty((B, n), device=dev, dtype=torch.int32)
        idx = torch.empty((B, n), device=dev, dtype=torch.int32)
        L = _lib.lib()
        for g in range(max(len(pl.groups) for pl in plans)):
            live, params = [], {}
            for b, pl in enumerate(plans):
                if g < len(pl.groups) and pl.groups[g]["n"] > 0:
                    grp = pl.groups[g]
                    params[b] = self._context(wss[b], pl, g, 1, pri[b])
                    _lib.check(L.basic_pgm_gauss_index_group_dev(params[b].data_ptr(), 1, C, H * W, grp["elems"].data_ptr(), grp["n"],
                                                                self._scale_table_dev.data_ptr(), self._scale_table_dev.numel(),
                                                                idx[b:b + 1].data_ptr(), n, grp["base"], K._stream()))
                    live.append(b)
            if not live:
                continue
            gi = torch.cat([idx[b, plans[b].groups[g]["base"]: plans[b].groups[g]["base"] + plans[b].groups[g]["n"]] for b in live]).contiguous()
            go = torch.empty_like(gi)
            seg = torch.tensor([0, gi.numel()], device=dev, dtype=torch.int64)
            self._tables.decode_batch(words, woff, gi, seg, out=go, state=state, pos=pos)
            at = 0
            for b in live:
                grp = plans[b].groups[g]
                sym[b, grp["base"]: grp["base"] + grp["n"]] = go[at: at + grp["n"]]
                at += grp["n"]
                _lib.check(L.basic_pgm_gauss_scatter_group_dev(sym[b:b + 1].data_ptr(), params[b].data_ptr(), 1, C, H * W, grp["elems"].data_ptr(), grp["n"],
                                                              n, grp["base"], wss[b]["ybuf"].data_ptr(), K._stream()))
        return torch.cat([ws["ybuf"] for ws in wss])

    def _decode_single_stream(self, stream, prior, B, H, W, plan):
        """One stream (bytes) of a whole call coded with ``plan`` -> the coded latent; the eager per-group loop."""
        dev = self.device
        words = torch.from_numpy(np.frombuffer(stream, dtype=np.int32).copy()).to(dev)
        woff = torch.tensor([0, len(stream) // 4], device=dev, dtype=torch.int64)
        return self._run_decode_impl(words, woff, prior, B, H, W, False, plan)

    def _run_decode_impl(self, d_words, d_woff, prior, B, H, W, per_image, plan):
        dev, C = self.device, self.in_channels
        n = plan.per_image
        ns = B if per_image else 1
        state = torch.zeros((ns,), device=dev, dtype=torch.int64)
        pos = torch.full((ns,), -1, device=dev, dtype=torch.int64)
        ws = self._alloc(B, H, W, prior, plan)
        sym = torch.empty((B, n), device=dev, dtype=torch.int32)
        idx = torch.empty((B, n), device=dev, dtype=torch.int32)
        L = _lib.lib()
        for g, grp in enumerate(plan.groups):
            ng = grp["n"]
            if ng == 0:
                continue
            params = self._context(ws, plan, g, B, prior)
            _lib.check(L.basic_pgm_gauss_index_group_dev(params.data_ptr(), B, C, H * W, grp["elems"].data_ptr(), ng,
                                                        self._scale_table_dev.data_ptr(), self._scale_table_dev.numel(),
                                                        idx.data_ptr(), n, grp["base"], K._stream()))
            if per_image:
                # stream b continues (decode_stream semantics, pgm_coder.py:971) with its ng symbols of this group,
                # in place on the dense [B][n] arrays
                _lib.check(L.basic_rans_decode_batch_strided_dev(self._tables._h, d_words.data_ptr(), d_woff.data_ptr(), idx.data_ptr(),
                                                                 grp["base"], n, ng, B, sym.data_ptr(), state.data_ptr(), pos.data_ptr(),
                                                                 K._stream()))
            else:
                # single stream over the whole batch: gather this group's indexes batch-major
                gi = idx[:, grp["base"]: grp["base"] + ng].reshape(-1).contiguous()
                go = torch.empty_like(gi)
                seg = torch.tensor([0, B * ng], device=dev, dtype=torch.int64)
                self._tables.decode_batch(d_words, d_woff, gi, seg, out=go, state=state, pos=pos)
                sym[:, grp["base"]: grp["base"] + ng] = go.reshape(B, ng)
            _lib.check(L.basic_pgm_gauss_scatter_group_dev(sym.data_ptr(), params.data_ptr(), B, C, H * W, grp["elems"].data_ptr(), ng,
                                                          n, grp["base"], ws["ybuf"].data_ptr(), K._stream()))
        return ws["ybuf"]


class CombinedNNTrainablePGMPriorCoder(HotPathModule):
    """pgm_coder.py:632-715: a bank of prior coders of which ``blend_weight`` (a one-hot from a controller node, e.g.
    BaSIC's ``pgmy``) selects one per call -- the entropy-coder side of BaSIC's complexity scaling (scanline AR down to
    2-stage grouped coders, configs/presets/lossy_latent_graph_scalable_ar_models.py:198-372)."""

    accepts_buffer = True

    def __init__(self, coders, *args, blend_weight_one_hot_threshold=0.9, fix_weight=False, training_use_max_capacity=False,
                 **kwargs):
        super().__init__()
        self.coders = nn.ModuleList(coders)
        if fix_weight:
            self.register_buffer("default_blend_weight", torch.zeros(len(coders)), persistent=False)
        else:
            self.default_blend_weight = nn.Parameter(torch.zeros(len(coders)))
        self.blend_weight_one_hot_threshold = blend_weight_one_hot_threshold
        self.training_use_max_capacity = training_use_max_capacity

    @property
    def estimate_rate(self):
        return all(getattr(c, "estimate_rate", False) for c in self.coders)

    @estimate_rate.setter
    def estimate_rate(self, value):
        for c in self.coders:
            if hasattr(c, "estimate_rate"):
                c.estimate_rate = value

    def _select(self, blend_weight):
        if blend_weight is None:
            blend_weight = torch.softmax(self.default_blend_weight, dim=0)
        if isinstance(blend_weight, int):
            return self.coders[blend_weight]
        return self.coders[int(torch.as_tensor(blend_weight).reshape(-1).argmax().item())]

    def forward(self, input, prior=None, blend_weight=None, **kwargs):
        coder = self._select(blend_weight)
        ret = coder(input, prior=prior, **kwargs)
        pe = coder.get_raw_cache("metric_dict").pop("prior_entropy", None)   # (:697-699)
        if pe is not None:
            self.update_cache("metric_dict", prior_entropy=pe)
        return ret

    def encode(self, input, *args, prior=None, blend_weight=None, **kwargs) -> bytes:
        return self._select(blend_weight).encode(input, prior=prior, **kwargs)

    def decode(self, byte_string: bytes, *args, prior=None, blend_weight=None, **kwargs):
        return self._select(blend_weight).decode(byte_string, prior=prior, **kwargs)

    def update_state(self, *args, **kwargs) -> None:
        for coder in self.coders:
            coder.update_state(*args, **kwargs)
