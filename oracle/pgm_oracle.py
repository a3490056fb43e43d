"""CPU ORACLE of the topo-group autoregressive Gaussian y-coder -- test infrastructure only.

Restates with PyTorch-CPU fp32 + the C rANS oracle:
    topo-group maps          pgm_coder.py:1416-1491 (_get_default_pgm)
    masked convolution       nn/layers/masked_conv.py:102-228 (TopoGroupDynamicMaskConv2d.forward)
    in-coder param merger    pgm_coder.py:1606-1638   ctx-model merger  masked_conv.py:287-305
    Gaussian index / offset  pgm_coder.py:735-821, torch_ans.py:279-282
    table construction       torch_ans.py:284-310
    AR encode / decode loops pgm_coder.py:912-981

Parity status: PINNED by tests/golden/*.npz, generated from the reference's own Python + compiled
csrc (tests/golden/make_golden.py): topo maps, masked-conv outputs, frequency tables, integer
(symbols, indexes) streams and encoded bytes.
"""
import numpy as np
import torch
import torch.distributions as D
import torch.nn.functional as F

from . import rans_oracle as ro
from .codec_oracle import scale_table


def default_pgm(method, G, h, w):
    t = torch.zeros(1, G, h, w, dtype=torch.long)
    if method == "none":
        pass
    elif method == "scanline":
        t = torch.arange(h * w).reshape(1, 1, h, w).repeat(1, G, 1, 1)
    elif method == "zigzag":
        t = (torch.arange(h).reshape(h, 1) + torch.arange(w).reshape(1, w)).reshape(1, 1, h, w).repeat(1, G, 1, 1)
    elif method == "checkerboard":
        t[..., 0::2, 1::2] = 1
        t[..., 1::2, 0::2] = 1
    elif method == "half-checkerboard":
        t.fill_(1)
        t[..., 1::2, 1::2] = 0
    elif method == "halfinv-checkerboard":
        t[..., 1::2, 1::2] = 1
    elif method == "quarter-checkerboard":
        t.fill_(1)
        t[..., 1::4, 3::4] = 0
        t[..., 3::4, 1::4] = 0
    elif method == "interlace-checkerboard":
        for i in range(G):
            if i % 2 == 0:
                t[..., i, 0::2, 0::2] = 1
                t[..., i, 1::2, 1::2] = 1
            else:
                t[..., i, 0::2, 1::2] = 1
                t[..., i, 1::2, 0::2] = 1
    elif method == "raster2x2":
        t[..., 0::2, 1::2] = 1
        t[..., 1::2, 0::2] = 2
        t[..., 1::2, 1::2] = 3
    elif method == "channelwise":
        for i in range(G):
            t[:, i] = i
    elif method == "channelwise-checkerboard":
        for i in range(G):
            t[:, i] = i * 2
            t[:, i, 1::2, 0::2] = i * 2 + 1
            t[:, i, 0::2, 1::2] = i * 2 + 1
    elif method == "channelwise-scanline":
        for i in range(G):
            t[:, i] = torch.arange(h * w).reshape(1, h, w) + i * h * w
    elif method == "channelwise-g10":
        s = 0
        for i, n in enumerate([1] * 9 + [G - 9]):
            t[:, s:s + n] = i
            s += n
    elif method == "elic":
        s = 0
        for i, n in enumerate([1, 1, 2, 4, G - 8]):
            t[:, s:s + n] = i * 2
            t[:, s:s + n, 1::2, 0::2] = i * 2 + 1
            t[:, s:s + n, 0::2, 1::2] = i * 2 + 1
            s += n
    else:
        raise NotImplementedError(method)
    return t


def masked_conv(x, weight, bias, topo, allow_same=False, channel_group_mask=None, out_groups=None):
    """TopoGroupDynamicMaskConv2d.forward for static weights (masked_conv.py:102-228), same tensor ops."""
    B = x.shape[0]
    Cout, Cin, k, _ = weight.shape
    pad = k // 2
    xu = F.unfold(x, (k, k), padding=pad).unsqueeze(1)
    tg = topo.type_as(x)
    off = tg - tg.max().ceil() - 1
    centre = off.reshape(tg.shape[0], tg.shape[1], 1, -1)
    unf = F.unfold(off, (k, k), padding=pad).unsqueeze(1)
    m = (unf <= centre) if allow_same else (unf < centre)
    Gi = tg.shape[1]
    Bt = tg.shape[0]   # topo groups per sample (masked_conv.py:119,171-173) or one map for the batch
    m = m.reshape(Bt, Gi, Gi, k * k, -1).repeat(1, 1, 1, Cin // Gi, 1).reshape(Bt, Gi, Cin * k * k, -1)
    if channel_group_mask is not None:
        m = m[:, channel_group_mask]
    Go = m.shape[1]
    xm = xu * m
    out = weight.reshape(1, Go, Cout // Go, Cin * k * k).matmul(xm)
    if bias is not None:
        out = out + bias.reshape(1, Go, Cout // Go, 1)
    return out.reshape(B, Cout, *x.shape[2:])


def gaussian_ans_params(table, freq_precision=16):
    """torch_ans.py:284-310 for GaussianPGMPriorCoderImpl (zero means, lower-bounded scales)."""
    freq_cnt = 1 << freq_precision
    tail = torch.tensor([0.5 / freq_cnt])
    cnts, nsym, offs = [], [], []
    for s in table:
        dist = D.Normal(torch.zeros(1), torch.max(s.reshape(1), torch.tensor([0.11])))
        lo = int(dist.icdf(tail).floor().item())
        hi = int(dist.icdf(1 - tail).ceil().item())
        offs.append(lo)
        nsym.append(hi - lo + 1)
        pts = torch.arange(lo - 1, hi + 1).float() + 0.5
        lp = (dist.cdf(pts[1:].unsqueeze(0)) - dist.cdf(pts[:-1].unsqueeze(0))).log()[0]
        cnts.append((torch.softmax(lp, -1) * freq_cnt).clamp_min(1).numpy().astype(np.int32))
    f = np.zeros((len(cnts), max(len(c) for c in cnts)), np.int32)
    for i, c in enumerate(cnts):
        f[i, : len(c)] = c
    return f, np.array(nsym, np.int32), np.array(offs, np.int32)


def topo_from_pgm(pgm, G, h, w):
    """Coding-mode _preprocess_pgm (pgm_coder.py:1340-1380, fast_mode=True): float logits [1, G*L, ph, pw] ->
    argmax over L; trim to the latent; tile whole patches with F.fold (stride = patch), so positions past the last
    whole patch are group 0.  Returns int64 [1, G, h, w]."""
    t = torch.as_tensor(pgm)
    if torch.is_floating_point(t):
        t = t.reshape(t.shape[0], G, t.shape[1] // G, *t.shape[2:]).movedim(2, -1).argmax(-1)
    assert t.shape[1] == G   # t.shape[0]: 1, or the batch size (per-sample topo groups)
    t = t[:, :, :h, :w].long()
    ph, pw = t.shape[2:]
    if ph < h or pw < w:
        cols = t.reshape(t.shape[0], -1, 1).repeat(1, 1, (h // ph) * (w // pw)).float()
        t = F.fold(cols, (h, w), (ph, pw), stride=(ph, pw)).long()
    return t


class TopoGroupGaussianOracle:
    """GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder on the CPU, from a state_dict."""

    def __init__(self, sd, in_channels, channel_groups=1, method="none", expand_bottleneck=False, use_param_merger=True,
                 context_model=False, pgm=None, joint_ar=False):
        self.sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.C, self.method = in_channels, method
        self.pgm = pgm  # supplied / learned topo groups (integer map or logits); None = the default pattern
        # use_joint_ar_model_impl (pgm_coder.py:1975-2070): raster-scan order (= the scanline groups), a plain 1x1
        # entropy_parameters network on cat(prior, ctx) (:1619-1620) and "chunk" parameters, scales first (:1047-1048)
        self.joint_ar = joint_ar
        if joint_ar:
            self.method = "scanline"
        self.G = in_channels // 16 if method in ("elic", "channelwise-g10") else channel_groups
        self.use_param_merger, self.context_model = use_param_merger, context_model
        self.table = scale_table()
        self.ans_params = gaussian_ans_params(self.table)
        self.enc, self.dec = ro.Rans64Encoder(16, True, 4), ro.Rans64Decoder(16, True, 4)
        self.enc.init_params(*self.ans_params)
        self.dec.init_params(*self.ans_params)

    def _params(self, buf, pgm, prior):
        sd, G, C2 = self.sd, self.G, 2 * self.C
        B = buf.shape[0]
        if self.context_model:
            p = "topo_group_context_model."
            ctx = masked_conv(buf, sd[p + "context_prediction.weight"], sd[p + "context_prediction.bias"], pgm)
            cat = torch.cat([ctx, prior], 1)
            cat_pgm = torch.cat([pgm, torch.zeros_like(pgm) - 1], 1)
            x = masked_conv(cat, sd[p + "param_merger_in.weight"], sd[p + "param_merger_in.bias"], cat_pgm, True,
                            channel_group_mask=[True] * G + [False] * G)
            i = 1
            while p + f"param_merger_out.{i}.weight" in sd:
                x = masked_conv(F.leaky_relu(x), sd[p + f"param_merger_out.{i}.weight"], sd[p + f"param_merger_out.{i}.bias"], pgm, True)
                i += 2
            return x
        ctx = masked_conv(buf, sd["context_prediction.weight"], sd["context_prediction.bias"], pgm)
        if prior is None:
            prior = torch.zeros_like(ctx)
        if self.joint_ar:
            x = torch.cat([prior, ctx], 1)
            for i in (0, 2, 4):
                x = F.conv2d(x, sd[f"entropy_parameters.{i}.weight"], sd[f"entropy_parameters.{i}.bias"])
                if i < 4:
                    x = F.leaky_relu(x)
            return x
        if not self.use_param_merger:
            return ctx + prior
        cat = torch.cat([ctx, prior], 1)
        cat_pgm = torch.cat([pgm, torch.zeros_like(pgm) - 1], 1)
        x = cat
        for i in (0, 2, 4):
            x = masked_conv(x, sd[f"param_merger.{i}.weight"], sd[f"param_merger.{i}.bias"], cat_pgm, True)
            if i < 4:
                x = F.leaky_relu(x)
        return x.reshape(B, 2 * G, C2 // G, *buf.shape[2:])[:, :G].reshape(B, C2, *buf.shape[2:])

    def _pgm(self, H, W):
        if self.pgm is not None:
            return topo_from_pgm(self.pgm, self.G, H, W)
        return default_pgm(self.method, self.G, H, W)

    def _split(self, params):
        if self.joint_ar:  # chunk + inverse_mean_scale: scales, then means
            half = params.shape[1] // 2
            return params[:, half:], params[:, :half]
        p = params.reshape(params.shape[0], params.shape[1] // 2, 2, *params.shape[2:])
        return p[:, :, 0], p[:, :, 1]  # split_interleave: mean, scale

    def _indexes(self, scales):
        return (scales.reshape(-1).unsqueeze(-1) - self.table.unsqueeze(0)).abs().argmin(-1).reshape_as(scales)

    def masks(self, pgm, shape):
        B, C = shape[0], shape[1]
        full = pgm.unsqueeze(2).repeat(B // pgm.shape[0], 1, C // pgm.shape[1], 1, 1).reshape(shape)
        return [(full == i) for i in range(int(pgm.max()) + 1)]

    def forward_entropy(self, y, prior, residual=False):
        """Eval forward's rate estimate (pgm_coder.py:391-429,374-389,520-522): nats per image.  residual: the coder was
        built with training_no_quantize_for_likelihood -- likelihood of round(y - mu) under the zero-mean density (:376-387)."""
        B, C, H, W = y.shape
        pgm = self._pgm(H, W)
        q = torch.round(y)
        mean, scale = self._split(self._params(q, pgm, prior))
        if residual:
            dist, q = D.Normal(torch.zeros_like(mean), torch.max(scale, torch.tensor([0.11]))), torch.round(y - mean)
        else:
            dist = D.Normal(mean, torch.max(scale, torch.tensor([0.11])))
        lik = dist.cdf(q + 0.5) - dist.cdf(q - 0.5)
        return (-torch.log(torch.max(lik, torch.tensor([1e-7])))).sum() / B

    def encode(self, y, prior):
        B, C, H, W = y.shape
        pgm = self._pgm(H, W)
        buf = torch.zeros_like(y)
        syms, idxs = [], []
        for mask in self.masks(pgm, y.shape):
            mean, scale = self._split(self._params(buf, pgm, prior))
            idx = self._indexes(scale)[mask]
            mu = mean[mask]
            q = torch.round(y[mask] - mu)
            syms.append(q)
            idxs.append(idx)
            buf[mask] = q + mu
        sym = torch.cat(syms).numpy().astype(np.int32)
        idx = torch.cat(idxs).numpy().astype(np.int32)
        return self.enc.encode_with_indexes(sym, idx), sym, idx, buf

    def decode(self, data, prior, shape):
        B, C, H, W = shape
        pgm = self._pgm(H, W)
        buf = torch.zeros(shape)
        self.dec.set_stream(data)
        for mask in self.masks(pgm, shape):
            mean, scale = self._split(self._params(buf, pgm, prior))
            idx = self._indexes(scale)[mask].numpy().astype(np.int32)
            sym = self.dec.decode_stream(idx)
            buf[mask] = torch.from_numpy(sym).float() + mean[mask]
        return buf
