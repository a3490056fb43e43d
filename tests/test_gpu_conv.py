"""GPU parity: MFMA implicit-GEMM conv / deconv / GDN kernels vs a plain PyTorch fp32 CPU
reference of the same op (tolerance 1e-4 relative to the output scale, BASELINE.json north_star)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _close(got, ref):
    scale = max(1.0, float(ref.abs().max()))
    err = float((got.cpu() - ref).abs().max())
    assert err <= TOL * scale, f"max abs err {err} vs scale {scale}"


def _gdn_ref(x, gamma, beta, inverse):
    C = x.shape[1]
    norm = F.conv2d(x * x, gamma.reshape(C, C, 1, 1), beta)
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


CASES = [
    # cin, cout, k, s, p, op, transposed, act, B, H, W
    (3, 128, 5, 2, 2, 0, False, "gdn", 2, 64, 64),
    (128, 128, 5, 2, 2, 0, False, "gdn", 1, 32, 48),
    (128, 192, 5, 2, 2, 0, False, "none", 2, 16, 16),
    (192, 128, 3, 1, 1, 0, False, "relu", 3, 16, 16),
    (128, 128, 5, 2, 2, 0, False, "leaky", 5, 8, 8),
    (128, 128, 5, 2, 2, 0, False, "none", 9, 4, 4),
    (192, 128, 5, 2, 2, 1, True, "igdn", 2, 16, 16),
    (128, 128, 5, 2, 2, 1, True, "igdn", 1, 24, 40),
    (128, 3, 5, 2, 2, 1, True, "none", 2, 32, 32),
    (128, 128, 5, 2, 2, 1, True, "relu", 3, 4, 4),
    (128, 192, 5, 2, 2, 1, True, "leaky", 2, 8, 8),
    (288, 384, 3, 1, 1, 0, False, "none", 2, 16, 16),
    (5, 7, 3, 1, 1, 0, False, "none", 1, 5, 7),       # ragged everything
    (48, 48, 5, 2, 2, 0, False, "gdn", 1, 18, 22),     # slimmable width, odd sizes
    (72, 96, 5, 2, 2, 1, True, "igdn", 1, 7, 9),
    (20, 3, 5, 2, 2, 1, True, "none", 2, 9, 13),       # VALU small-Cout path, ragged tile
    (37, 4, 5, 2, 2, 1, True, "relu", 1, 33, 17),
    (128, 1, 5, 2, 2, 1, True, "leaky", 1, 16, 16),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_matches_torch(case):
    from cbench_basic_amd.nn import kernels as K
    cin, cout, k, s, p, op, tr, act, B, H, W = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % (2 ** 31))   # reproducible (hash() of a str is salted per process)
    x = torch.randn(B, cin, H, W, generator=g)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(wshape, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    gamma = beta = None
    if act in ("gdn", "igdn"):
        gamma = torch.rand(cout, cout, generator=g) * 0.02 + 0.1 * torch.eye(cout)
        beta = torch.rand(cout, generator=g) + 0.5
    ref = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "leaky":
        ref = F.leaky_relu(ref)
    elif act in ("gdn", "igdn"):
        ref = _gdn_ref(ref, gamma, beta, act == "igdn")
    code = dict(none=K.ACT_NONE, relu=K.ACT_RELU, leaky=K.ACT_LEAKY_RELU, gdn=K.ACT_GDN, igdn=K.ACT_IGDN)[act]
    plan = K.ConvPlan(w, b, s, p, op, tr, code, gamma, beta)
    got = plan(x.cuda())
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    _close(got, ref)


@pytest.mark.parametrize("case", [c for c in CASES if c[6] and c[1] > 4], ids=lambda c: str(c))
def test_transposed_conv_unfused_and_sliced_paths(case, monkeypatch):
    """The transposed convolutions normally run with their column phases fused (8-byte pair stores); the four-phase
    fallback (BASIC_CONV_DEBUG bit 512: what runs when the output rows are not 8-byte aligned) and the forced / forbidden
    32-channel-slice variants (bits 4 / 8) must give the same results."""
    from cbench_basic_amd.nn import kernels as K
    cin, cout, k, s, p, op, tr, act, B, H, W = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn((cin, cout, k, k), generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    gamma = beta = None
    if act == "igdn":
        gamma = torch.rand(cout, cout, generator=g) * 0.02 + 0.1 * torch.eye(cout)
        beta = torch.rand(cout, generator=g) + 0.5
    ref = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)
    ref = F.relu(ref) if act == "relu" else F.leaky_relu(ref) if act == "leaky" else _gdn_ref(ref, gamma, beta, True) if act == "igdn" else ref
    code = dict(none=K.ACT_NONE, relu=K.ACT_RELU, leaky=K.ACT_LEAKY_RELU, igdn=K.ACT_IGDN)[act]
    plan = K.ConvPlan(w, b, s, p, op, tr, code, gamma, beta)
    for debug in ("512", "516", "520", "4", "8"):
        monkeypatch.setenv("BASIC_CONV_DEBUG", debug)
        got = plan(x.cuda())
        torch.cuda.synchronize()
        _close(got, ref)


def _random_case(rng):
    """Random layer geometry, biased towards the kernels' special paths (8-wave 25-tap 128-channel convs, the persistent
    first layer, the VALU output layer, 32-channel slices on tiny grids, runtime tap tables for k in {1, 2, 4})."""
    kind = rng.integers(0, 8)
    tr = bool(rng.integers(0, 2))
    k = int(rng.choice([1, 2, 3, 4, 5]))
    s = int(rng.choice([1, 2]))
    p = int(rng.integers(0, k // 2 + 1))
    op = int(rng.integers(0, s)) if tr else 0
    cin, cout = int(rng.integers(1, 200)), int(rng.integers(1, 193))
    act = str(rng.choice(["none", "relu", "leaky", "gdn", "igdn"]))
    B, H, W = int(rng.integers(1, 6)), int(rng.integers(3, 40)), int(rng.integers(3, 40))
    if kind == 0:    # 8-wave path
        tr, k, s, p, op, cout = False, 5, 2, 2, 0, int(rng.integers(97, 129))
    elif kind == 1:  # persistent first layer
        tr, k, s, p, op, cin, cout, act = False, 5, int(rng.choice([1, 2])), 2, 0, int(rng.integers(1, 5)), 128, "gdn"
        H, W = int(rng.integers(20, 70)), int(rng.integers(20, 70))
    elif kind == 2:  # output layer
        tr, k, s, p, op, cout, act = True, 5, 2, 2, 1, int(rng.integers(1, 4)), str(rng.choice(["none", "relu"]))
        W = int(rng.choice([8, 12, 16, 64, 68, 9, 13]))
    elif kind == 3:  # codec deconv phases
        tr, k, s, p, op = True, 5, 2, 2, 1
    if act in ("gdn", "igdn") and cout > 192:
        act = "none"
    if tr and op >= s:
        op = 0
    oh = (H - 1) * s - 2 * p + k + op if tr else (H + 2 * p - k) // s + 1
    ow = (W - 1) * s - 2 * p + k + op if tr else (W + 2 * p - k) // s + 1
    if oh < 1 or ow < 1:
        return None
    return (cin, cout, k, s, p, op, tr, act, B, H, W)


@pytest.mark.parametrize("seed", range(160))
def test_conv_fuzz(seed):
    rng = np.random.default_rng(1000 + seed)
    case = None
    while case is None:
        case = _random_case(rng)
    test_conv_matches_torch(case)


def test_slimmable_weight_slicing():
    """DynamicConv2d semantics (slimmable_layers.py:142-170): W[:co,:ci], b[:co]."""
    from cbench_basic_amd.nn import kernels as K
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 72, 12, 12, generator=g)
    w = torch.randn(192, 192, 5, 5, generator=g) * 0.02
    b = torch.randn(192, generator=g) * 0.1
    ref = F.conv2d(x, w[:96, :72], b[:96], stride=2, padding=2)
    plan = K.ConvPlan(w, b, 2, 2, cin_active=72, cout_active=96)
    _close(plan(x.cuda()), ref)


def _masked_conv_ref(x, w, b, topo_in, topo_out, allow_same):
    """Restates TopoGroupDynamicMaskConv2d.forward (masked_conv.py:102-228) with explicit loops over groups."""
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Gi, Go = topo_in.shape[0], topo_out.shape[0]
    pad = k // 2
    xu = F.unfold(x, k, padding=pad).reshape(B, Cin, k * k, H * W)
    big = 10 ** 6
    tu = F.unfold(topo_in.float().unsqueeze(0) - big, k, padding=pad).reshape(Gi, k * k, H * W) + big  # padded -> huge
    out = torch.zeros(B, Cout, H * W)
    gs_i, gs_o = Cin // Gi, Cout // Go
    for go in range(Go):
        c = topo_out[go].reshape(1, 1, H * W).float()
        m = (tu <= c) if allow_same else (tu < c)  # [Gi, kk, HW]
        m = m & (tu < big / 2)
        mm = m.unsqueeze(1).repeat(1, gs_i, 1, 1).reshape(1, Cin, k * k, H * W).float()
        wg = w[go * gs_o:(go + 1) * gs_o].reshape(gs_o, Cin * k * k)
        out[:, go * gs_o:(go + 1) * gs_o] = torch.matmul(wg, (xu * mm).reshape(B, Cin * k * k, H * W))
    return (out + b.reshape(1, Cout, 1)).reshape(B, Cout, H, W)


@pytest.mark.parametrize("cfg", [(192, 384, 5, 1, 1, False), (192, 384, 5, 4, 4, False), (768, 640, 1, 2, 1, True),
                                 (768, 1536, 1, 24, 24, True), (96, 192, 5, 12, 12, False), (40, 24, 3, 2, 2, True)])
def test_masked_conv_positions(cfg):
    from cbench_basic_amd.nn import kernels as K
    cin, cout, k, gi, go, same = cfg
    g = torch.Generator().manual_seed(cin * 7 + cout)
    B, H, W = 2, 6, 7
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    topo_in = torch.randint(-1, 5, (gi, H, W), generator=g)
    topo_out = torch.randint(0, 5, (go, H, W), generator=g)
    ref = _masked_conv_ref(x, w, b, topo_in, topo_out, same)
    plan = K.MaskedConvPlan(w, b, gi, go, same)
    sel = torch.randperm(B * H * W, generator=g)[: B * H * W - 9].sort().values.int()
    out = torch.full((B, cout + 8, H, W), -7.0).cuda()
    plan(x.cuda(), topo_in.int().cuda(), topo_out.int().cuda(), sel.cuda(), out, out_offset=8)
    torch.cuda.synchronize()
    out = out.cpu()
    mask = torch.zeros(B * H * W, dtype=torch.bool)
    mask[sel.long()] = True
    mask = mask.reshape(B, 1, H, W)
    assert torch.all(out[:, :8] == -7.0)
    got = out[:, 8:]
    assert torch.all(got[(~mask).expand_as(got)] == -7.0), "positions outside the list were touched"
    err = ((got - ref).abs() * mask).max()
    assert err <= TOL * max(1.0, float(ref.abs().max())), float(err)


def _run_mconv_variants(plans, x, topo_in, topo_out, sel, cout, off, variants, **call):
    """The same launch through several (plan, BASIC_MCONV_KERNEL) variants; returns the outputs on the host."""
    import os
    B, _, H, W = x.shape
    outs = []
    for pl, kernel in variants:
        os.environ["BASIC_MCONV_KERNEL"] = kernel[:3] if kernel.startswith("dma") else kernel
        if kernel in ("dma4", "dma8"):   # the two workgroup shapes of the LDS-DMA kernel
            os.environ["BASIC_MCONV_DMA_WAVES"] = kernel[3]
        try:
            out = torch.full((B, cout + off, H, W), -7.0).cuda()
            plans[pl](x.cuda(), topo_in.int().cuda(), topo_out.int().cuda(), sel.cuda(), out, out_offset=off, **call)
            torch.cuda.synchronize()
        finally:
            del os.environ["BASIC_MCONV_KERNEL"]
            os.environ.pop("BASIC_MCONV_DMA_WAVES", None)
        outs.append(out.cpu())
    return outs


@pytest.mark.parametrize("seed", range(40))
def test_masked_conv_fuzz(seed):
    """Random channel counts / group structures / map sizes / position lists for the topo-group masked conv.  Every case runs
    through ALL kernels (BASIC_MCONV_KERNEL: the register-gather kernel with the plan's row tiles per wave and with one tile
    per wave, the block-parallel kernel + reduce of tiny launches, and -- where the layer qualifies -- the LDS-DMA kernel of
    GEMM-shaped launches).  They all sum in the canonical order (csrc/mconv.hip), so their outputs must be IDENTICAL, bit for
    bit: which kernel serves a launch may depend on the batch, the integers the coder derives from the sums may not."""
    import os
    from cbench_basic_amd.nn import kernels as K
    rng = np.random.default_rng(900 + seed)
    gi, go = int(rng.choice([1, 2, 3, 4, 6])), int(rng.choice([1, 2, 3, 4, 6]))
    cin, cout = gi * int(rng.integers(1, 40)), go * int(rng.integers(1, 40))
    if seed % 2:  # whole 32-row tiles per group: 1..6 tiles -> the 1/2/3/4-tiles-per-wave variants
        cout = go * 32 * int(rng.integers(1, 7))
    if seed % 4 == 3:  # a layer the LDS-DMA kernel takes: 128-row chunks, 32-channel stages (blocks of 64 and a 32 / 96 remainder)
        gi, go = int(rng.choice([1, 2])), int(rng.choice([1, 2]))
        cin, cout = gi * 32 * int(rng.integers(1, 6)), go * 128 * int(rng.integers(1, 3))
    k = int(rng.choice([1, 3, 5]))
    same = bool(rng.integers(0, 2))
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(2, 20)), int(rng.integers(2, 20))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    topo_in = torch.randint(-1, 5, (gi, H, W), generator=g)
    topo_out = torch.randint(0, 5, (go, H, W), generator=g)
    ref = _masked_conv_ref(x, w, b, topo_in, topo_out, same)
    plan = K.MaskedConvPlan(w, b, gi, go, same)
    os.environ["BASIC_MCONV_MAX_MT"] = "1"
    try:
        plan_mt1 = K.MaskedConvPlan(w, b, gi, go, same)
    finally:
        del os.environ["BASIC_MCONV_MAX_MT"]
    npos = int(rng.integers(1, B * H * W + 1))
    sel = torch.randperm(B * H * W, generator=g)[:npos].sort().values.int()
    off = int(rng.choice([0, 3]))
    outs = _run_mconv_variants(dict(p=plan, p1=plan_mt1), x, topo_in, topo_out, sel, cout, off,
                               [("p", "gather"), ("p1", "gather"), ("p", "block"), ("p", "dma8"), ("p", "dma4")])
    for i in (1, 2, 3, 4):
        assert torch.equal(outs[0], outs[i]), f"kernel variant {i} differs from the gather kernel"
    mask = torch.zeros(B * H * W, dtype=torch.bool)
    mask[sel.long()] = True
    mask = mask.reshape(B, 1, H, W)
    got = outs[0][:, off:]
    assert torch.all(outs[0][:, :off] == -7.0)
    assert torch.all(got[(~mask).expand_as(got)] == -7.0), "positions outside the list were touched"
    err = ((got - ref).abs() * mask).max()
    assert err <= TOL * max(1.0, float(ref.abs().max())), float(err)


@pytest.mark.parametrize("case", ["merger", "context", "step"])
def test_masked_conv_codec_sized_layers_all_kernels_identical(case):
    """The layers of the 192-channel codecs at a GEMM-shaped size (8 images 16 x 16, checkerboard positions): the 768 -> 1536
    merger layer (2 x 2 channel groups, the prior half masked for the context rows), the 5 x 5 context convolution, and the
    coding-loop step rule with four channel groups -- LDS-DMA kernel == gather kernel == block kernel, bit for bit, and equal
    to the fp32 reference within 1e-4."""
    from cbench_basic_amd.nn import kernels as K
    g = torch.Generator().manual_seed(7)
    B, H, W = 8, 16, 16
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cb = ((yy + xx) % 2).int()
    call = {}
    if case == "merger":
        cin, cout, k, gi, go, same = 768, 1536, 1, 2, 2, True
        topo_in = torch.stack([cb, torch.full_like(cb, -1)])
        topo_out = topo_in.clone()
    elif case == "context":
        cin, cout, k, gi, go, same = 192, 384, 5, 1, 1, False
        topo_in = cb[None].clone()
        topo_out = cb[None].clone()
    else:
        cin, cout, k, gi, go, same = 256, 512, 3, 4, 4, False
        topo_in = torch.stack([cb + 2 * i for i in range(4)])
        topo_out = topo_in.clone()
        call = dict(step=3, first_step=topo_in.min(0).values.int().cuda())
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    plan = K.MaskedConvPlan(w, b, gi, go, same, K.ACT_LEAKY_RELU)
    sel = torch.nonzero((cb.reshape(-1) == 1)[None].expand(B, -1).reshape(-1)).reshape(-1).int()
    if case == "merger":   # the coder's layout of this layer: step-contiguous planes on both sides (16-byte B pieces in the DMA kernel)
        perm = torch.empty(H * W, dtype=torch.int32)
        perm[torch.argsort(1 - cb.reshape(-1), stable=True)] = torch.arange(H * W, dtype=torch.int32)
        call = dict(in_perm=perm.cuda(), out_perm=perm.cuda())
        xp = torch.empty_like(x).reshape(B, cin, H * W)
        xp[:, :, perm.long()] = x.reshape(B, cin, H * W)
        x_in = xp.reshape(B, cin, H, W)
    else:
        perm, x_in = None, x
    outs = _run_mconv_variants(dict(p=plan), x_in, topo_in, topo_out, sel, cout, 0,
                               [("p", "gather"), ("p", "dma8"), ("p", "block"), ("p", "dma4")], **call)
    assert torch.equal(outs[0], outs[1]), "LDS-DMA kernel (8 waves) differs from the gather kernel"
    assert torch.equal(outs[0], outs[2]), "block kernel differs from the gather kernel"
    assert torch.equal(outs[0], outs[3]), "LDS-DMA kernel (4 waves) differs from the gather kernel"
    if perm is not None:   # back to row-major planes for the comparison with the reference
        outs = [o.reshape(B, cout, H * W)[:, :, perm.long()].reshape(B, cout, H, W) for o in outs]
    if case != "step":
        ref = torch.nn.functional.leaky_relu(_masked_conv_ref(x, w, b, topo_in, topo_out, same), 0.01)
        mask = torch.zeros(B * H * W, dtype=torch.bool)
        mask[sel.long()] = True
        mask = mask.reshape(B, 1, H, W)
        err = ((outs[1] - ref).abs() * mask).max()
        assert err <= TOL * max(1.0, float(ref.abs().max())), float(err)
    else:   # only (group, position) pairs whose id equals the step were written
        wrote = (outs[1] != -7.0).reshape(B, 4, cout // 4, H, W).any(2)
        assert torch.equal(wrote, ((topo_out == 3)[None] & (cb == 1)[None, None]).expand(B, -1, -1, -1))


def test_entropy_param_kernels():
    from cbench_basic_amd.nn import kernels as K
    g = torch.Generator().manual_seed(0)
    table = torch.exp(torch.linspace(np.log(0.11), np.log(256), 64))
    y = torch.randn(2, 192, 16, 16, generator=g) * 5
    y.view(-1)[:8] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 3.5, -2.5, 0.49999997])
    s = torch.rand(2, 192, 16, 16, generator=g) * 30
    s.view(-1)[:70] = torch.cat([table, torch.tensor([0.0, 0.05, 0.11, 300.0, 256.0, 1e-9])])
    sym, idx, yhat = K.gc_quantize_index(y.cuda(), s.cuda(), table.cuda())
    sb = torch.max(s, torch.tensor(0.11))
    ref_idx = torch.full(s.shape, 63, dtype=torch.int32)
    for t in table[:-1]:
        ref_idx -= (sb <= t).int()
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(sym.cpu(), torch.round(y).int())
    assert torch.equal(yhat.cpu(), torch.round(y))
    # the index search is a bisection on non-decreasing tables and the reference's counting loop otherwise: table entries hit
    # exactly, repeated entries, values outside the table, an unsorted table, NaN scales
    for tab in (table, torch.tensor([0.2, 0.5, 0.5, 0.5, 1.0, 3.0]), torch.tensor([1.0, 0.3, 2.0, 0.7]), torch.tensor([0.8])):
        s2 = torch.cat([tab, tab * 0.999, tab * 1.001, torch.tensor([0.0, 1e9, float("nan"), 0.11]), s.flatten()[:50]])
        _, i2, _ = K.gc_quantize_index(torch.zeros_like(s2).cuda(), s2.cuda(), tab.cuda())
        sb = torch.maximum(s2, torch.tensor(0.11))   # LowerBound (NaN stays NaN in torch; fmaxf drops it)
        sb = torch.where(torch.isnan(s2), torch.tensor(0.11), sb)
        want = (len(tab) - 1) - (sb.unsqueeze(1) <= tab[:-1]).sum(1)
        assert torch.equal(i2.cpu(), want.int()), tab
    z = torch.randn(3, 128, 4, 4, generator=g) * 3
    med = torch.randn(128, generator=g)
    sym, idx, zhat = K.eb_quantize_index(z.cuda(), med.cuda())
    m4 = med.reshape(1, -1, 1, 1)
    assert torch.equal(sym.cpu(), torch.round(z - m4).int())
    assert torch.equal(zhat.cpu(), torch.round(z - m4) + m4)
    assert torch.equal(idx.cpu(), torch.arange(128).reshape(1, -1, 1, 1).expand(3, 128, 4, 4).int())
    assert torch.equal(K.eb_dequantize(sym, med.cuda()).cpu(), zhat.cpu())
    a, b = torch.rand(4, 3, 32, 32, generator=g), torch.rand(4, 3, 32, 32, generator=g)
    mse = K.mse_per_image(a.cuda(), b.cuda()).cpu()
    assert torch.allclose(mse, ((a - b) ** 2).reshape(4, -1).mean(1), rtol=1e-5)
