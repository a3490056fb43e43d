#!/usr/bin/env python3
"""GPU probe: the batched persistent scan-line kernel alone (HIP-event time of the launch), BaSIC context-model coder, C = 192."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.modules.prior_model.prior_coder.pgm_coder import (GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder as Coder,
                                                                        TopoGroupDynamicMaskConv2dContextModel as Ctx)
C = 192
c = Coder(in_channels=C, default_topo_group_method="scanline", topo_group_context_model=Ctx(in_channels=C, out_channels=2 * C))
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p in c.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() > 1 else 0.02))
c = c.eval().cuda()
c.update_state()
c.persistent_scanline_max_batch = 1024
c._ready()
shapes = [(int(a), int(b), int(d)) for a, b, d in (s.split("x") for s in os.environ.get("PROBE", "8x16x16,32x16x16,64x16x16,3x16x16").split(","))]
for B, H, W in shapes:
    y = (torch.randn(B, C, H, W, generator=g) * 2).cuda()
    prior = torch.stack([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 3 + 0.1], 2).reshape(B, 2 * C, H, W).cuda()
    plan = c._plan(H, W, None)
    sl = c._scanline_plan(plan, prior, B, width=W)
    tab = c._scale_table_dev
    for _ in range(3):
        sl.encode(y, prior, tab)
    sl.check()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 5
    ev[0].record()
    for _ in range(n):
        sl.encode(y, prior, tab)
    ev[1].record()
    torch.cuda.synchronize()
    enc = ev[0].elapsed_time(ev[1]) / n
    # decode: through the coder (stream upload + launch), kernel time from rocprofv3
    data = c.encode(y, prior=prior)
    for _ in range(2):
        c.decode(data, prior=prior)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        c.decode(data, prior=prior)
    torch.cuda.synchronize()
    dec = (time.time() - t0) / n * 1e3
    flops = 2.0 * B * H * W * 1.9e6
    print(f"B={B:3d} {H}x{W}: encode launch (+ prior transpose, memset) {enc:7.3f} ms = {enc / (H * W) * 1e3:6.2f} us/step, {flops / enc / 1e9:6.2f} TFLOP/s | decode call {dec:7.3f} ms = {dec / (H * W) * 1e3:6.2f} us/step", flush=True)
