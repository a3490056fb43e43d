#!/usr/bin/env python3
"""GPU probe: the scan-line AR y-coder (BaSIC context-model coder, C = 192): per-step launch path vs persistent kernel."""
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
shapes = ((1, 32, 48), (8, 16, 16), (64, 16, 16), (24, 32, 48))
if os.environ.get("PROBE_SMALL_BATCHES"):   # where does the per-step path take over?
    shapes = ((2, 32, 48), (3, 32, 48), (4, 32, 48), (6, 32, 48))
if os.environ.get("PROBE_SHAPES"):   # ablation runs (BASIC_SCAN_DEBUG: wrong results, timing only): the batch-1 Kodak shape
    shapes = shapes[:int(os.environ["PROBE_SHAPES"])]
for B, H, W in shapes:
    y = (torch.randn(B, C, H, W, generator=g) * 2).cuda()
    prior = torch.stack([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 3 + 0.1], 2).reshape(B, 2 * C, H, W).cuda()
    res = {}
    for mode in ((True,) if os.environ.get("PROBE_PERSISTENT_ONLY") else (False, True)):
        c.use_persistent_scanline = mode
        c.persistent_scanline_max_batch = 1024
        for what in ("enc", "dec"):
            if what == "enc":
                fn = lambda: c.encode(y, prior=prior)
            else:
                data = c.encode(y, prior=prior)
                fn = lambda: c.decode(data, prior=prior)
            fn(); fn()
            torch.cuda.synchronize()
            t0 = time.time()
            n = 3
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            res[(mode, what)] = (time.time() - t0) / n * 1e3
    res.setdefault((False, 'enc'), float('nan')); res.setdefault((False, 'dec'), float('nan'))
    print(f"B={B:3d} {H}x{W}: per-step enc {res[(False, 'enc')]:8.2f} ms dec {res[(False, 'dec')]:8.2f} ms | persistent enc {res[(True, 'enc')]:8.2f} ms dec {res[(True, 'dec')]:8.2f} ms"
          f"   ({H * W} steps: {res[(True, 'enc')] / (H * W) * 1e3:.1f} us/step enc)", flush=True)
