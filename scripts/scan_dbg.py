import os, sys, time, torch
sys.path.insert(0, ".")
from cbench_basic_amd.modules.prior_model.prior_coder.pgm_coder import (GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder as Coder, TopoGroupDynamicMaskConv2dContextModel as Ctx)
C = 192
c = Coder(in_channels=C, default_topo_group_method="scanline", topo_group_context_model=Ctx(in_channels=C, out_channels=2 * C))
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p in c.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() > 1 else 0.02))
c = c.eval().cuda(); c.update_state(); c.persistent_scanline_max_batch = 1024
for B, H, W in ((1, 32, 48), (8, 16, 16)):
    y = (torch.randn(B, C, H, W, generator=g) * 2).cuda()
    prior = torch.stack([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 3 + 0.1], 2).reshape(B, 2 * C, H, W).cuda()
    fn = lambda: c._run_encode(y, prior)
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"debug={os.environ.get('BASIC_SCAN_DEBUG','0')} B={B} {H}x{W}: kernel loop {dt*1e3:.2f} ms = {dt/(H*W)*1e6:.1f} us/step", flush=True)
