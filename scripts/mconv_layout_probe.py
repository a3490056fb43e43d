"""Does the merger GEMM care about the position layout?  Same masked 1x1 layer (1536 -> 1536, two channel groups pgm / -1), 32768
positions: (a) every position of 128 images (contiguous lines), (b) the checkerboard half of 256 images (stride-2 positions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K
torch.manual_seed(0)
cin = cout = 1536
w = torch.randn(cout, cin, 1, 1) * 0.02
b = torch.zeros(cout)
plan = K.MaskedConvPlan(w, b, 2, 2, True, K.ACT_LEAKY_RELU)
H = W = 16
for name, B, sel in (("all positions of 128 images", 128, None), ("checkerboard half of 256 images", 256, "cb")):
    x = torch.randn(B, cin, H, W, device="cuda")
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    topo = ((yy + xx) % 2).int() if sel else torch.zeros(H, W, dtype=torch.int32)
    tcat = torch.stack([topo, torch.full_like(topo, -1)]).cuda()
    p = torch.arange(H * W)[(topo.reshape(-1) == 0)] if sel else torch.arange(H * W)
    pos = (torch.arange(B).reshape(-1, 1) * H * W + p.reshape(1, -1)).reshape(-1).int().cuda()
    out = torch.zeros(B, cout, H, W, device="cuda")
    for _ in range(3):
        plan(x, tcat, tcat, pos, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan(x, tcat, tcat, pos, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gf = 2 * 0.75 * cin * cout * pos.numel() / 1e9
    print(f"{name}: {pos.numel()} positions, {ms:.3f} ms, {gf / ms:.1f} TFLOP/s (algorithmic, 3 of 4 blocks open)")
