#!/usr/bin/env python3
"""Micro-benchmark of single transform layers (HIP events), for kernel tuning."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K

def bench(name, cin, cout, k, s, tr, act, B, H, W, reps=5):
    g = torch.Generator().manual_seed(0)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * 0.02
    b = torch.randn(cout, generator=g)
    gamma = beta = None
    if act in (K.ACT_GDN, K.ACT_IGDN):
        gamma = torch.rand(cout, cout, generator=g) * 0.01 + 0.1 * torch.eye(cout); beta = torch.ones(cout)
    plan = K.ConvPlan(w, b, s, k // 2, s - 1 if tr else 0, tr, act, gamma, beta)
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    y = plan(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): plan(x, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = plan.flops(B, H, W)
    print(f"{name:28s} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  ({100*fl/ms/1e9/157.3:5.1f}% of fp32 MFMA peak)")

B = int(os.environ.get("B", "256"))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
L = [("g_a.1 3->128 s2 gdn @256", 3, 128, 5, 2, False, K.ACT_GDN, B, 256, 256),
     ("g_a.2 128->128 s2 gdn @128", 128, 128, 5, 2, False, K.ACT_GDN, B, 128, 128),
     ("g_a.3 128->128 s2 gdn @64", 128, 128, 5, 2, False, K.ACT_GDN, B, 64, 64),
     ("g_a.4 128->192 s2 @32", 128, 192, 5, 2, False, K.ACT_NONE, B, 32, 32),
     ("h_a.1 192->128 k3 relu @16", 192, 128, 3, 1, False, K.ACT_RELU, B, 16, 16),
     ("h_a.2 128->128 s2 relu @16", 128, 128, 5, 2, False, K.ACT_RELU, B, 16, 16),
     ("h_a.3 128->128 s2 @8", 128, 128, 5, 2, False, K.ACT_NONE, B, 8, 8),
     ("h_s.1 128->128 T relu @4", 128, 128, 5, 2, True, K.ACT_RELU, B, 4, 4),
     ("h_s.2 128->128 T relu @8", 128, 128, 5, 2, True, K.ACT_RELU, B, 8, 8),
     ("h_s.3 128->192 k3 relu @16", 128, 192, 3, 1, False, K.ACT_RELU, B, 16, 16),
     ("g_s.1 192->128 T igdn @16", 192, 128, 5, 2, True, K.ACT_IGDN, B, 16, 16),
     ("g_s.2 128->128 T igdn @32", 128, 128, 5, 2, True, K.ACT_IGDN, B, 32, 32),
     ("g_s.3 128->128 T igdn @64", 128, 128, 5, 2, True, K.ACT_IGDN, B, 64, 64),
     ("g_s.4 128->3 T @128", 128, 3, 5, 2, True, K.ACT_NONE, B, 128, 128),
     # ablations (not layers of the codec): the two big layers without their GDN epilogue
     ("x_a.2 128->128 s2 relu @128", 128, 128, 5, 2, False, K.ACT_RELU, B, 128, 128),
     ("x_s.3 128->128 T relu @64", 128, 128, 5, 2, True, K.ACT_RELU, B, 64, 64),
     ("y_a.2 512->128 s2 relu @128 (4x longer K)", 512, 128, 5, 2, False, K.ACT_RELU, B // 4, 128, 128)]
for l in L:
    if (which == "all" and not l[0].startswith("x_")) or which in l[0]:
        bench(*l)
