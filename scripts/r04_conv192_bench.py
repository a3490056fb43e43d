#!/usr/bin/env python3
"""BaSIC level-0 transform layers (192 channels everywhere) alone, 64 images: efficiency of the 6-tile convolution kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K

def bench(name, cin, cout, k, s, tr, act, B, H, W, reps=10):
    g = torch.Generator().manual_seed(0)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * 0.02
    b = torch.randn(cout, generator=g)
    gamma = beta = None
    if act in (K.ACT_GDN, K.ACT_IGDN):
        gamma = torch.rand(cout, cout, generator=g) * 0.01 + 0.1 * torch.eye(cout); beta = torch.ones(cout)
    plan = K.ConvPlan(w, b, s, k // 2, s - 1 if tr else 0, tr, act, gamma, beta)
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    y = plan(x)
    for _ in range(10): plan(x, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): plan(x, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = plan.flops(B, H, W)
    print(f"{name:34s} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  ({100*fl/ms/1e9/157.3:5.1f}% of peak)", flush=True)
    return ms, fl

B = int(os.environ.get("B", "64"))
tot = [0.0, 0.0]
for C in (192, 128):
    t = f = 0.0
    for l in [(f"g_a.1 3->{C} s2 gdn @256", 3, C, 5, 2, False, K.ACT_GDN, B, 256, 256),
              (f"g_a.2 {C}->{C} s2 gdn @128", C, C, 5, 2, False, K.ACT_GDN, B, 128, 128),
              (f"g_a.3 {C}->{C} s2 gdn @64", C, C, 5, 2, False, K.ACT_GDN, B, 64, 64),
              (f"g_a.4 {C}->192 s2 @32", C, 192, 5, 2, False, K.ACT_NONE, B, 32, 32),
              (f"g_s.1 192->{C} T igdn @16", 192, C, 5, 2, True, K.ACT_IGDN, B, 16, 16),
              (f"g_s.2 {C}->{C} T igdn @32", C, C, 5, 2, True, K.ACT_IGDN, B, 32, 32),
              (f"g_s.3 {C}->{C} T igdn @64", C, C, 5, 2, True, K.ACT_IGDN, B, 64, 64),
              (f"g_s.4 {C}->3 T @128", C, 3, 5, 2, True, K.ACT_NONE, B, 128, 128)]:
        ms, fl = bench(*l)
        t += ms; f += fl
    print(f"== width {C}: {t:.2f} ms for {f/1e9:.0f} GFLOP = {f/t/1e9:.1f} TFLOP/s ({100*f/t/1e9/157.3:.1f}% of peak)\n", flush=True)
