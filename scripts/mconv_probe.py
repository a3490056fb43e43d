"""Per-layer timing of the checkerboard codec's masked-convolution launches (256 images 16 x 16, 32,768 coded positions) under
environment variants: which kernel (BASIC_MCONV_KERNEL) and the dma kernel's timing ablations (BASIC_MCONV_DEBUG: 1 no staging,
2 no MFMA stages, 4 no stores).  Usage: python scripts/mconv_probe.py [variant ...]   variant = name:ENV=VAL,ENV=VAL"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K
torch.manual_seed(0)
B, H, W = int(os.environ.get("PROBE_BATCH", 256)), 16, 16
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
cb = ((yy + xx) % 2).int()
t2 = torch.stack([cb, torch.full_like(cb, -1)]).cuda()
t1 = cb[None].cuda()
p = torch.arange(H * W)[cb.reshape(-1) == 1]
pos = (torch.arange(B).reshape(-1, 1) * H * W + p.reshape(1, -1)).reshape(-1).int().cuda()
layers = [("768->1536 g2/2", 768, 1536, 1, 2, 2, True, 0.75), ("1536->1536 g2/2", 1536, 1536, 1, 2, 2, True, 0.75),
          ("1536->384 g2/1", 1536, 384, 1, 2, 1, True, 1.0), ("ctx 5x5 192->384", 192, 384, 5, 1, 1, False, 12 / 25)]
# step-contiguous plane order of the hidden activations (PROBE_PERM=1): the coded parity first
perm = torch.empty(H * W, dtype=torch.int32)
order = torch.argsort(1 - cb.reshape(-1), stable=True)
perm[order] = torch.arange(H * W, dtype=torch.int32)
perm = perm.cuda() if os.environ.get("PROBE_PERM") else None
perms = [(perm, perm), (perm, perm), (perm, None), (None, perm)]
plans = []
for name, cin, cout, k, gi, go, same, frac in layers:
    w = torch.randn(cout, cin, k, k) * 0.02
    plan = K.MaskedConvPlan(w, torch.zeros(cout), gi, go, same, K.ACT_LEAKY_RELU)
    x = torch.randn(B, cin, H, W, device="cuda")
    out = torch.zeros(B, cout, H, W, device="cuda")
    ti, to = (t2 if gi == 2 else t1), (t2 if go == 2 else t1)
    plans.append((name, plan, x, ti, to, out, 2 * frac * cin * cout * k * k * pos.numel() / 1e9))
if os.environ.get("PROBE_LAYERS"):   # e.g. PROBE_LAYERS=1 : only the 1536 -> 1536 layer (counter passes)
    keep = [int(v) for v in os.environ["PROBE_LAYERS"].split(",")]
    plans, perms = [plans[i] for i in keep], [perms[i] for i in keep]
variants = sys.argv[1:] or ["default:"]
for _ in range(40):   # clocks up before the first variant (the first timed variant of a run used to read ~10 % slow)
    for (name, plan, x, ti, to, out, gf), (ip, op) in zip(plans, perms):
        plan(x, ti, to, pos, out, in_perm=ip, out_perm=op)
torch.cuda.synchronize()
for v in variants:
    vname, _, envs = v.partition(":")
    kv = dict(e.split("=") for e in envs.split(",") if e)
    os.environ.update(kv)
    line = []
    for (name, plan, x, ti, to, out, gf), (ip, op) in zip(plans, perms):
        for _ in range(2):
            plan(x, ti, to, pos, out, in_perm=ip, out_perm=op)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            plan(x, ti, to, pos, out, in_perm=ip, out_perm=op)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        line.append(f"{name} {ms:.3f} ms {gf / ms:.0f} TF")
    for k_ in kv:
        del os.environ[k_]
    print(f"{vname:>22s} | " + " | ".join(line), flush=True)
