#!/usr/bin/env python3
"""In-kernel step profile (BASIC_SCAN_PROFILE=1) of the batched scan-line launches on CODEC data: BaSIC level 0, 64 synthetic images."""
import os, sys, torch
os.environ["BASIC_SCAN_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import image
from cbench_basic_amd.presets import basic_codec, seed_synthetic_weights
codec = seed_synthetic_weights(basic_codec(), seed=0).eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for n, p in codec.named_parameters():
        if ".latent_node_entropy_coders.y." in n:
            p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.01))
codec = codec.cuda()
codec.update_state()
codec.set_complex_level(0)
if os.environ.get("SIZE"):   # one Kodak-shaped image
    hh, ww = (int(v) for v in os.environ["SIZE"].split("x"))
    torch.manual_seed(0)
    x = torch.rand(int(os.environ.get("B", "1")), 3, hh, ww).cuda()
else:
    x = torch.stack([image(i, 256) for i in range(int(os.environ.get("B", "64")))]).cuda()
for _ in range(2):
    data = codec.compress(x)
    codec.decompress(data)
torch.cuda.synchronize()
