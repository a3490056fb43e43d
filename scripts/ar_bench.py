#!/usr/bin/env python3
"""Timing of the AR codec graphs (cfg-3 topo-group patterns, cfg-4 BaSIC levels) -- informational, not the headline."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.presets import topogroup_ar_codec, basic_codec, seed_synthetic_weights

def prep(codec):
    codec = seed_synthetic_weights(codec, 0).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in codec.named_parameters():
            if ".latent_node_entropy_coders.y." in n: p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.01))
    codec = codec.cuda(); codec.update_state()
    return codec

def run(name, codec, B, size=256, steps=3):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, size, size, generator=g).cuda()
    d = codec.compress(x); codec.decompress(d); torch.cuda.synchronize()
    te = td = 0.0
    for _ in range(steps):
        t0 = time.time(); d = codec.compress(x); torch.cuda.synchronize(); t1 = time.time(); xh = codec.decompress(d); torch.cuda.synchronize(); t2 = time.time()
        te += t1 - t0; td += t2 - t1
    pix = B * size * size * steps
    print(f"{name:34s} B={B:3d} enc {te/steps*1e3:8.1f} ms dec {td/steps*1e3:8.1f} ms  {pix/(te+td)/1e6:7.1f} Mpix/s  bpp {len(d)*8/(B*size*size):.3f}")

which = sys.argv[1] if __name__ == "__main__" and len(sys.argv) > 1 else ("all" if __name__ == "__main__" else "none")
if which in ("all", "ar"):
    for m, G in [("none", 1), ("checkerboard", 1), ("raster2x2", 1), ("channelwise", 4), ("elic", 1)]:
        run(f"topogroup {m}", prep(topogroup_ar_codec(m, channel_groups=G)), 256)
if which in ("all", "basic"):
    c = prep(basic_codec())
    for lvl in (0, 7):
        c.set_complex_level(lvl)
        run(f"BaSIC scanline level {lvl}", c, 64)
if which in ("all", "combined"):
    c = prep(basic_codec(combined_entropy_coder=True))
    for lvl in range(8):
        c.set_complex_level(lvl)
        sel = int(c.entropy_coder._complexity_param_all_levels[lvl]()["pgmy"].argmax())
        run(f"BaSIC combined level {lvl} (y-coder {sel})", c, 64)
if which in ("all", "search"):
    # cost of ONE controller setting of the complexity search on a Kodak-shaped set (24 x 3x512x768, batch 1);
    # the full product search of the preset evaluates 5^4 = 625 of them
    c = prep(basic_codec())
    ec = c.entropy_coder
    data = [torch.rand(1, 3, 512, 768, generator=torch.Generator().manual_seed(i)) for i in range(24)]
    setting = {n: ec.node_generators[n](0) for n in ec.complexity_level_controller_nodes}
    ec._test_dataset_complexity_performance(data[:2], **setting)
    torch.cuda.synchronize(); t0 = time.time()
    flops, loss = ec._test_dataset_complexity_performance(data, **setting)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"complexity search: one setting over 24 Kodak-shaped images {dt*1e3:.1f} ms (FLOPs/dim {flops:.1f}, loss/dim {loss:.4f}); x625 = {dt*625:.0f} s")
