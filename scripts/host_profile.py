#!/usr/bin/env python3
"""cProfile of the host side of one compress+decompress step (where the GPU idles between kernels)."""
import sys, os, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights

codec = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to("cuda")
codec.update_state()
x = torch.rand(256, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(2):
    codec.decompress(codec.compress(x))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    codec.decompress(codec.compress(x))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
