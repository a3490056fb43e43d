#!/bin/bash
# last synthesis layer: one strip per lane vs two (identical results), HIP events (through gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04last
for v in 0 1; do
  for i in 1 2; do BASIC_CONV_LAST_2ROW=$v python scripts/conv_layer_bench.py g_s.4 2>/dev/null | sed "s/^/2row=$v  /"; done
done > gpurun_out/r04last/bench.txt
cat gpurun_out/r04last/bench.txt
python - <<'PY'
import os, torch, sys
sys.path.insert(0, ".")
from cbench_basic_amd.nn import kernels as K
g = torch.Generator().manual_seed(0)
w = torch.randn(128, 3, 5, 5, generator=g) * 0.02
b = torch.randn(3, generator=g)
plan = K.ConvPlan(w, b, 2, 2, 1, True, K.ACT_NONE, None, None)
for (B, H, W) in ((3, 128, 128), (2, 72, 100), (1, 64, 192), (2, 96, 68)):
    x = torch.randn(B, 128, H, W, generator=g).cuda()
    os.environ["BASIC_CONV_LAST_2ROW"] = "0"; y0 = plan(x).clone()
    os.environ["BASIC_CONV_LAST_2ROW"] = "1"; y1 = plan(x).clone()
    ref = torch.nn.functional.conv_transpose2d(x.cpu(), w, b, stride=2, padding=2, output_padding=1)
    print((B, H, W), "identical:", bool(torch.equal(y0, y1)), "max err vs torch-CPU:", float((y1.cpu() - ref).abs().max()))
PY
