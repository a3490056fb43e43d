"""One-off: random first-layer geometries (5x5, 1-3 input channels, 128 outputs, GDN) through the row-interleaved persistent kernel vs torch."""
import sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K
rng = np.random.default_rng(int(os.environ.get("SEED", 5)))
worst = 0.0
for it in range(int(os.environ.get("N", 120))):
    cin, s = int(rng.integers(1, 4)), int(rng.choice([1, 2]))
    B, H, W = int(rng.integers(1, 7)), int(rng.integers(5, 90)), int(rng.integers(5, 90))
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, cin, H, W, generator=g)
    if it % 7 == 0:
        x[0, 0, H // 2, W // 3] = float("inf")   # a non-finite pixel must only reach the outputs whose window holds it
    w = torch.randn(128, cin, 5, 5, generator=g) * 0.1
    b = torch.randn(128, generator=g) * 0.1
    gamma = torch.rand(128, 128, generator=g) * 0.02 + 0.1 * torch.eye(128)
    beta = torch.rand(128, generator=g) + 0.5
    ref = F.conv2d(x, w, b, stride=s, padding=2)
    ref = ref / torch.sqrt(torch.einsum("ij,bjhw->bihw", gamma, ref ** 2) + beta.view(1, -1, 1, 1))
    got = K.ConvPlan(w, b, s, 2, 0, False, K.ACT_GDN, gamma, beta)(x.cuda()).cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(fin, torch.isfinite(got)), (it, cin, s, B, H, W, int((fin != torch.isfinite(got)).sum()))
    err = float((got[fin] - ref[fin]).abs().max()) if fin.any() else 0.0
    worst = max(worst, err)
    assert err < 2e-4 * max(1.0, float(ref[fin].abs().max())), (it, cin, s, B, H, W, err)
print("ok", worst)
