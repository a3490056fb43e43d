import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K
g = torch.Generator().manual_seed(0)
B, H, W = int(os.environ.get("B", 2)), 64, 64
x = torch.randn(B, 3, H, W, generator=g)
w = torch.randn(128, 3, 5, 5, generator=g) * 0.1
b = torch.randn(128, generator=g) * 0.1
ref = F.conv2d(x, w, b, stride=2, padding=2)
plan = K.ConvPlan(w, b, 2, 2, 0, False, K.ACT_GDN, 0.1 * torch.eye(128), torch.ones(128))
nrm = torch.einsum("ij,bjhw->bihw", 0.1 * torch.eye(128), ref ** 2) + 1.0
ref = ref / torch.sqrt(nrm)
got = plan(x.cuda()).cpu()
err = (got - ref).abs()
print("max err", float(err.max()), "shape", tuple(got.shape))
print("per image", [float(err[i].max()) for i in range(B)])
print("per channel block of 8:", [round(float(err[:, c:c + 8].max()), 3) for c in range(0, 128, 8)])
print("per row block of 4:", [round(float(err[:, :, r:r + 4].max()), 3) for r in range(0, 32, 4)])
print("per col block of 4:", [round(float(err[:, :, :, r:r + 4].max()), 3) for r in range(0, 32, 4)])
bad = (err > 1e-3).nonzero()
print("bad count", len(bad), "first", bad[:6].tolist())
if len(bad):
    i = bad[0].tolist(); print("got", float(got[tuple(i)]), "ref", float(ref[tuple(i)]))
    # is the value found elsewhere?
    v = got[tuple(i)]
    m = ((ref - v).abs() < 1e-5).nonzero()
    print("ref positions holding that value:", m[:4].tolist())
