"""Why is a 32-image leg slow after 256-image legs in the same process (bench.py: 523 Mpix/s as the first leg, 245-288 after)?
Runs a sequence of legs (batch x workers x token lanes) on ONE pool of six stream workers and prints each leg's rate.
usage: python scripts/leg_order_probe.py 32x6x4 32x6x4 256x3x1 32x6x4 ..."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool
from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def make_codec():
    c = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to(dev)
    c.update_state()
    c.entropy_coder.fused_rans_waves = 8
    return c


pool = StreamWorkerPool(make_codec, 6, dev)
g = torch.Generator().manual_seed(0)
x256 = torch.rand(256, 3, 256, 256, generator=g).to(dev)
for leg in sys.argv[1:]:
    b, w, lanes = (int(v) for v in leg.split("x"))
    x = x256[:b].contiguous()
    steps = max(12, 96 * 32 // b) if b < 256 else 12
    for c in pool.codecs:
        c.entropy_coder.fused_transform_token = lanes
    counts = [len(range(i, steps, w)) for i in range(w)]

    def loop(c, n):
        for _ in range(n):
            c.decompress(c.compress(x))
    pool.map(lambda c, n: loop(c, 1), [1] * w)
    torch.cuda.synchronize()
    t0 = time.time()
    pool.map(loop, counts)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"leg {leg}: {b * steps * 65536 / dt / 1e6:7.1f} Mpix/s  ({dt / steps * 1e3:.2f} ms per step, torch reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)", flush=True)
    if os.environ.get("PROBE_EMPTY_CACHE"):
        torch.cuda.empty_cache()
pool.close()
