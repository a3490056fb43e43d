#!/usr/bin/env python3
"""GPU probe: step time of compress+decompress over 256 images with 1..4 concurrent stream workers, input either
resident in HBM or pinned on the host (uploaded inside the step)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per worker stream (the default 4 make streams share)
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--host-input", action="store_true")
    ap.add_argument("--stagger", action="store_true")
    ap.add_argument("--waves", type=int, default=0, help="rANS wavefronts per workgroup of the fused session")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--token", action="store_true")
    ap.add_argument("--free", action="store_true", help="free-running workers: each loops over its shard --steps times, one join at the end")
    ap.add_argument("--prio", action="store_true", help="descending stream priority per worker (breaks the lock-step)")
    args = ap.parse_args()
    from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool, split_batch
    from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
    dev = torch.device("cuda", 0)

    def make():
        c = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to(dev)
        c.update_state()
        c.entropy_coder.fused_rans_waves = args.waves
        c.entropy_coder.use_fused_session = not args.no_fused
        c.entropy_coder.fused_transform_token = args.token
        return c

    g = torch.Generator().manual_seed(1234)
    x = torch.rand(args.batch, 3, 256, 256, generator=g)
    x = x.pin_memory() if args.host_input else x.to(dev)

    def work(codec, shard):
        data = codec.compress(shard)
        xh = codec.decompress(data)
        return len(data)

    for w in args.workers:
        lo_hi = torch.cuda.Stream.priority_range()
        pr = [min(lo_hi) + i for i in range(w)] if args.prio else None
        with StreamWorkerPool(make, w, dev, priorities=pr) as pool:
            shards = split_batch(x, w)
            for _ in range(3):
                pool.map(work, shards, stagger=args.stagger)
            torch.cuda.synchronize()
            t0 = time.time()
            if args.free:
                def loop(codec, shard):
                    for _ in range(args.steps):
                        n_ = work(codec, shard)
                    return n_
                n = sum(pool.map(loop, shards, stagger=args.stagger))
            else:
                for _ in range(args.steps):
                    n = sum(pool.map(work, shards, stagger=args.stagger))
            torch.cuda.synchronize()
            dt = (time.time() - t0) / args.steps
        print(json.dumps(dict(workers=w, wpb=args.waves, fused=not args.no_fused, token=args.token, host_input=args.host_input, stagger=args.stagger, free=args.free, prio=str(pr) + str(lo_hi), ms_per_step=dt * 1e3,
                              mpix_s=args.batch * 65536 / dt / 1e6, bytes=n)), flush=True)


if __name__ == "__main__":
    main()
