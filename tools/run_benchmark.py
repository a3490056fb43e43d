#!/usr/bin/env python3
"""Run the codec benchmark on one MI355X and write metrics.csv / metrics_2d.csv (the artefacts of the reference's
tools/run_benchmark.py for the testing half of a benchmark config).

  python tools/run_benchmark.py --codec hyperprior --images /data/kodak --out runs/hp_kodak
  python tools/run_benchmark.py --codec basic --synthetic 8 --size 256 --complexity-levels 0 1 2 3 4 5 6 7 --out runs/basic
  python tools/run_benchmark.py --codec topogroup --method checkerboard --synthetic 4 --out runs/ckbd

Weights: --checkpoint takes a state_dict saved from the reference (keys as in INTEGRATION.md); without it the
seeded synthetic weights of cbench_basic_amd.presets are used (plumbing / throughput runs).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--codec", choices=["hyperprior", "topogroup", "basic"], default="hyperprior")
    ap.add_argument("--method", default="checkerboard", help="topo-group pattern of --codec topogroup")
    ap.add_argument("--checkpoint", default=None, help="state_dict (.pt) of the entropy coder or of the whole codec")
    ap.add_argument("--images", default=None, help="folder of PNG/JPEG images")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic images (torch.manual_seed(i); rand)")
    ap.add_argument("--size", type=int, default=256, help="height = width of the synthetic images")
    ap.add_argument("--height", type=int, default=None, help="synthetic image height (default --size)")
    ap.add_argument("--width", type=int, default=None, help="synthetic image width (default --size)")
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--complexity-levels", type=int, nargs="*", default=[])
    ap.add_argument("--rate-levels", type=int, nargs="*", default=[])
    ap.add_argument("--forward-pass", action="store_true", help="also record the forward-pass rate estimate")
    ap.add_argument("--complexity-search", action="store_true",
                    help="--codec basic: find the complexity levels with the greedy search over the test images "
                         "(post_training_process) instead of the fixed ladder")
    ap.add_argument("--warmup", action="store_true",
                    help="code the first batch once before the timed run (plan building, table upload, graph capture are "
                         "one-time costs; the reference's harness has no such step)")
    ap.add_argument("--workers", type=int, default=0,
                    help="num_testing_workers: concurrent stream workers on the GPU, each with its own replica of the codec; dataset "
                         "items are coded concurrently (the reference's multiprocessing pool, basic_benchmark.py:829-858, GPU-native)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    if args.workers > 1:   # before HIP initialises: one hardware queue per worker stream
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        os.environ.setdefault("BASIC_RANS_WPB", "8")
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: the codec has no CPU path")

    from cbench_basic_amd import presets
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    from cbench_basic_amd.data import ImageFolderDataset, RandomImageDataset, batched
    if args.images:
        ds = ImageFolderDataset(args.images)
    else:
        ds = RandomImageDataset(num=args.synthetic or 8, size=(3, args.height or args.size, args.width or args.size))
    batches = list(batched(ds, args.batch_size))
    builders = dict(hyperprior=presets.hyperprior_codec, basic=presets.basic_codec,
                    topogroup=lambda: presets.topogroup_ar_codec(method=args.method))
    codec = presets.basic_codec(search_dataset=batches) if (args.codec == "basic" and args.complexity_search) else builders[args.codec]()
    if args.checkpoint:
        sd = torch.load(args.checkpoint, map_location="cpu")
        sd = sd.get("state_dict", sd)
        missing = codec.load_state_dict(sd, strict=False) if any(k.startswith("entropy_coder.") for k in sd) \
            else codec.entropy_coder.load_state_dict(sd, strict=False)
        print("checkpoint loaded; missing keys:", len(missing.missing_keys), "unexpected:", len(missing.unexpected_keys))
    else:
        presets.seed_synthetic_weights(codec, seed=0)
    codec = codec.eval().to("cuda")
    if args.warmup:
        codec.update_state()
        for lvl in (args.complexity_levels or [None]):
            if lvl is not None:
                codec.set_complex_level(lvl)
            codec.decompress(codec.compress(batches[0].to("cuda")))
    bench = BasicLosslessCompressionBenchmark(codec, batches,
                                              distortion_metric=PytorchBatchedDistortion(),
                                              nn_codec_use_forward_pass=args.forward_pass,
                                              testing_complexity_levels=args.complexity_levels,
                                              testing_variable_rate_levels=args.rate_levels, output_dir=args.out,
                                              num_testing_workers=args.workers,
                                              codec_builder=(lambda: presets.seed_synthetic_weights(builders[args.codec](), seed=0)) if args.workers > 1 else None)
    if args.workers > 1 and args.warmup:   # the replicas' one-time costs too, one replica at a time (HIP-graph capture)
        pool = bench._worker_pool(batches[0])
        for r, st in zip(pool.codecs, pool.streams):
            with torch.cuda.stream(st):
                for lvl in (args.complexity_levels or [None]):
                    if lvl is not None:
                        r.set_complex_level(lvl)
                    r.decompress(r.compress(batches[0].to("cuda")))
            torch.cuda.synchronize()
    metrics = bench.run_benchmark(ignore_exist_metrics=True)
    bench.close()
    print(json.dumps(metrics, indent=1))


if __name__ == "__main__":
    main()
