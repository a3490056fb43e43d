"""num_testing_workers of the harness (basic_benchmark.py:829-858 made GPU-native: concurrent stream workers with codec
replicas): same per-item results as the sequential run, whatever the worker count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(n, size):
    out = []
    for i in range(n):
        torch.manual_seed(i)
        out.append(torch.rand(1, 3, size, size))
    return out


@pytest.mark.parametrize("kind", ["hyperprior", "basic"])
def test_parallel_dataset_pass_equals_sequential(kind, tmp_path):
    from cbench_basic_amd import presets
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    build = (lambda: presets.seed_synthetic_weights(presets.hyperprior_codec(), seed=0)) if kind == "hyperprior" else \
            (lambda: presets.seed_synthetic_weights(presets.basic_codec(), seed=0))
    size = 128 if kind == "hyperprior" else 64
    items = _batches(7, size)
    levels = [] if kind == "hyperprior" else [0, 5]
    res = {}
    codec = build().eval().cuda()    # ONE codec for both passes: the presets' y-coder parameters are not part of the seeded recipe
    for workers in (0, 3):
        bench = BasicLosslessCompressionBenchmark(codec, items, distortion_metric=PytorchBatchedDistortion(), testing_complexity_levels=levels,
                                                  output_dir=str(tmp_path / f"w{workers}"), num_testing_workers=workers,
                                                  codec_builder=build if workers else None)
        res[workers] = bench.run_benchmark(ignore_exist_metrics=True)
        bench.close()
    seq, par = res[0], res[3]
    keys = [k for k in seq if k.endswith(("compressed_length", "compression_ratio", "original_length", "psnr"))]
    assert keys and all(k in par for k in keys)
    for k in keys:     # sizes and distortion are properties of the items, not of who coded them
        assert abs(seq[k] - par[k]) <= 1e-9 * max(1.0, abs(seq[k])), (k, seq[k], par[k])
    assert any(k.endswith("speed_wall_dataset") for k in par)
    with pytest.raises(ValueError):
        BasicLosslessCompressionBenchmark(build(), items, num_testing_workers=2)


def test_parallel_pass_of_a_grouped_codec(tmp_path):
    """GroupedVariableRateCodec (codecs/base.py:138-243) through the harness with workers: the replicas follow the active member,
    the complexity level and the weights of every member."""
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    from cbench_basic_amd.codecs.grouped import GroupedVariableRateCodec
    from cbench_basic_amd.presets import basic_codec, seed_synthetic_weights

    def build():
        return GroupedVariableRateCodec([seed_synthetic_weights(basic_codec(widths=[16, 32], M=32, num_complex_levels=8), seed=10 + i,
                                                                y_std=0.3 + 0.3 * i).eval() for i in range(2)])
    codec = build().to("cuda")
    items = _batches(5, 64)
    res = {}
    for workers in (0, 2):
        bench = BasicLosslessCompressionBenchmark(codec, items, distortion_metric=PytorchBatchedDistortion(), testing_variable_rate_levels=[0, 1],
                                                  testing_complexity_levels=[0, 6], output_dir=str(tmp_path / f"g{workers}"),
                                                  num_testing_workers=workers, codec_builder=build if workers else None)
        res[workers] = bench.run_benchmark(ignore_exist_metrics=True)
        bench.close()
    keys = [k for k in res[0] if k.endswith(("compressed_length", "psnr"))]
    assert len(keys) == 8
    for k in keys:
        assert abs(res[0][k] - res[2][k]) <= 1e-9 * max(1.0, abs(res[0][k])), (k, res[0][k], res[2][k])
    assert len({round(res[2][f"sclevel0_vrlevel{r}_compressed_length"], 3) for r in (0, 1)}) == 2
