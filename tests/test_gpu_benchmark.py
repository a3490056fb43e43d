"""The benchmark harness (cbench_basic_amd/benchmark, mirror of cbench/benchmark/basic_benchmark.py) on the GPU:
metric names and their arithmetic for a plain codec and for the BaSIC complexity ladder, metrics.csv / metrics_2d.csv."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def test_harness_hyperprior(tmp_path):
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    from cbench_basic_amd.data import RandomImageDataset, batched
    from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
    codec = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to("cuda")
    ds = RandomImageDataset(num=3, size=(3, 128, 128))
    bench = BasicLosslessCompressionBenchmark(codec, list(batched(ds, 1)), distortion_metric=PytorchBatchedDistortion(),
                                              nn_codec_use_forward_pass=True, output_dir=str(tmp_path))
    m = bench.run_benchmark()
    # no level prefixes: the reference's f"{prefix}_{key}" leaves a leading underscore
    for key in ("original_length", "compression_ratio", "compressed_length", "time_compress", "speed_compress",
                "time_decompress", "speed_decompress", "time_total", "speed_total", "psnr",
                "compression_ratio_nn_forward", "compressed_length_nn_forward"):
        assert "_" + key in m, key
    assert m["_original_length"] == 3 * 128 * 128 * 4
    assert abs(m["_time_total"] - m["_time_compress"] - m["_time_decompress"]) < 1e-6
    # the forward-pass estimate tracks the coded size
    assert abs(m["_compressed_length_nn_forward"] - m["_compressed_length"]) < 0.1 * m["_compressed_length"]
    # reproduce the numbers by hand
    lens, psnr = [], []
    for x in batched(ds, 1):
        data = codec.compress(x.cuda())
        lens.append(len(data))
        xh = codec.decompress(data).cpu()
        psnr.append(-10 * np.log10(torch.mean((xh - x) ** 2).item()))
    assert abs(m["_compressed_length"] - np.mean(lens)) < 1e-9
    assert abs(m["_compression_ratio"] - np.mean(lens) / (3 * 128 * 128 * 4)) < 1e-12
    assert abs(m["_psnr"] - np.mean(psnr)) < 1e-3
    rows = _rows(tmp_path / "metrics.csv")
    assert len(rows) == 1 and abs(float(rows[0]["_compressed_length"]) - np.mean(lens)) < 1e-6
    # a second run_benchmark returns the stored metrics (metrics file exists) unless told otherwise
    assert bench.run_benchmark() == m


def test_harness_basic_complexity_levels(tmp_path):
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    from cbench_basic_amd.data import RandomImageDataset, batched
    from cbench_basic_amd.presets import basic_codec, seed_synthetic_weights
    codec = seed_synthetic_weights(basic_codec(), seed=0).eval().to("cuda")
    ds = RandomImageDataset(num=2, size=(3, 64, 128))
    levels = [0, 3, 7]
    bench = BasicLosslessCompressionBenchmark(codec, list(batched(ds, 1)), distortion_metric=PytorchBatchedDistortion(),
                                              testing_complexity_levels=levels, output_dir=str(tmp_path))
    m = bench.run_benchmark()
    for lv in levels:
        assert f"sclevel{lv}_compressed_length" in m and f"sclevel{lv}_psnr" in m
    rows = _rows(tmp_path / "metrics_2d.csv")
    assert [r["name"] for r in rows] == [f"sclevel{lv}" for lv in levels]
    for lv, r in zip(levels, rows):
        codec.set_complex_level(lv)
        codec.update_state()
        lens = [len(codec.compress(x.cuda())) for x in batched(ds, 1)]
        assert abs(float(r["compressed_length"]) - np.mean(lens)) < 1e-6


def test_harness_grouped_codec_rate_x_complexity_sweep(tmp_path):
    """BASELINE configs[3] as written ("4 lambda-codecs x 8 complexity levels",
    configs/presets/lossy_latent_graph_scalable_ar_models.py:733-757): a GroupedVariableRateCodec of four BaSIC codecs
    swept through the harness; metrics_2d.csv carries one row per (complexity, rate) pair named sclevel{i}_vrlevel{j}."""
    from cbench_basic_amd.benchmark import BasicLosslessCompressionBenchmark, PytorchBatchedDistortion
    from cbench_basic_amd.codecs.grouped import GroupedVariableRateCodec
    from cbench_basic_amd.data import RandomImageDataset, batched
    from cbench_basic_amd.presets import basic_codec, seed_synthetic_weights
    members = [seed_synthetic_weights(basic_codec(widths=[16, 32], M=32, num_complex_levels=8), seed=10 + i, y_std=0.3 + 0.3 * i).eval()
               for i in range(4)]
    codec = GroupedVariableRateCodec(members).to("cuda")
    assert codec.num_rate_levels == 4 and codec.num_complex_levels == 8
    assert sorted({k.split(".")[0] for k in codec.state_dict()}) == ["codec_0", "codec_1", "codec_2", "codec_3"]
    ds = RandomImageDataset(num=2, size=(3, 64, 64))
    bench = BasicLosslessCompressionBenchmark(codec, list(batched(ds, 1)), distortion_metric=PytorchBatchedDistortion(),
                                              testing_variable_rate_levels=list(range(4)), testing_complexity_levels=list(range(8)),
                                              output_dir=str(tmp_path))
    m = bench.run_benchmark()
    rows = _rows(tmp_path / "metrics_2d.csv")
    assert [r["name"] for r in rows] == [f"sclevel{c}_vrlevel{r}" for c in range(8) for r in range(4)]
    for c in (0, 5, 7):
        for r in (0, 3):
            codec.set_complex_level(c)
            codec.set_rate_level(r)
            codec.update_state()
            assert codec.active_codec is members[r]
            lens = [len(codec.compress(x)) for x in batched(ds, 1)]
            assert abs(m[f"sclevel{c}_vrlevel{r}_compressed_length"] - np.mean(lens)) < 1e-6
            assert abs(float(rows[c * 4 + r]["compressed_length"]) - np.mean(lens)) < 1e-6
    # the members really differ (four rate points) and every level decodes
    assert len({round(m[f"sclevel0_vrlevel{r}_compressed_length"], 3) for r in range(4)}) == 4
    x = next(iter(batched(ds, 1)))
    assert codec.decompress(codec.compress(x)).shape == x.shape
