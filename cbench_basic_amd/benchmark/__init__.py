"""Benchmark harness around the codec: the caller side of the hot path.

Mirrors the parts of ``cbench/benchmark`` that drive ``codec.compress`` / ``decompress`` over a dataset and write the
reference's artefacts (``metrics.csv``, ``metrics_2d.csv``, BD-rate), so that ``tools/collect_results.py`` style
post-processing keeps working.  Training, caching, multiprocessing pools and logging backends are not part of it.
"""
from .basic_benchmark import BasicLosslessCompressionBenchmark  # noqa: F401
from .metrics import BJDeltaMetric, MetricLogger, PytorchBatchedDistortion, bj_delta  # noqa: F401
