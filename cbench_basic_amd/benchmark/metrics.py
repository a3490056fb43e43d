"""Metric accumulation, distortion and Bjontegaard-delta metrics.

Reference: ``cbench/utils/logger.py`` (MetricLogger: running means), ``cbench/benchmark/metrics/pytorch_distortion.py:12-60``
(PSNR over the whole batch), ``cbench/benchmark/metrics/bj_delta.py:6-94``.
"""
import math
from collections import OrderedDict

import numpy as np
import torch


class MetricLogger:
    """Running sums; ``get_global_average()`` -> {name: mean over all updates}."""

    def __init__(self):
        self._sum, self._cnt = OrderedDict(), OrderedDict()

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self._sum[k] = self._sum.get(k, 0.0) + float(v)
            self._cnt[k] = self._cnt.get(k, 0) + 1

    def get_global_average(self):
        return OrderedDict((k, self._sum[k] / self._cnt[k]) for k in self._sum)

    def clear(self):
        self._sum.clear()
        self._cnt.clear()

    def merge(self, other):
        """Adds another logger's sums and counts (the per-worker loggers of a parallel run)."""
        for k in other._sum:
            self._sum[k] = self._sum.get(k, 0.0) + other._sum[k]
            self._cnt[k] = self._cnt.get(k, 0) + other._cnt[k]


class PytorchBatchedDistortion:
    """pytorch_distortion.py:21-68: ``"psnr"`` (of the mean squared error over the whole batch) and ``"ms-ssim"`` (mean over the
    batch; benchmark/ms_ssim.py restates the absent pytorch_msssim package: parity-unpinned)."""

    def __init__(self, *args, metrics="psnr", max_val=1.0, **kwargs):
        self._metrics = metrics if isinstance(metrics, list) else [metrics]
        for m in self._metrics:
            if m not in ("psnr", "ms-ssim"):
                raise NotImplementedError(f"{m} is not implemented!")
        self.max_val = max_val
        self.metric_logger = MetricLogger()

    @property
    def name(self):
        return "&".join(self._metrics)

    @property
    def metrics(self):
        return self._metrics

    def __call__(self, output, target, cache_metrics=True):
        output = output.type_as(target)[..., :target.shape[-2], :target.shape[-1]]  # make spatial size equal
        result = {}
        if "psnr" in self._metrics:
            if output.is_cuda:
                from ..nn import kernels as K
                mse = float(K.mse_per_image(output.contiguous(), target.contiguous()).double().mean())
            else:
                mse = torch.mean((output - target) ** 2).item()
            result["psnr"] = 20 * np.log10(self.max_val) - 10 * np.log10(mse)
        if "ms-ssim" in self._metrics:
            from .ms_ssim import ms_ssim
            result["ms-ssim"] = float(ms_ssim(output, target, data_range=self.max_val))
        result = {m: result[m] for m in self._metrics}   # the reference's order
        if cache_metrics:
            self.metric_logger.update(**result)
        return result

    def collect_metrics(self):
        return self.metric_logger.get_global_average()

    def reset(self):
        self.metric_logger.clear()


def bj_delta(R1, PSNR1, R2, PSNR2, mode=0):
    """Bjontegaard delta (bj_delta.py:48-94): mode 0 = average PSNR difference, mode 1 = average rate difference in %,
    from cubic fits over log-rate integrated on the common interval."""
    lR1, lR2 = np.log(R1), np.log(R2)
    if mode == 0:
        p1, p2 = np.polyfit(lR1, PSNR1, 3), np.polyfit(lR2, PSNR2, 3)
        lo, hi = max(min(lR1), min(lR2)), min(max(lR1), max(lR2))
        i1, i2 = np.polyint(p1), np.polyint(p2)
        int1 = np.polyval(i1, hi) - np.polyval(i1, lo)
        int2 = np.polyval(i2, hi) - np.polyval(i2, lo)
        return (int2 - int1) / (hi - lo)
    p1, p2 = np.polyfit(PSNR1, lR1, 3), np.polyfit(PSNR2, lR2, 3)
    lo, hi = max(min(PSNR1), min(PSNR2)), min(max(PSNR1), max(PSNR2))
    i1, i2 = np.polyint(p1), np.polyint(p2)
    int1 = np.polyval(i1, hi) - np.polyval(i1, lo)
    int2 = np.polyval(i2, hi) - np.polyval(i2, lo)
    return (math.exp((int2 - int1) / (hi - lo)) - 1) * 100


class BJDeltaMetric:
    """bj_delta.py:6-45: called with (rate_pts, distortion_pts) of the tested codec against ``reference_pts``."""

    def __init__(self, reference_pts=None, collect_metric_names=("compressed_length", "psnr"), mode=0, **kwargs):
        assert mode in (0, 1)
        self.reference_pts, self.collect_metric_names, self.mode = reference_pts, collect_metric_names, mode
        self.metric_logger = MetricLogger()

    @property
    def name(self):
        return "BD-" + (self.collect_metric_names[1] if self.mode == 0 else "rate")

    def __call__(self, output, target=None):
        target = self.reference_pts if target is None else target
        (R1, P1), (R2, P2) = output, target
        try:
            result = {self.name: bj_delta(R1, P1, R2, P2, mode=self.mode)}
        except Exception:  # the reference swallows fit failures the same way
            result = {self.name: -100}
        self.metric_logger.update(**result)
        return result
