"""Module protocol the reference's benchmark harness expects from a codec:
nn.Module semantics plus ``profiler``, named caches (``loss_dict``/``metric_dict``/...),
``.device``, ``post_training_process``, ``update_state``.

Reference: cbench/modules/base.py:36-47,159-169 (profiler), cbench/nn/base.py:226-455
(NNCacheImpl), :489-491 (.device).  Only what the encode/decode path touches is kept.
"""
import time
from collections import defaultdict
from contextlib import contextmanager

import torch
import torch.nn as nn


class TimeProfiler:
    """Aggregates wall-clock spans per name (utils/logging_utils.py:82-160 behaviour)."""

    def __init__(self):
        self.total = defaultdict(float)
        self.count = defaultdict(int)

    @contextmanager
    def start_time_profile(self, name):
        t0 = time.time()
        try:
            yield
        finally:
            self.total[name] += time.time() - t0
            self.count[name] += 1

    def reset(self):
        self.total.clear()
        self.count.clear()

    def results(self):
        return {k: self.total[k] / max(1, self.count[k]) for k in self.total}


class HotPathModule(nn.Module):
    """nn.Module + profiler + caches.  Equivalent of NNTrainableModule for inference."""

    _CACHE_NAMES = ("loss_dict", "metric_dict", "moniter_dict", "hist_dict", "image_dict")

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.profiler = TimeProfiler()
        self._cache = {name: dict() for name in self._CACHE_NAMES}
        self.register_buffer("_device_indicator", torch.zeros(1), persistent=False)

    @property
    def device(self):
        return self._device_indicator.device

    # ---- caches (nn/base.py:226-455)
    def update_cache(self, cache_name="common", **kwargs):
        self._cache.setdefault(cache_name, dict()).update(**kwargs)

    def get_raw_cache(self, cache_name="common"):
        return self._cache.setdefault(cache_name, dict())

    def get_cache(self, cache_name="common", recursive=True, prefix=""):
        out = {prefix + k: v for k, v in self._cache.get(cache_name, {}).items()}
        if recursive:
            for name, child in self.named_children():
                if isinstance(child, HotPathModule):
                    out.update(child.get_cache(cache_name, True, prefix + name + "."))
                else:
                    for sub_name, sub in child.named_modules():
                        if isinstance(sub, HotPathModule) and sub is not child:
                            pass
        return out

    def reset_cache(self, cache_name="common"):
        self._cache[cache_name] = dict()

    def reset_all_cache(self):
        for m in self.modules():
            if isinstance(m, HotPathModule):
                for name in list(m._cache.keys()):
                    m._cache[name] = dict()

    def collect_profiler_results(self, recursive=True, clear=True):
        out = dict(self.profiler.results())
        if recursive:
            for name, m in self.named_modules():
                if m is not self and isinstance(m, HotPathModule):
                    out.update({f"{name}.{k}": v for k, v in m.profiler.results().items()})
                    if clear:
                        m.profiler.reset()
        if clear:
            self.profiler.reset()
        return out

    # ---- harness protocol
    def post_training_process(self, *args, **kwargs):
        pass

    def update_state(self, *args, **kwargs):
        pass
