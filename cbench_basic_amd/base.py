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


class TorchCheckpointLoader:
    """cbench/nn/base.py:175-196: load a ``torch.save``d state_dict into a module -- optional ``key`` (sub-dict of the
    file), ``prefix`` (stripped from the keys that carry it), ``filter_keys`` (dropped; a missing one is a KeyError as in the
    reference), ``strict``.  Returns what ``load_state_dict`` returns."""

    def __init__(self, checkpoint_file, *args, strict=True, key=None, prefix=None, filter_keys=None, map_location="cpu", **kwargs):
        self.checkpoint_file = checkpoint_file
        self.strict = strict
        self.key = key
        self.prefix = prefix
        self.filter_keys = filter_keys
        self.map_location = map_location

    def load(self, model: nn.Module):
        state_dict = torch.load(self.checkpoint_file, map_location=self.map_location)
        if self.key is not None:
            state_dict = state_dict[self.key]
        if self.prefix is not None:
            state_dict = {(k[len(self.prefix):] if k.startswith(self.prefix) else k): v for k, v in state_dict.items()}
        if self.filter_keys is not None:
            for name in self.filter_keys:
                state_dict.pop(name)
        return model.load_state_dict(state_dict, strict=self.strict)


class HotPathModule(nn.Module):
    """nn.Module + profiler + caches.  Equivalent of NNTrainableModule for inference."""

    checkpoint_loader = None   # NNTrainableModule's constructor argument (nn/base.py:461-470); set it or pass one to load_checkpoint

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

    def load_checkpoint(self, checkpoint_loader=None):
        """NNTrainableModule.load_checkpoint (nn/base.py:507-519): a path is loaded non-strictly, a TorchCheckpointLoader
        with its own settings, None falls back to ``self.checkpoint_loader`` (no-op when that is None too).  Tables are
        the caller's next ``update_state()`` as in the reference."""
        if checkpoint_loader is None:
            checkpoint_loader = self.checkpoint_loader
        if checkpoint_loader is None:
            return None
        if isinstance(checkpoint_loader, str):
            return self.load_state_dict(torch.load(checkpoint_loader, map_location="cpu"), strict=False)
        if isinstance(checkpoint_loader, TorchCheckpointLoader):
            return checkpoint_loader.load(self)
        raise ValueError("Unsupported checkpoint_loader!")

    # ---- harness protocol
    def post_training_process(self, *args, **kwargs):
        pass

    def update_state(self, *args, **kwargs):
        pass
