"""The testing half of ``BasicLosslessCompressionBenchmark`` (cbench/benchmark/basic_benchmark.py:105-326,640-1060) and
``BaseBenchmark.save_metrics`` (cbench/benchmark/base.py:54-112): run codec.compress / decompress over a dataset for
every (complexity level, rate level), collect the reference's metric names, write ``metrics.csv`` and ``metrics_2d.csv``.

Step metrics (basic_benchmark.py:121-260): ``original_length`` (bytes of the input tensor), ``compression_ratio``,
``compressed_length``, ``time_compress`` / ``time_decompress`` / ``time_total`` (ms), ``speed_*`` (MiB/s of original
data), the distortion metric (``psnr``), and with ``nn_codec_use_forward_pass`` the ``*_nn_forward`` pair from
``codec.forward_estimate_bitlen``.  Level loop and key prefixes (``sclevel{c}_vrlevel{r}_<metric>``) follow
:913-1026, including BD-rate over the rate levels when there are more than three of them.

``num_testing_workers`` (:829-858): the reference hands contiguous index ranges of the dataloader to a multiprocessing pool --
and switches the pool off when testing on a device (``force_testing_device is None`` is required, :836).  Here the workers are
concurrent stream workers on the ONE GPU (stream_workers.py: host threads, each with its own HIP stream and its own replica of
the codec built by ``codec_builder``), so dataset items are coded concurrently: one item's serial rANS chains and launch-bound
AR steps run beside another item's transforms.  Per-item times are the latencies of the calls as they ran (concurrently);
``time_wall_dataset`` / ``speed_wall_dataset`` give the wall time and MiB/s of the whole dataset pass.
"""
import copy
import csv
import os
import pickle
import time
from collections import OrderedDict

import numpy as np
import torch

from .metrics import MetricLogger


class BasicLosslessCompressionBenchmark:
    def __init__(self, codec, dataloader, *args, distortion_metric=None, skip_decompress=False,
                 nn_codec_use_forward_pass=False, nn_codec_forward_pass_skip_compression=False,
                 testing_variable_rate_levels=None, testing_variable_rate_bj_delta_metric=None,
                 testing_complexity_levels=None, force_testing_device="cuda", output_dir=None, num_repeats=1,
                 num_testing_workers=0, codec_builder=None, worker_transform_token=None, **kwargs):
        self.codec = codec
        self.dataloader = dataloader
        self.distortion_metric = distortion_metric
        self.skip_decompress = skip_decompress
        self.nn_codec_use_forward_pass = nn_codec_use_forward_pass
        self.nn_codec_forward_pass_skip_compression = nn_codec_forward_pass_skip_compression
        self.testing_variable_rate_levels = list(testing_variable_rate_levels or [])
        self.testing_variable_rate_bj_delta_metric = testing_variable_rate_bj_delta_metric
        self.testing_complexity_levels = list(testing_complexity_levels or [])
        self.force_testing_device = force_testing_device
        self.output_dir = output_dir
        self.num_repeats = num_repeats
        self.metric_logger = MetricLogger()
        self.num_testing_workers = int(num_testing_workers or 0)
        self.codec_builder = codec_builder
        # concurrent sessions with ordered MFMA phases + packed rANS workgroups: pays for batches (measured: 48
        # images in batches of 8, 3 workers: 248 -> 297 MiB/s), costs at batch 1 where everything is latency (Kodak-shaped,
        # 4 workers: 21.7 -> 13.8 Mpix/s).  None = on for batches of 8 or more
        self.worker_transform_token = worker_transform_token
        if self.num_testing_workers > 1 and codec_builder is None:
            raise ValueError("num_testing_workers > 1 needs codec_builder: a callable returning a fresh codec of the same "
                             "architecture (every worker codes with its own replica; weights and levels are copied from `codec`)")
        self._pool = None

    # ---- files (base.py:41-52)
    @property
    def metric_file(self):
        return os.path.join(self.output_dir, "metrics.csv")

    @property
    def metric_raw_file(self):
        return os.path.join(self.output_dir, "metrics.pkl")

    # ---- one step (basic_benchmark.py:105-260)
    @staticmethod
    def _estimate_byte_length(data):
        if isinstance(data, bytes):
            return len(data)
        if isinstance(data, str):
            return len(data.encode("utf-8"))
        if isinstance(data, torch.Tensor):
            return int(np.prod(data.shape)) * data.element_size()
        if isinstance(data, np.ndarray):
            return int(data.size * data.itemsize)
        raise ValueError("Bitstream of data {} in type {} cannot be estimated!".format(data, type(data)))

    def _sync(self):
        # the queue of THIS caller is empty before and after every timed region: the current stream -- the default stream in a
        # sequential run (where it is the only one in use), the worker's own stream in a parallel one (a device-wide
        # synchronisation there would make every worker wait for all the others at every item)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()

    def _run_step(self, step, data, metric_logger, codec=None, distortion_metric=None):
        codec = self.codec if codec is None else codec
        # the batch stays where the loader left it (the host): codec.compress() uploads it INSIDE the timed region, as the
        # reference's does (general_codec.py:46-47 under basic_benchmark.py:199-202); only the metric's target is moved
        data_input = data
        data_target = data.to(self.force_testing_device) if self.force_testing_device else data
        original_length = self._estimate_byte_length(data_input)
        metric_logger.update(original_length=original_length)
        dm = self.distortion_metric if distortion_metric is None else distortion_metric
        if self.nn_codec_use_forward_pass and hasattr(codec, "forward_estimate_bitlen"):
            decompressed, compressed_length = codec.forward_estimate_bitlen(data_input)
            compressed_length = float(compressed_length)
            metric_logger.update(compression_ratio_nn_forward=compressed_length / original_length,
                                 compressed_length_nn_forward=compressed_length)
            if dm is not None:
                dm(decompressed, data_target)
            if self.nn_codec_forward_pass_skip_compression:
                return
        self._sync()
        t0 = time.time()
        compressed = codec.compress(data_input)
        self._sync()
        time_compress = time.time() - t0
        compressed_length = self._estimate_byte_length(compressed)
        metric_logger.update(compression_ratio=compressed_length / original_length, compressed_length=compressed_length,
                             time_compress=time_compress * 1000,
                             speed_compress=original_length / time_compress / 1024 / 1024)
        if not self.skip_decompress:
            t0 = time.time()
            decompressed = codec.decompress(compressed)
            self._sync()
            time_decompress = time.time() - t0
            metric_logger.update(time_decompress=time_decompress * 1000,
                                 speed_decompress=original_length / time_decompress / 1024 / 1024,
                                 time_total=(time_compress + time_decompress) * 1000,
                                 speed_total=original_length / (time_compress + time_decompress) / 1024 / 1024)
            if dm is not None:
                dm(decompressed, data_target)

    def _worker_pool(self, first_item=None):
        """The stream workers and their codec replicas, brought to the main codec's weights, levels and tables."""
        from .stream_workers import StreamWorkerPool
        if self._pool is None:
            self._token_on = self.worker_transform_token if self.worker_transform_token is not None else \
                bool(first_item is not None and hasattr(first_item, "shape") and len(first_item.shape) == 4 and first_item.shape[0] >= 8)
            def make():
                c = self.codec_builder()
                if hasattr(c, "eval"):
                    c.eval()
                if self._token_on:
                    for m in (c.modules() if hasattr(c, "modules") else []):   # concurrent sessions: ordered MFMA phases, packed rANS workgroups
                        if hasattr(m, "fused_transform_token"):
                            m.fused_transform_token, m.fused_rans_waves = True, 8
                return c.to(self.force_testing_device or "cuda") if hasattr(c, "to") else c
            self._pool = StreamWorkerPool(make, self.num_testing_workers, torch.device(self.force_testing_device or "cuda"))
        level_attrs = ("_current_complex_level", "_current_rate_level", "_current_task_idx", "active_codec_idx")
        # replicas already at the main codec's weights and levels (a warm-up pass, the next pass at the same level) are left
        # alone: reloading and update_state() would rebuild tables and layer plans, and the first item of every worker would
        # pay for it inside the timed pass.  Fingerprint: every state tensor's storage and in-place version counter + the levels.
        def fingerprint():
            mods = list(self.codec.named_modules()) if hasattr(self.codec, "named_modules") else [("", self.codec)]
            levels = tuple((n, a, repr(getattr(m, a))) for n, m in mods for a in level_attrs if hasattr(m, a))
            searched = tuple((n, id(getattr(m, "_complexity_param_all_levels", None)), getattr(m, "_num_complex_levels", None)) for n, m in mods)
            sd = self.codec.state_dict() if hasattr(self.codec, "state_dict") else {}
            state = tuple((k, t.data_ptr(), t._version) for k, t in sd.items())
            # ... and a content checksum (sum and sum of squares of every state tensor, one synchronisation per pass): edits that do
            # not bump the version counter (`.data`, a checkpoint loader writing through views) are seen too
            fl = [t.detach().double().reshape(-1) for t in sd.values() if torch.is_tensor(t) and t.numel() > 0]
            content = tuple(torch.stack([torch.stack([f.sum(), (f * f).sum()]) for f in fl]).sum(0).tolist()) if fl else ()
            return levels, searched, state, content
        fp = fingerprint()
        if getattr(self._pool, "_synced_to", None) == fp:
            return self._pool
        for r in self._pool.codecs:
            if hasattr(r, "named_modules"):
                theirs = dict(r.named_modules())
                for name, src in self.codec.named_modules():   # the codec itself, grouped members, their entropy coders
                    dst = theirs.get(name)
                    if dst is None:
                        continue
                    if hasattr(src, "_complexity_param_all_levels"):   # searched levels: modules made after construction
                        dst._complexity_param_all_levels = copy.deepcopy(src._complexity_param_all_levels)
                        dst._num_complex_levels = src._num_complex_levels
                    for a in level_attrs:
                        if hasattr(src, a):
                            setattr(dst, a, getattr(src, a))
                    if hasattr(dst, "_valid_host"):
                        dst._valid_host = None
                r.load_state_dict(self.codec.state_dict(), strict=False)
            else:
                for a in level_attrs:
                    if hasattr(self.codec, a):
                        setattr(r, a, getattr(self.codec, a))
            r.update_state()
        self._pool._synced_to = fp
        return self._pool

    def invalidate_replicas(self):
        """Force the stream workers' codec replicas to be re-synchronised before the next pass.  The automatic check covers the
        state_dict (storage, version counters, content checksum), the level attributes and the searched levels; call this after
        changing anything ELSE on the main codec that the replicas must share (a tuning attribute set by hand, a module swapped in)."""
        if self._pool is not None:
            self._pool._synced_to = None

    def _run_dataset(self):
        logger = MetricLogger()
        n_items, n_bytes = 0, 0
        self._sync()
        t0 = time.time()
        if self.num_testing_workers > 1 and (self.force_testing_device or "cuda").startswith("cuda"):
            items = list(self.dataloader)
            pool = self._worker_pool(items[0] if items else None)
            W = min(self.num_testing_workers, max(1, len(items)))
            base, extra = divmod(len(items), W)
            # contiguous index ranges as in the reference (:851-852), remainder spread (the reference drops len % W items)
            bounds = [0]
            for w in range(W):
                bounds.append(bounds[-1] + base + (1 if w < extra else 0))
            loggers = [MetricLogger() for _ in range(W)]
            dms = [copy.deepcopy(self.distortion_metric) if self.distortion_metric is not None else None for _ in range(W)]
            codec_of = {id(c): i for i, c in enumerate(pool.codecs)}
            self._sync()
            t0 = time.time()

            def work(codec, w):
                for step in range(bounds[w], bounds[w + 1]):
                    self._run_step(step, items[step], loggers[w], codec=codec, distortion_metric=dms[w])
                return None
            pool.map(work, list(range(W)))
            for w in range(W):
                logger.merge(loggers[w])
                if dms[w] is not None:
                    self.distortion_metric.metric_logger.merge(dms[w].metric_logger)
            for d in items:
                n_items, n_bytes = n_items + 1, n_bytes + self._estimate_byte_length(d)
        else:
            for step, data in enumerate(self.dataloader):
                self._run_step(step, data, logger)
                n_items, n_bytes = n_items + 1, n_bytes + self._estimate_byte_length(data)
        self._sync()
        wall = time.time() - t0
        out = logger.get_global_average()
        if n_items:
            out["time_wall_dataset"] = wall * 1000
            out["speed_wall_dataset"] = n_bytes / wall / 1024 / 1024
        return out

    def close(self):
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    # ---- level loop (basic_benchmark.py:640-1030): rate levels innermost, then complexity levels
    def run_testing(self, *args, **kwargs):
        metrics, metrics_2d = OrderedDict(), OrderedDict()
        vr, vc = self.testing_variable_rate_levels, self.testing_complexity_levels
        if hasattr(self.codec, "eval"):
            self.codec.eval()
        if self.force_testing_device and hasattr(self.codec, "to"):
            self.codec.to(self.force_testing_device)
        for _ in range(self.num_repeats):
            for ci, clevel in enumerate(vc if vc else [None]):
                rate_pts, distortion_pts = [], []
                for ri, rlevel in enumerate(vr if vr else [None]):
                    if clevel is not None:
                        self.codec.set_complex_level(clevel)
                    if rlevel is not None:
                        self.codec.set_rate_level(rlevel)
                    if hasattr(self.codec, "post_training_process"):
                        self.codec.post_training_process()
                    self.codec.update_state()
                    if self.distortion_metric is not None:
                        self.distortion_metric.reset()
                    current = self._run_dataset()
                    if self.distortion_metric is not None:
                        current.update(**self.distortion_metric.collect_metrics())
                    prefixes = []
                    if clevel is not None:
                        prefixes.append(f"sclevel{clevel}")
                    if rlevel is not None:
                        prefixes.append(f"vrlevel{rlevel}")
                        bj = self.testing_variable_rate_bj_delta_metric
                        if bj is not None:
                            rn, dn = bj.collect_metric_names
                            if rn in current and dn in current:
                                rate_pts.append(current[rn])
                                distortion_pts.append(current[dn])
                    prefix = "_".join(prefixes)
                    if prefixes:
                        metrics_2d.setdefault(prefix, OrderedDict())
                    for key, value in current.items():
                        metrics[f"{prefix}_{key}"] = value  # (sic: the reference keeps the underscore without a prefix)
                        if prefixes:
                            metrics_2d[prefix][key] = value
                    if len(vc) > 1 and hasattr(self.codec, "get_current_complex_metrics"):
                        for key, value in self.codec.get_current_complex_metrics().items():
                            metrics[f"{prefix}_{key}"] = value
                            if prefixes:
                                metrics_2d[prefix][key] = value
                # BD metric over this complexity level's rate points (needs at least 4 points, :979-992)
                bj = self.testing_variable_rate_bj_delta_metric
                if bj is not None and len(vr) > 3 and len(rate_pts) == len(vr):
                    bj_prefix = f"sclevel{ci}_" if vc else ""
                    value = bj((rate_pts, distortion_pts))[bj.name]
                    metrics[bj_prefix + bj.name] = value
                    metrics_2d[prefix][bj.name] = value
        if metrics_2d and self.output_dir:
            self.save_metrics(metric_file=os.path.join(self.output_dir, "metrics_2d.csv"),
                              metric_data=list(metrics_2d.values()), names=list(metrics_2d.keys()), raw=False)
        return metrics

    def run_benchmark(self, *args, run_testing=True, ignore_exist_metrics=False, **kwargs):
        if self.codec is None:
            raise ValueError("No codec to benchmark!")
        if self.output_dir:
            os.makedirs(self.output_dir, exist_ok=True)
            if not ignore_exist_metrics and os.path.exists(self.metric_raw_file):
                with open(self.metric_raw_file, "rb") as f:  # the benchmark has been run: its metrics are the result
                    return pickle.load(f)
        if run_testing:
            metric_dict = self.run_testing(*args, **kwargs)
            if self.output_dir:
                self.save_metrics(metric_dict)
            return metric_dict

    # ---- base.py:54-112 (CSV: name column first, then metric columns in first-seen order, then hparams)
    def save_metrics(self, metric_data=None, hparams=None, names=None, metric_file=None, raw=True):
        metric_file = metric_file or self.metric_file
        if metric_data is None:
            return
        if raw:
            with open(self.metric_raw_file, "wb") as f:
                pickle.dump(metric_data, f)
        fieldnames, f_metrics, f_hparams, rows = OrderedDict(), OrderedDict(), OrderedDict(), []

        def _append(metric, hparam=None, name=None):
            row = OrderedDict()
            if isinstance(metric, dict):
                if name is not None:
                    fieldnames["name"] = None
                    row["name"] = name
                for k in metric:
                    f_metrics[k] = None
                row.update(**metric)
                if isinstance(hparam, dict):
                    for k in hparam:
                        f_hparams[k] = None
                    row.update(**hparam)
            rows.append(row)

        if isinstance(metric_data, (list, tuple)):
            hparams = hparams or [None] * len(metric_data)
            names = names or [None] * len(metric_data)
            for m, h, n in zip(metric_data, hparams, names):
                _append(m, h, n)
        else:
            _append(metric_data, hparams, names)
        with open(metric_file, "w", newline="") as f:
            writer = csv.DictWriter(f, fieldnames=list(fieldnames) + list(f_metrics) + list(f_hparams))
            writer.writeheader()
            writer.writerows(rows)
