"""Concurrent testing workers on ONE GPU: the analogue of the reference's ``num_testing_workers`` pool
(cbench/benchmark/basic_benchmark.py:829-858: disjoint index ranges mapped to pickled workers in a multiprocessing.Pool,
which the reference itself can only use on the CPU because its pybind11 coders do not pickle).

Here a worker is a host THREAD with its own HIP stream and its own codec replica (own layer plans, table sets, pinned
staging buffers; weights are a few MB).  Why it pays on an MI355X: a compress() / decompress() call alternates between
phases that fill the chip (the MFMA transforms) and phases that cannot (the rANS chains: one wavefront per image stream,
packed 16 to a compute unit, strictly serial, milliseconds long whatever the batch size).  With two or more workers the
chain phase of one sub-batch runs on a handful of compute units while the transforms of another sub-batch keep the
rest busy, and the pinned-memory upload of the next input overlaps both.  Images are independent, so there is no
data-path exchange between workers; the bytes each worker returns are exactly what a single call on its shard returns.
"""
import queue
import threading

import torch


class StreamWorkerPool:
    def __init__(self, make_codec, n_workers, device=None, priorities=None):
        """make_codec(): a NEW codec on ``device`` with update_state() done (called n_workers times).
        priorities: HIP stream priority per worker (lower = more urgent); default: all equal."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("stream workers live on a GPU")
        if self.device.index is None:   # torch.cuda.set_device() in the worker threads needs the ordinal
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n = int(n_workers)
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, 0)
        self._jobs, self._threads, self._done = [], [], queue.Queue()
        self.codecs, self.streams = [], []
        for i in range(self.n):
            pr = 0 if priorities is None else int(priorities[i])
            pr = max(min(pr, max(lo, hi)), min(lo, hi))
            self.codecs.append(make_codec())
            self.streams.append(torch.cuda.Stream(device=self.device, priority=pr))
            q = queue.Queue()
            t = threading.Thread(target=self._loop, args=(i, q), daemon=True, name=f"basic-worker-{i}")
            t.start()
            self._jobs.append(q)
            self._threads.append(t)

    def _loop(self, i, q):
        setup_error = None
        try:
            torch.cuda.set_device(self.device)
        except BaseException as e:   # reported with every job: a worker that could not start must not leave map() waiting
            setup_error = e
        with torch.cuda.stream(self.streams[i]):
            while True:
                job = q.get()
                if job is None:
                    return
                fn, arg, ticket = job
                try:
                    if setup_error is not None:
                        raise setup_error
                    out = fn(self.codecs[i], arg)
                    self.streams[i].synchronize()   # the worker's results are complete when it reports them
                    self._done.put((ticket, i, out, None))
                except BaseException as e:          # surfaced by map()
                    self._done.put((ticket, i, None, e))

    def _install_stagger(self):
        """Order the workers' ANALYSIS phases: worker i + 1 may start its job on the GPU only when worker i's analysis
        transforms are done (a HIP event the next stream waits for; a host flag tells when the event has been recorded).
        Left alone, equal workers run in lock-step -- all in their transforms together, all in their rANS chains together --
        and nothing overlaps; chained like this, worker i's chains run beside worker i + 1's transforms."""
        self._gate_evt = [torch.cuda.Event() for _ in range(self.n)]
        self._gate_flag = [threading.Event() for _ in range(self.n)]
        for i, codec in enumerate(self.codecs):
            ec = getattr(codec, "entropy_coder", None)
            if ec is None or not hasattr(ec, "after_inference_hook"):
                raise ValueError("stagger needs a latent-graph entropy coder (after_inference_hook)")

            def hook(i=i):
                if not self._gate_flag[i].is_set():     # first encode() of this job only
                    self._gate_evt[i].record(self.streams[i])
                    self._gate_flag[i].set()
            ec.after_inference_hook = hook

    def map(self, fn, shards, stagger=False):
        """Run fn(codec_i, shards[i]) on worker i for every i (len(shards) <= n), all concurrently; returns the results
        in shard order after ALL have finished (the one join of a step).  stagger: see _install_stagger."""
        assert len(shards) <= self.n
        if stagger and not hasattr(self, "_gate_evt"):
            self._install_stagger()
        if hasattr(self, "_gate_flag"):
            for f in self._gate_flag:
                f.clear() if stagger else f.set()
        for i, s in enumerate(shards):
            job = fn
            if stagger:
                def job(codec, arg, i=i, fn=fn):
                    try:
                        if i > 0:
                            self._gate_flag[i - 1].wait()
                            self.streams[i].wait_event(self._gate_evt[i - 1])
                        return fn(codec, arg)
                    finally:
                        # a job that raised, or never reached encode()'s hook, must still open the gate of the next worker
                        if not self._gate_flag[i].is_set():
                            self._gate_evt[i].record(self.streams[i])
                            self._gate_flag[i].set()
            self._jobs[i].put((job, s, i))
        out, err = [None] * len(shards), None
        for _ in shards:
            ticket, _, res, e = self._done.get()
            out[ticket] = res
            err = err or e
        if err is not None:
            raise err
        return out

    def close(self):
        for q in self._jobs:
            q.put(None)
        for t in self._threads:
            t.join(timeout=10)
        self._jobs, self._threads = [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def split_batch(x, n):
    """Contiguous shards of a batch tensor, sizes differing by at most one image."""
    B = x.shape[0]
    n = max(1, min(n, B))
    base, extra = divmod(B, n)
    out, cur = [], 0
    for i in range(n):
        k = base + (1 if i < extra else 0)
        out.append(x[cur:cur + k])
        cur += k
    return out
