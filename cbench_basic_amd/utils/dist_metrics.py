"""Image sharding and metric reduction for multi-GPU runs (one process per GPU).

The encode/decode path shards by independent images -- image i goes to rank i mod world, like the
reference's own parallel test mode maps disjoint index ranges to workers
(cbench/benchmark/basic_benchmark.py:851-858) -- so the DATA path has no collective at all.  The only
exchange is the end-of-run reduction of metric sums, the analogue of reduce_across_processes
(cbench/utils/logging_utils.py:458-465): one all-reduce of a few float64 (RCCL over xGMI on the GPU
node, gloo in CPU tests).
"""
from typing import Dict, List

import torch


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin ownership: item i belongs to rank i % world_size."""
    return list(range(rank, n_items, world_size))


def reduce_metric_sums(sums: Dict[str, float], device=None, time_key: str = "time_s") -> Dict[str, float]:
    """All-reduce a dict of per-rank SUMS (count, bytes, psnr_sum, ...).  ``time_key`` is reduced with MAX
    (the job is as slow as its slowest rank); everything else with SUM.  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(sums)
    keys = sorted(k for k in sums if k != time_key)
    vec = torch.tensor([float(sums[k]) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    out = {k: float(v) for k, v in zip(keys, vec.cpu())}
    if time_key in sums:
        t = torch.tensor([float(sums[time_key])], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[time_key] = float(t.cpu()[0])
    return out


def gather_per_image(values: torch.Tensor, n_items: int, rank: int, world_size: int) -> torch.Tensor:
    """Gather per-image metrics (e.g. bytes, PSNR) of a round-robin sharded set back into item order on every rank.
    values: float64 [len(shard_indices(n_items, rank, world_size)), k]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world_size == 1:
        return values
    per = (n_items + world_size - 1) // world_size
    pad = torch.zeros((per, values.shape[1]), dtype=values.dtype, device=values.device)
    pad[: values.shape[0]] = values
    bufs = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(bufs, pad)
    out = torch.zeros((n_items, values.shape[1]), dtype=values.dtype, device=values.device)
    for r in range(world_size):
        idx = shard_indices(n_items, r, world_size)
        out[idx] = bufs[r][: len(idx)]
    return out
