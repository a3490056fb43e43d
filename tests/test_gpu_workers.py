"""Concurrent stream workers (cbench_basic_amd/benchmark/stream_workers.py): running compress / decompress of disjoint
shards on several HIP streams at once must give EXACTLY what one codec gives for the same shards one after the other."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make():
    from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
    c = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().cuda()
    c.update_state()
    return c


def _work(codec, shard):
    data = codec.compress(shard)
    return data, codec.decompress(data).cpu()


@pytest.mark.parametrize("workers", [2, 4])
def test_workers_bytes_equal_sequential(workers):
    from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool, split_batch
    torch.manual_seed(3)
    x = torch.rand(24, 3, 128, 128)
    shards = split_batch(x.cuda(), workers)
    single = _make()
    want = [_work(single, s) for s in shards]
    with StreamWorkerPool(_make, workers) as pool:
        for rep in range(6):
            got = pool.map(_work, shards)
            for i, ((gb, gx), (wb, wx)) in enumerate(zip(got, want)):
                assert gb == wb, f"rep {rep} shard {i}: bytes differ ({len(gb)} vs {len(wb)})"
                assert torch.equal(gx, wx), (rep, i)
    # host-resident (pinned) shards are uploaded by the worker itself
    hshards = [s.cpu().pin_memory() for s in shards]
    with StreamWorkerPool(_make, workers) as pool:
        got = pool.map(_work, hshards)
        assert [g[0] for g in got] == [w[0] for w in want]


def test_split_batch():
    from cbench_basic_amd.benchmark.stream_workers import split_batch
    x = torch.arange(10).reshape(10, 1)
    parts = split_batch(x, 4)
    assert [p.shape[0] for p in parts] == [3, 3, 2, 2] and torch.equal(torch.cat(parts), x)
    assert len(split_batch(x[:2], 4)) == 2
