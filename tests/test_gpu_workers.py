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


@pytest.mark.parametrize("batch,size,lanes", [(32, 256, 4), (32, 256, 2), (256, 256, 4), (256, 256, 1)])
def test_bench_schedule_bytes_equal_quiet_single_stream(batch, size, lanes):
    """What bench.py TIMES: six stream workers, every step a whole batch (step k on worker k mod 6, six batches in flight), the
    transform token with `lanes` sessions admitted at a time, eight image streams per rANS workgroup, the fused C entry points.
    Every stream and every reconstruction any worker returns in any of 12 steps must equal what ONE quiet codec -- no workers, no
    token, default rANS packing -- returns for the same batch: the overlapped schedule is race-free or this fails."""
    from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool
    from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(batch + lanes)
    x = torch.rand(batch, 3, size, size, generator=g).to(dev)
    quiet = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to(dev)
    quiet.update_state()
    want = quiet.compress(x)
    want_x = quiet.decompress(want)

    def make():
        c = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to(dev)
        c.update_state()
        c.entropy_coder.fused_rans_waves = 8
        c.entropy_coder.fused_transform_token = lanes
        return c
    workers, steps = 6, 12
    counts = [len(range(w, steps, workers)) for w in range(workers)]

    def loop(c, n):
        outs = []
        for _ in range(n):
            data = c.compress(x)
            xh = c.decompress(data)
            outs.append((data, torch.equal(xh, want_x)))
        return outs
    with StreamWorkerPool(make, workers, dev) as pool:
        for rep in range(2):
            got = pool.map(loop, counts)
            for w, outs in enumerate(got):
                assert len(outs) == counts[w]
                for k, (data, same_x) in enumerate(outs):
                    assert data == want, f"rep {rep} worker {w} step {k}: bytes differ from the quiet codec's ({len(data)} vs {len(want)})"
                    assert same_x, f"rep {rep} worker {w} step {k}: reconstruction differs from the quiet codec's"
