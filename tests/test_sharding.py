"""Multi-rank path on CPU: world_size-2 gloo process group, the same code bench.py runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from graph_weather_amd import sharding


def test_shard_ranges_partition_the_batch():
    for gb in (0, 1, 2, 7, 16, 64):
        for world in (1, 2, 3, 4, 8):
            ranges = [sharding.shard_range(gb, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == gb
            for (lo0, hi0), (lo1, hi1) in zip(ranges, ranges[1:]):
                assert hi0 == lo1
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(64, 8, 3) == (24, 32)  # BASELINE configs[3]: 64 over 8 GPUs = 8 per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, world: int, port: int, out_dir: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time

    from graph_weather_amd import sharding as sh

    ctx = sh.init_from_env(backend="gloo")
    assert ctx.world == world and ctx.rank == rank and ctx.backend == "gloo"
    # every rank owns a slice of one global batch; "forward" of a shard = a per-sample function with no exchange
    global_batch = 5
    feats = torch.arange(global_batch * 3, dtype=torch.float32).reshape(global_batch, 3)
    lo, hi = sh.shard_range(global_batch, world, rank)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one: the reported time must be ITS time
        return feats[lo:hi] * 2.0

    elapsed = sh.timed_steps(ctx, step, steps=3, warmup=1)
    assert len(calls) == 4
    assert elapsed >= 3 * 0.01 * world * 0.9  # max over ranks, not this rank's own time
    rate = sh.whole_job_rate(ctx, (hi - lo) * 3, elapsed)
    assert abs(rate - global_batch * 3 / elapsed) < 1e-9  # all ranks' units / slowest time
    assert sh.max_over_ranks(ctx, float(rank)) == float(world - 1)
    assert sh.sum_over_ranks(ctx, 1.0) == float(world)
    # the shards together are the whole batch (gather only for the check - the product path has no collective)
    import torch.distributed as dist

    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, (feats[lo:hi] * 2.0).tolist()))
    whole = torch.cat([torch.tensor(p[2]).reshape(-1, 3) for p in sorted(parts)])
    assert torch.equal(whole, feats * 2.0)
    # data-parallel gradient step: every rank ends with the mean gradient, bucketed into flat collectives
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(5, 3), (7,), (2, 2, 2), (1,)]]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    n = sh.allreduce_gradients(ctx, params, bucket_bytes=64)  # small buckets: several collectives
    assert n >= 2
    for i, p in enumerate(params):
        assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * (world + 1) / 2.0))
    sh.shutdown(ctx)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as fh:
        fh.write("ok")


@pytest.mark.timeout(180)
def test_world_size_two_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
