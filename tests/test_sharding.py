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


# ---- bench.py's own main() under a world-2 gloo group, with a stand-in model ------------------------------------------------------
class _StubEncoder(torch.nn.Module):
    def __init__(self, graphs):
        super().__init__()
        self.graphs = graphs


class _StubForecaster(torch.nn.Module):
    """Stands for GraphWeatherForecaster in the launch-path test: same call signature and attributes bench.py reads
    (``encoder.graphs``), a per-sample function on the CPU (the HIP kernels cannot run here)."""

    def __init__(self, lat_lons):
        super().__init__()
        from graph_weather_amd.graphs import build_forecast_graphs

        self.encoder = _StubEncoder(build_forecast_graphs(lat_lons, 2))
        self.w = torch.nn.Parameter(torch.full((78,), 0.5))

    def forward(self, features):
        import time

        time.sleep(0.005 * (1 + int(os.environ.get("RANK", "0"))))  # rank 1 is slower: the reported time must be its time
        return features[..., :78] * self.w


def _stub_factory(cfg, dev):
    from graph_weather_amd.utils import regular_lat_lons

    lat_lons = regular_lat_lons(30.0)
    return _StubForecaster(lat_lons), lat_lons


def _bench_worker(rank: int, world: int, port: int, out_dir: str, config: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import contextlib
    import io

    import bench

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "2", "--config", config, "--no-cpu-baseline"],
                   backend="gloo", device="cpu", model_factory=_stub_factory)
    with open(os.path.join(out_dir, f"out{rank}.txt"), "w") as fh:
        fh.write(buf.getvalue())


@pytest.mark.timeout(240)
@pytest.mark.parametrize("config,per_rank,scaling", [("c2", [2, 2], "weak"), ("c4", [32, 32], "strong")])
def test_bench_main_world_two_gloo(tmp_path, config, per_rank, scaling):
    """The rank / launch / barrier / max-over-ranks / one-JSON-line path of bench.py itself: rank 0 alone prints, value is
    all ranks' forecasts over the slowest rank's time, c4 shards the global batch of 64 over the ranks."""
    import json

    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), config), nprocs=world, join=True)
    out0 = (tmp_path / "out0.txt").read_text().strip().splitlines()
    out1 = (tmp_path / "out1.txt").read_text().strip()
    assert out1 == "" and len(out0) == 1
    d = json.loads(out0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == scaling
    assert d["config"]["global_batch"] == sum(per_rank)
    assert d["unit"] == "forecasts/s" and d["cpu_baseline"] is None and d["roofline"] is None
    # slowest rank sleeps 10 ms per step: 4 steps >= 40 ms, so value <= global batch * 4 / 0.04
    assert d["ms_per_step"] >= 10.0 * 0.9
    assert abs(d["value"] - sum(per_rank) * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]


_TRAINED = {}


def _train_worker(rank: int, world: int, port: int, out_dir: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import contextlib
    import io

    import bench

    holder = {}

    def factory(cfg, dev):
        model, lat_lons = _stub_factory(cfg, dev)
        holder["model"] = model
        return model, lat_lons

    def loss_factory(lat_lons):
        return lambda pred, target: (pred - target).square().mean()

    def opt_factory(params, flat):
        holder["flat"] = flat
        return torch.optim.SGD(list(params), lr=0.5)  # stands for the one-launch AdamW kernel (updates the flat views in place)

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--mode", "train", "--config", "c2"],
                   backend="gloo", device="cpu", model_factory=factory, train_factories=(loss_factory, opt_factory))
    flat = holder["flat"]
    assert flat.views_intact() and flat.collectives == 4 * len(flat.buckets)  # every step reduced every bucket once
    torch.save(holder["model"].w.detach().clone(), os.path.join(out_dir, f"w{rank}.pt"))
    with open(os.path.join(out_dir, f"train{rank}.txt"), "w") as fh:
        fh.write(buf.getvalue())


@pytest.mark.timeout(240)
def test_bench_train_loop_world_two_gloo(tmp_path):
    """bench.py --mode train itself on two gloo ranks: flat buffer -> hooks -> bucketed all-reduce -> optimizer step.  Ranks see
    different data, so equal parameters afterwards mean the averaged gradient reached both; rank 0 alone prints one line."""
    import json

    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    assert torch.equal(w0, w1) and not torch.allclose(w0, torch.full_like(w0, 0.5))  # trained, and in lock-step
    out0 = (tmp_path / "train0.txt").read_text().strip().splitlines()
    assert (tmp_path / "train1.txt").read_text().strip() == "" and len(out0) == 1
    d = json.loads(out0[0])
    assert d["n_gpus"] == 2 and d["unit"] == "samples/s" and d["config"]["global_batch"] == 4 and d["collectives_per_step"] >= 1


def _flat_worker(rank: int, world: int, port: int, out_dir: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from graph_weather_amd import sharding as sh

    ctx = sh.init_from_env(backend="gloo")
    torch.manual_seed(0)  # same weights on every rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 3))
    ref = [p.detach().clone() for p in model.parameters()]
    flat = sh.FlatGradients(model.parameters(), bucket_bytes=256).attach(ctx)  # several buckets
    assert len(flat.buckets) >= 3 and flat.views_intact(), flat.buckets
    for p, r in zip(model.parameters(), ref):
        assert torch.equal(p, r)  # flattening keeps the values
    torch.manual_seed(100 + rank)  # different data per rank
    x, y = torch.randn(16, 6), torch.randn(16, 3)
    for it in range(2):
        flat.zero_()
        loss = (model(x) - y).square().mean()
        loss.backward()
        launched_in_backward = flat.collectives
        n = flat.allreduce(ctx)
        assert n == len(flat.buckets) and flat.views_intact()
        assert launched_in_backward >= (it + 1) * len(flat.buckets)  # every bucket's collective started from its hook
    # the averaged gradient equals the mean of the per-rank gradients computed the plain way
    import torch.distributed as dist

    model2 = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 3))
    with torch.no_grad():
        for p2, r in zip(model2.parameters(), ref):
            p2.copy_(r)
    (model2(x) - y).square().mean().backward()
    for p, p2 in zip(model.parameters(), model2.parameters()):
        g = p2.grad.clone()
        dist.all_reduce(g)
        assert torch.allclose(p.grad, g / world, atol=1e-6)
    # --- gradient accumulation: two micro-batches, the first under no_sync(); equals the gradient of the summed loss ----------
    x2, y2 = torch.randn(16, 6), torch.randn(16, 3)
    flat.zero_()
    before = flat.collectives
    with flat.no_sync():
        (model(x) - y).square().mean().backward()
    assert flat.collectives == before  # nothing launched while accumulating
    (model(x2) - y2).square().mean().backward()
    assert flat.allreduce(ctx) == len(flat.buckets)
    for p2 in model2.parameters():
        p2.grad = None
    ((model2(x) - y).square().mean() + (model2(x2) - y2).square().mean()).backward()
    for p, p2 in zip(model.parameters(), model2.parameters()):
        g = p2.grad.clone()
        dist.all_reduce(g)
        assert torch.allclose(p.grad, g / world, atol=1e-6)
    # --- a second backward into an armed step (no no_sync) must not be silently mixed into in-flight buckets -------------------
    flat.zero_()
    (model(x) - y).square().mean().backward()
    try:
        (model(x2) - y2).square().mean().backward()
        raised = False
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    assert raised
    flat.allreduce(ctx)
    # --- gradients zeroed some other way (optimizer.zero_grad(set_to_none=False) keeps the views): the next step still reduces --
    for p in model.parameters():
        p.grad.zero_()
    before = flat.collectives
    (model(x) - y).square().mean().backward()
    assert flat.allreduce(ctx) == len(flat.buckets) and flat.collectives == before + len(flat.buckets)
    for p2 in model2.parameters():
        p2.grad = None
    (model2(x) - y).square().mean().backward()
    for p, p2 in zip(model.parameters(), model2.parameters()):
        g = p2.grad.clone()
        dist.all_reduce(g)
        assert torch.allclose(p.grad, g / world, atol=1e-6)
    sh.shutdown(ctx)
    with open(os.path.join(out_dir, f"flat{rank}"), "w") as fh:
        fh.write("ok")


@pytest.mark.timeout(180)
def test_flat_gradient_buckets_overlap_world_two_gloo(tmp_path):
    """FlatGradients: parameters / gradients as views of flat buffers, bucketed all-reduce launched from the backward."""
    world = 2
    mp.spawn(_flat_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"flat{r}").exists() for r in range(world))


@pytest.mark.timeout(300)
def test_bench_self_launch_from_a_clean_environment(tmp_path):
    """`python bench.py --gpus 2 ...` as the driver would type it on an 8-GPU node - NO launcher, no RANK / WORLD_SIZE in the
    environment: bench.py forks its own ranks (bench.self_launch), rank 0 prints the one JSON line, the exit code is 0.  Runs the
    real command line in a subprocess; the GW_BENCH_* hooks put it on gloo / CPU with the stand-in model."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env.update(GW_BENCH_BACKEND="gloo", GW_BENCH_DEVICE="cpu", GW_BENCH_FACTORY="tests.test_sharding:_stub_factory",
               PYTHONPATH=root + os.pathsep + env.get("PYTHONPATH", ""))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4  # c2: batch 2 on each of the two ranks
    # what a SCALE record can check: the process group really had 2 ranks, each reported its own clock, value uses the slowest
    assert d["dist_world_size"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) <= 1e-6 * d["ms_per_step"]
    # the stand-in hooks are refused on anything but the CPU device (a stand-in model must never print the driver's GPU line)
    env_gpu = dict(env)
    env_gpu.pop("GW_BENCH_DEVICE")
    r = subprocess.run(cmd, cwd=root, env=env_gpu, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0 and "GW_BENCH_DEVICE=cpu" in (r.stderr + r.stdout)
    # a failing rank makes the launcher exit non-zero (bad factory -> every rank raises)
    env["GW_BENCH_FACTORY"] = "tests.test_sharding:_failing_factory"
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0


_FAIL_CALLS = {"n": 0}


def _failing_factory(cfg, dev):
    """Works in the launcher (which builds the graphs once through it), raises in the ranks."""
    if "WORLD_SIZE" in os.environ:
        raise RuntimeError("rank failure injected by the test")
    return _stub_factory(cfg, dev)


def test_forecast_graphs_are_built_once_per_grid():
    from graph_weather_amd.graphs import build_forecast_graphs
    from graph_weather_amd.utils import regular_lat_lons

    ll = regular_lat_lons(30.0)
    a, b = build_forecast_graphs(ll, 2), build_forecast_graphs(list(ll), 2)
    assert a is b
    assert build_forecast_graphs(regular_lat_lons(20.0), 2) is not a


def test_flat_gradients_world_one_second_step_with_zero_grad_in_place():
    """a single-process FlatGradients whose gradients are zeroed with optimizer.zero_grad(set_to_none=False) (not
    flat.zero_()) must train for more than one step, attached to a world-1 context or not; local accumulation likewise."""
    for attach in (False, True):
        torch.manual_seed(0)
        lin = torch.nn.Linear(4, 3)
        flat = sharding.FlatGradients(lin.parameters(), bucket_bytes=16)
        ctx = sharding.ShardContext(0, 0, 1, None)
        if attach:
            flat.attach(ctx)
        opt = torch.optim.SGD(lin.parameters(), lr=0.1)
        for step in range(3):
            opt.zero_grad(set_to_none=False)
            lin(torch.randn(5, 4)).square().mean().backward()
            lin(torch.randn(5, 4)).square().mean().backward()  # accumulation without no_sync(): fine with one rank
            assert flat.allreduce(ctx) == 0
            opt.step()
        assert flat.views_intact()
