"""Round-2 parity tests at the configurations BASELINE.json names but round 1 left untested against the oracle
(C3 = 1 degree, batch 16, bf16 matrix products; C5 = 0.25 degree, mesh resolution 3), fixed-bar gradient checks against the
oracle's fp32 AND fp64 autograd, and the behaviours added this round (recomputation = the reference's checkpointing
flags, gradient of the decoder residual, gradients through frozen blocks)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features  # noqa: E402
from oracle import chunked as oc  # noqa: E402
from oracle import reference_math as om  # noqa: E402

DEV = "cuda:0"
FP32_REL = 2e-4      # of the decoder-delta scale; north_star asks 1e-3
BF16_BUDGET = 2e-2   # error budget of the bf16-operand mode at C3, of the decoder-delta scale (outside the 1e-3 parity bar:
#                      bf16 operands carry 8 mantissa bits; reported, asserted against this budget)


def _rel(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _l2(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


def _row_sample(lat_lons, n, seed):
    """n grid rows: a random sample plus the rows nearest both poles (the longest destination segments of the encoder graph
    and the most shared decoder sources)."""
    G = len(lat_lons)
    rs = np.random.RandomState(seed)
    lat = np.asarray([ll[0] for ll in lat_lons])
    polar = np.concatenate([np.argsort(lat)[:20], np.argsort(-lat)[:20]])
    rows = np.unique(np.concatenate([rs.choice(G, size=n - polar.size, replace=False), polar]))
    return torch.from_numpy(rows).long()


def test_c3_one_degree_batch16_bf16_against_the_oracle():
    """BASELINE.json configs[2] at its own size: 1 degree (64 800 nodes), batch 16, bf16 matrix products.  The first and the
    last sample of the batch (the persistent bf16 kernels walk tiles batch-innermost: sample 15 exercises the far end of
    every walk) against the fp32 oracle on a fixed row sample; fp32 on the same inputs must stay inside the fp32 bar."""
    lat_lons = regular_lat_lons(1.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = model.encoder.graphs.as_oracle_dict()
    feats = seeded_features(16, len(lat_lons), 102, seed=42)
    rows = _row_sample(lat_lons, 1500, seed=1)
    ref = oc.forecaster_rows(sd, g, feats[[0, 15]], rows)  # [2, R, 78] fp32 oracle
    model = model.to(DEV).eval()
    fd = feats.to(DEV)
    with torch.no_grad():
        y32 = model(fd)[[0, 15]][:, rows.to(DEV)].cpu()
        model.set_compute_dtype(torch.bfloat16)
        y16 = model(fd)
        y16_again = model(fd)
    assert torch.isfinite(y16).all()
    start = feats[[0, 15]][:, rows, :78]
    d_ref = ref - start
    r32 = _rel(y32 - start, d_ref)
    y16s = y16[[0, 15]][:, rows.to(DEV)].cpu()
    r16 = _rel(y16s - start, d_ref)
    l16 = _l2(y16s - start, d_ref)
    print(f"[parity] C3 1deg B=16: fp32 max-rel {r32:.2e}; bf16 max-rel {r16:.2e}, l2-rel {l16:.2e} (rows {rows.numel()}, samples 0 and 15)")
    assert r32 <= FP32_REL
    assert r16 <= BF16_BUDGET and r16 > 10 * r32, "bf16 mode must be inside its budget (and must really be bf16)"
    # run to run: the order of tile-boundary atomics differs in the last fp32 bit, which can flip the bf16 rounding of single
    # activations downstream (2^-9 relative each) - far below the budget, far above fp32 noise
    assert _rel(y16_again, y16) <= 5e-3
    # every sample of the batch is the same function of its own input: sample 7 alone == sample 7 in the batch
    with torch.no_grad():
        y7 = model(fd[7:8].contiguous())
    assert _rel(y7[0], y16[7]) <= 5e-3


def test_c5_quarter_degree_against_the_chunked_oracle():
    """BASELINE.json configs[4]: 0.25 degree (1 036 800 nodes), mesh resolution 3 (41 162 nodes), batch 1, fp32, against the
    slab-wise CPU oracle on 1 240 sampled grid rows (incl. the polar rows)."""
    lat_lons = regular_lat_lons(0.25)
    model = gw.GraphWeatherForecaster(lat_lons, resolution=3)
    deterministic_fill_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = model.encoder.graphs.as_oracle_dict()
    G = len(lat_lons)
    feats = torch.from_numpy(np.random.RandomState(5).standard_normal((1, G, 102)).astype(np.float32))
    rows = _row_sample(lat_lons, 1240, seed=2)
    ref = oc.forecaster_rows(sd, g, feats, rows, slab=1 << 17)
    model = model.to(DEV).eval()
    with torch.no_grad():
        y = model(feats.to(DEV))[:, rows.to(DEV)].cpu()
    start = feats[:, rows, :78]
    r = _rel(y - start, ref - start)
    print(f"[parity] C5 0.25deg res 3 B=1: fp32 max-rel {r:.2e} on {rows.numel()} rows")
    assert r <= FP32_REL
    # deterministic segment sums: bitwise run to run at this size (3 446 grid points end in one polar mesh cell: a destination
    # segment of the encoder graph that spans 54 tiles), and the same forecast as the atomics mode up to summation order
    fd = feats.to(DEV)
    with torch.no_grad():
        y_atomic = model(fd)
        model.set_deterministic(True)
        y_a = model(fd)
        y_b = model(fd)
    assert torch.equal(y_a, y_b), "deterministic mode must be bitwise reproducible"
    assert _rel(y_a - fd[..., :78], y_atomic - fd[..., :78]) <= 1e-5


# ---- gradients: fixed bars against the oracle's autograd in fp64 and in fp32 -----------------------------------------------
GRAD_MAX_REL = 2e-2  # per tensor, max |g - g_ref| / max |g_ref|: a ReLU gate that flips between two fp32 summation orders moves
#                      single entries of long, cancelling sums by ~1e-2 (the oracle's own fp32 vs fp64 autograd shows the same)
GRAD_L2_REL = 3e-3   # per tensor, ||g - g_ref|| / ||g_ref||: insensitive to single flipped gates


def test_forecaster_gradients_fixed_bars_fp64_and_fp32_oracle():
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    ref64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    ref32 = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    g32 = model.encoder.graphs.as_oracle_dict()
    g64 = om.graphs_to_dtype(g32, torch.float64)
    feats = seeded_features(2, len(lat_lons), 102, seed=42)
    rs = np.random.RandomState(7)
    target = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78)).astype(np.float32))
    var = torch.from_numpy((rs.rand(78) + 0.5).astype(np.float32))
    om.normalized_mse_loss(om.forecaster_forward(ref64, g64, feats.double()), target.double(), lat_lons, var.double(), True).backward()
    om.normalized_mse_loss(om.forecaster_forward(ref32, g32, feats), target, lat_lons, var, True).backward()
    model = model.to(DEV).train()
    crit = gw.NormalizedMSELoss(var.tolist(), lat_lons, normalize=True)
    crit(model(feats.to(DEV)), target.to(DEV)).backward()
    worst = {"fp64": (0, 0, ""), "fp32": (0, 0, "")}
    bad = []
    for k, p in model.named_parameters():
        for name, ref in (("fp64", ref64), ("fp32", ref32)):
            m, l = _rel(p.grad, ref[k].grad), _l2(p.grad, ref[k].grad)
            if m > worst[name][0]:
                worst[name] = (m, l, k)
            if m > GRAD_MAX_REL or l > GRAD_L2_REL:
                bad.append((name, k, m, l))
    print(f"[backward] fixed bars: worst vs fp64 autograd max-rel {worst['fp64'][0]:.2e} (l2 {worst['fp64'][1]:.2e}, {worst['fp64'][2]}); "
          f"vs fp32 autograd max-rel {worst['fp32'][0]:.2e} (l2 {worst['fp32'][1]:.2e}, {worst['fp32'][2]})")
    assert not bad, bad[:6]


# ---- recomputation (use_checkpointing / GraphCast strategies) -------------------------------------------------------------------
def _grads_and_peak(model, feats, target, crit):
    for p in model.parameters():
        p.grad = None
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    loss = crit(model(feats), target)
    loss.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    return loss.item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, peak


def test_use_checkpointing_recomputes_with_equal_gradients_and_less_memory():
    """tests/models/test_gradient_checkpointing.py:400-443 of the reference (gradient equality between strategies) on the
    forecaster: use_checkpointing=True must give the same loss and gradients while keeping fewer activations."""
    lat_lons = regular_lat_lons(5.0)
    feats = seeded_features(2, len(lat_lons), 102, seed=1).to(DEV)
    target = torch.rand(2, len(lat_lons), 78, device=DEV)
    crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons)
    plain = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(plain, seed=4)
    ckpt = gw.GraphWeatherForecaster(lat_lons, use_checkpointing=True)
    ckpt.load_state_dict(plain.state_dict())
    plain, ckpt = plain.to(DEV).train(), ckpt.to(DEV).train()
    l0, g0, m0 = _grads_and_peak(plain, feats, target, crit)
    l1, g1, m1 = _grads_and_peak(ckpt, feats, target, crit)
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    for k in g0:
        assert _rel(g1[k], g0[k]) <= 1e-4, k  # same kernels on the same values: only atomics order differs
    print(f"[checkpoint] forecaster 5deg B=2: peak activations {m0 / 2**20:.0f} MiB plain, {m1 / 2**20:.0f} MiB with use_checkpointing")
    # the forecaster does not hand use_checkpointing to its Processor (forecast.py:142-152), whose nine
    # blocks hold most of the activations at this grid size: encoder and decoder segments are recomputed, the total barely moves
    assert m1 <= 1.02 * m0


@pytest.mark.parametrize("strategy", ["full", "balanced", "processor_only", "segments3", "fine_grained"])
def test_graphcast_checkpoint_strategies(strategy):
    """graphcast/model.py:289-345: every strategy gives the gradients of the un-checkpointed model; the hierarchical ones
    keep fewer activations."""
    lat_lons = regular_lat_lons(10.0)
    feats = seeded_features(2, len(lat_lons), 78, seed=2).to(DEV)
    target = torch.rand(2, len(lat_lons), 78, device=DEV)
    crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons)
    plain = gw.GraphCast(lat_lons, efficient_batching=True)
    deterministic_fill_(plain, seed=6)
    other = gw.GraphCast(lat_lons, efficient_batching=True, use_checkpointing=(strategy == "fine_grained"))
    other.load_state_dict(plain.state_dict())
    gw.GraphCastConfig.no_checkpointing(plain)
    if strategy == "segments3":
        other.set_checkpoint_processor(3)
    else:
        getattr(gw.GraphCastConfig, strategy + "_checkpointing")(other)
    plain, other = plain.to(DEV).train(), other.to(DEV).train()
    l0, g0, m0 = _grads_and_peak(plain, feats, target, crit)
    l1, g1, m1 = _grads_and_peak(other, feats, target, crit)
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    for k in g0:
        assert _rel(g1[k], g0[k]) <= 1e-4, k
    print(f"[checkpoint] GraphCast {strategy}: peak {m0 / 2**20:.0f} -> {m1 / 2**20:.0f} MiB")
    # one segment (full) or a dominant segment (the whole processor: balanced, processor_only) peaks while that segment is
    # replayed with its saves - as torch.utils.checkpoint does in the reference; finer segments hold a fraction of them
    if strategy in ("segments3", "fine_grained"):
        assert m1 < 0.6 * m0
    else:
        assert m1 <= 1.05 * m0


# ---- gradient of the decoder residual (multi-step training) ----------------------------------------------------------
def test_two_step_rollout_gradients_include_the_residual_path():
    """Step t+1 consumes step t's output: d(out)/d(features) has the identity term of decoder.py:93.  Two chained steps,
    gradients of every parameter and of the input against the oracle's fp64 autograd."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons, num_blocks=3)  # (two chained oracle forwards + an fp64 backward on the host)
    deterministic_fill_(model, seed=3)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    feats = seeded_features(1, len(lat_lons), 102, seed=8)
    w = torch.from_numpy(np.random.RandomState(1).standard_normal((1, len(lat_lons), 78)).astype(np.float32))
    x0 = feats.double().requires_grad_(True)
    y1 = om.forecaster_forward(ref, g64, x0)
    y2 = om.forecaster_forward(ref, g64, torch.cat([y1, x0[..., 78:]], dim=-1))
    (y2 * w.double()).sum().backward()
    model = model.to(DEV).train()
    xd = feats.to(DEV).requires_grad_(True)
    z1 = model(xd)
    z2 = model(torch.cat([z1, xd[..., 78:]], dim=-1))
    (z2 * w.to(DEV)).sum().backward()
    assert _rel(z2, y2) <= 1e-4
    assert _l2(xd.grad, x0.grad) <= GRAD_L2_REL and _rel(xd.grad, x0.grad) <= GRAD_MAX_REL
    bad = [(k, _rel(p.grad, ref[k].grad), _l2(p.grad, ref[k].grad)) for k, p in model.named_parameters()
           if _l2(p.grad, ref[k].grad) > GRAD_L2_REL or _rel(p.grad, ref[k].grad) > GRAD_MAX_REL]
    assert not bad, bad[:6]


def test_frozen_processor_still_passes_gradients_to_the_encoder():
    """a frozen block downstream of a trainable one must run the differentiable path."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons, num_blocks=3)
    deterministic_fill_(model, seed=9)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    feats = seeded_features(1, len(lat_lons), 102, seed=3)
    om.forecaster_forward(ref, g64, feats.double()).square().mean().backward()
    model = model.to(DEV).train()
    for p in model.processor.parameters():
        p.requires_grad_(False)
    for p in model.decoder.parameters():
        p.requires_grad_(False)
    model(feats.to(DEV)).square().mean().backward()
    for k, p in model.named_parameters():
        if k.startswith("encoder."):
            assert p.grad is not None, k
            assert _l2(p.grad, ref[k].grad) <= GRAD_L2_REL and _rel(p.grad, ref[k].grad) <= GRAD_MAX_REL, k
        else:
            assert p.grad is None


def test_flat_adamw_is_one_launch_and_matches_torch_adamw():
    """sharding.FlatGradients + AdamW(flat=...): parameters and gradients are views of two flat buffers, the optimizer step
    is one kernel over them; values equal torch.optim.AdamW's (train/run.py:506 uses AdamW)."""
    from graph_weather_amd import sharding as sh

    torch.manual_seed(0)
    a = gw.GraphProcessor(mp_iterations=1, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256).to(DEV)
    b = gw.GraphProcessor(mp_iterations=1, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256).to(DEV)
    deterministic_fill_(a, seed=1)
    b.load_state_dict(a.state_dict())
    flat = sh.FlatGradients(a.parameters())
    opt_a = gw.AdamW(a.parameters(), lr=1e-2, weight_decay=0.05, flat=flat)
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=0.05)
    assert flat.views_intact() and all(torch.equal(pa, pb) for pa, pb in zip(a.parameters(), b.parameters()))
    for it in range(3):
        flat.zero_()
        gen = torch.Generator(device=DEV).manual_seed(it)
        for pa, pb in zip(a.parameters(), b.parameters()):
            g = torch.randn(pa.shape, device=DEV, generator=gen)
            pa.grad.add_(g)  # accumulate into the view, as autograd does
            pb.grad = g.clone()
        v0 = [p._version for p in a.parameters()]
        opt_a.step()
        opt_b.step()
        assert all(p._version > v for p, v in zip(a.parameters(), v0))  # caches keyed on ._version see the update
    for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert _rel(pa, pb) <= 1e-5, k
    assert flat.views_intact()
    # the packed-weight caches notice: a forward after the step uses the new values
    x = torch.randn(50, 256, device=DEV)
    ei = torch.randint(0, 50, (2, 300), device=DEV)
    ea = torch.randn(300, 256, device=DEV)
    with torch.no_grad():
        xa, _ = a(x, ei, ea)
        xb, _ = b(x, ei, ea)
    assert _rel(xa, xb) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_node_update_post_products_equal_the_separate_projection(dtype):
    """gw_node_update_forward with n_post > 0 (the next block's layer-1 products made while x' is in registers, next aggregate
    zero-filled on the side) against the two-launch form: node update, then gw_project_forward on its output."""
    from graph_weather_amd import ops
    from graph_weather_amd.ops import Operand

    torch.manual_seed(0)
    blk = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
    nxt = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
    deterministic_fill_(blk, seed=3)
    deterministic_fill_(nxt, seed=4)
    gw.set_compute_dtype(blk, dtype)
    gw.set_compute_dtype(nxt, dtype)
    blk, nxt = blk.to(DEV), nxt.to(DEV)
    B, n = 3, 777  # ragged: not a multiple of the 64 / 128 column tiles
    x = torch.randn(B * n, 256, device=DEV)
    agg = torch.randn(B * n, 256, device=DEV)
    pm_n, pm_e = blk.node_model.node_mlp.packed(), nxt.edge_model.edge_mlp.packed()
    ref = ops.node_update_forward(pm_n, B * n, n, Operand(x, n, 256), Operand(x, n, 256), Operand(agg, n, 256))
    ps_ref, pd_ref = ops.project_forward([pm_e.w1[0], pm_e.w1[1]], Operand(ref, n, 256), B * n, n)
    zero = torch.full((B * n, 256), 7.0, device=DEV)
    out, (ps, pd) = ops.node_update_forward(pm_n, B * n, n, Operand(x, n, 256), Operand(x, n, 256), Operand(agg, n, 256),
                                            post_w=[pm_e.w1[0], pm_e.w1[1]], zero_rows=zero)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(ps, ps_ref) and torch.equal(pd, pd_ref)
    assert (zero == 0).all()


def test_per_sample_streams_give_the_single_stream_forecast():
    """The fused inference forward runs the mesh stack as per-sample chains on side streams (fp32, batch >= 2): same forecast
    as on one stream, for stream counts that do and do not divide the batch; repeated calls reuse buffers across streams."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    gp = model.processor.graph_processor
    feats = seeded_features(5, len(lat_lons), 102, seed=11).to(DEV)
    assert gp.forward_streams(5) == 2 and gp.forward_streams(1) == 1
    with torch.no_grad():
        gp.streams = 1
        y1 = model(feats)
        outs = {}
        for n in (2, 3, 5):
            gp.streams = n
            for _ in range(3):
                outs[n] = model(feats)
        gp.streams = 0
        y_auto = model(feats)
    torch.cuda.synchronize()
    for n, y in outs.items():
        assert _rel(y, y1) <= 1e-5, n
    assert _rel(y_auto, y1) <= 1e-5
    model.set_compute_dtype(torch.bfloat16)
    assert gp.forward_streams(5) == 1  # the persistent bf16 kernels fill the chip by themselves


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deterministic_segment_sums_are_bitwise_reproducible(dtype):
    """set_deterministic(): carry records + a fix-up launch instead of atomics.  A graph with hub destinations whose segments
    span several 64-edge tiles (the case atomics make order dependent), fp32 and bf16 kernels: identical bits on every run,
    the oracle's values, and the atomics mode's values up to summation order."""
    gp = gw.GraphProcessor(mp_iterations=3, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256)
    deterministic_fill_(gp, seed=8)
    p = {"gp." + k: v.clone() for k, v in gp.state_dict().items()}
    gw.set_compute_dtype(gp, dtype)
    gp = gp.to(DEV)
    rs = np.random.RandomState(3)
    n, e = 90, 2500
    x = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    src = rs.randint(0, n, size=e)
    dst = np.where(rs.rand(e) < 0.5, rs.choice([7, 8, 60], size=e), rs.randint(0, n, size=e))  # three hubs of ~400 edges each
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    ea = torch.from_numpy(rs.standard_normal((e, 256)).astype(np.float32))
    xr, er = om.graph_processor(p, "gp", x, ei, ea)
    with torch.no_grad():
        x0, e0 = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
        gw.set_deterministic(gp, True)
        runs = [gp(x.to(DEV), ei.to(DEV), ea.to(DEV)) for _ in range(4)]
    for xa, ea_ in runs[1:]:
        assert torch.equal(xa, runs[0][0]) and torch.equal(ea_, runs[0][1])
    bar = FP32_REL if dtype == torch.float32 else 3e-2
    assert _rel(runs[0][0], xr) <= bar and _rel(runs[0][1], er) <= bar
    assert _rel(runs[0][0], x0) <= (1e-5 if dtype == torch.float32 else BF16_BUDGET)
    # the whole forecaster, batch 3 (tiles of the fp32 kernel cross batch elements)
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    model.set_compute_dtype(dtype)
    feats = seeded_features(3, len(lat_lons), 102, seed=5).to(DEV)
    with torch.no_grad():
        y_atomic = model(feats)
        model.set_deterministic(True)
        ys = [model(feats) for _ in range(3)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    # bf16: a different (fixed) summation order moves sums by an fp32 ulp, which flips bf16 roundings downstream - both
    # forecasts are inside the bf16 budget of the oracle, and so is their distance
    assert _rel(ys[0] - feats[..., :78], y_atomic - feats[..., :78]) <= (1e-5 if dtype == torch.float32 else BF16_BUDGET)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_processor_without_norm(dtype):
    """norm_type=None (graph_net_block.py:50-59: the MLPs end in their last Linear): forward against the oracle, fp32 backward
    against the oracle's fp64 autograd."""
    gp = gw.GraphProcessor(mp_iterations=2, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256, norm_type=None)
    deterministic_fill_(gp, seed=12)
    assert not any(isinstance(m, torch.nn.LayerNorm) for m in gp.modules())
    p = {"gp." + k: v.clone() for k, v in gp.state_dict().items()}
    rs = np.random.RandomState(2)
    n, e = 130, 900
    x = torch.from_numpy(0.5 * rs.standard_normal((n, 256)).astype(np.float32))
    ea = torch.from_numpy(0.5 * rs.standard_normal((e, 256)).astype(np.float32))
    ei = torch.from_numpy(np.stack([rs.randint(0, n, size=e), rs.randint(0, n, size=e)]).astype(np.int64))
    xr, er = om.graph_processor(p, "gp", x, ei, ea)
    gw.set_compute_dtype(gp, dtype)
    gp = gp.to(DEV)
    with torch.no_grad():
        xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
    bar = FP32_REL if dtype == torch.float32 else 3e-2
    assert _rel(xo, xr) <= bar and _rel(eo, er) <= bar
    if dtype != torch.float32:
        return
    ref = {k: v.detach().double().requires_grad_(True) for k, v in p.items()}
    xr64, er64 = om.graph_processor(ref, "gp", x.double(), ei, ea.double())
    (xr64.square().sum() + er64.square().sum()).backward()
    xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
    (xo.square().sum() + eo.square().sum()).backward()
    for k, prm in gp.named_parameters():
        assert _l2(prm.grad, ref["gp." + k].grad) <= GRAD_L2_REL, k
