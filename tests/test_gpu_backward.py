"""Backward pass of the hot path against the oracle's autograd in fp64 on the CPU.
Gradients are compared per tensor relative to that tensor's largest reference entry (fp32 arithmetic in a
different summation order; weight gradients are long sums with cancellation, so the bar is 2e-3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features  # noqa: E402
from oracle import reference_math as om  # noqa: E402

DEV = "cuda:0"
REL = 2e-3  # gradients are sums over thousands of columns in fp32: looser than the forward bar


def _rel(a, ref):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _l2(a, ref):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


def _check_param_grads(module, ref_params, prefix, worst, bar=REL):
    bad = []
    for k, p in module.named_parameters():
        r = ref_params[prefix + k].grad
        assert p.grad is not None, f"no gradient for {k}"
        worst[k] = (_rel(p.grad, r), _l2(p.grad, r))
        if worst[k][0] >= bar:
            bad.append((k, worst[k]))
    assert not bad, f"{len(bad)} gradients off: " + ", ".join(f"{k}: max {m:.2e} l2 {l:.2e}" for k, (m, l) in bad[:8])


@pytest.mark.parametrize("i,o,h,norm,rows", [(102, 256, 256, "LayerNorm", 700), (2, 256, 256, "LayerNorm", 300), (256, 78, 128, None, 515)])
def test_mlp_backward(i, o, h, norm, rows):
    m = gw.MLP(i, o, h, 2, norm)
    deterministic_fill_(m, seed=31)
    rs = np.random.RandomState(rows)
    x = torch.from_numpy(rs.standard_normal((rows, i)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((rows, o)).astype(np.float32))
    ref = {"m." + k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    om.mlp(ref, "m", xr).backward(dy.double())
    m = m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    y.backward(dy.to(DEV))
    worst = {}
    _check_param_grads(m, ref, "m.", worst)
    assert _rel(xd.grad, xr.grad) < REL


def test_graph_processor_backward_random_coo():
    gp = gw.GraphProcessor(mp_iterations=2, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256)
    deterministic_fill_(gp, seed=8)
    ref = {"gp." + k: v.detach().double().requires_grad_(True) for k, v in gp.state_dict().items()}
    rs = np.random.RandomState(3)
    n, e = 200, 1500
    x = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    ea = torch.from_numpy(rs.standard_normal((e, 256)).astype(np.float32))
    src = rs.randint(0, n, size=e)
    dst = np.where(rs.rand(e) < 0.2, 7, rs.randint(0, n, size=e))
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    gx = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    ge = torch.from_numpy(rs.standard_normal((e, 256)).astype(np.float32))
    xr, er = x.double().requires_grad_(True), ea.double().requires_grad_(True)
    xo_r, eo_r = om.graph_processor(ref, "gp", xr, ei, er)
    ((xo_r * gx.double()).sum() + (eo_r * ge.double()).sum()).backward()
    gp = gp.to(DEV)
    xd, ed = x.to(DEV).requires_grad_(True), ea.to(DEV).requires_grad_(True)
    xo, eo = gp(xd, ei.to(DEV), ed)
    assert _rel(xo, xo_r) < REL and _rel(eo, eo_r) < REL
    ((xo * gx.to(DEV)).sum() + (eo * ge.to(DEV)).sum()).backward()
    worst = {}
    _check_param_grads(gp, ref, "gp.", worst)
    assert _rel(xd.grad, xr.grad) < REL
    assert _rel(ed.grad, er.grad) < REL


def test_forecaster_training_step_gradients_10deg():
    """loss.backward() of the whole model (pattern of the reference's tests/test_model.py:157-172) vs the oracle."""
    lat_lons = regular_lat_lons(10.0)
    # 3 processor blocks: the oracle's fp64 backward on the host is what this test spends its time on, and all 9 blocks are
    # checked with fixed bars in tests/test_gpu_round2.py - this one is about the loss (variance-normalised) at the end
    model = gw.GraphWeatherForecaster(lat_lons, num_blocks=3)
    deterministic_fill_(model, seed=0)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    feats = seeded_features(2, len(lat_lons), 102, seed=42)
    rs = np.random.RandomState(7)
    target = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78)).astype(np.float32))
    var = torch.from_numpy((rs.rand(78) + 0.5).astype(np.float32))
    y_ref = om.forecaster_forward(ref, g64, feats.double())
    loss_ref = om.normalized_mse_loss(y_ref, target.double(), lat_lons, var.double(), normalize=True)
    loss_ref.backward()
    model = model.to(DEV).train()
    crit = gw.NormalizedMSELoss(var.tolist(), lat_lons, normalize=True)
    y = model(feats.to(DEV))
    loss = crit(y, target.to(DEV))
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
    loss.backward()
    # context: how far the oracle's own fp32 autograd is from its fp64 autograd (ReLU gates that flip between the two
    # precisions change a few gradient paths - the HIP path is asked to be no worse than a few times that, or 2e-3)
    ref32 = {k: v.detach().float().requires_grad_(True) for k, v in ref.items()}
    g32 = model.encoder.graphs.as_oracle_dict()
    om.normalized_mse_loss(om.forecaster_forward(ref32, g32, feats), target, lat_lons, var, normalize=True).backward()
    noise = max(_rel(ref32[k].grad, ref[k].grad) for k in ref)
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=max(REL, 4 * noise))
    k = max(worst, key=lambda n: worst[n][0])
    print(f"[backward] 10deg: {len(worst)} parameter gradients, worst max-rel {worst[k][0]:.2e} l2 {worst[k][1]:.2e} ({k}); "
          f"oracle fp32-vs-fp64 autograd max-rel {noise:.2e}")
    # one AdamW step through the HIP optimiser kernel keeps the model finite and changes every parameter
    opt = gw.AdamW(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    opt.step()
    opt.zero_grad()
    for p, b in zip(model.parameters(), before):
        assert torch.isfinite(p).all() and not torch.equal(p, b)


@pytest.mark.parametrize("irregular,checkpoint", [(False, False), (True, False), (False, True)])
def test_forecaster_and_loss_like_the_reference_tests(irregular, checkpoint):
    """tests/test_model.py:157-172 (regular 5 degree grid), :174-191 (jittered lat/lons: every latitude unique) and
    :220-234 (use_checkpointing=True): forward, NormalizedMSELoss, loss.backward() - here also checked against the oracle."""
    rs = np.random.RandomState(11)
    lat_lons = [(float(lat), float(lon)) for lat in range(-90, 90, 5) for lon in range(0, 360, 5)]
    if irregular:
        lat_lons = [(lat + rs.random_sample(), lon + rs.random_sample()) for lat, lon in lat_lons]
    var = torch.from_numpy(rs.standard_normal(78).astype(np.float32))
    model = gw.GraphWeatherForecaster(lat_lons, use_checkpointing=checkpoint)
    deterministic_fill_(model, seed=2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    feats = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78 + 24)).astype(np.float32))
    target = torch.from_numpy(rs.random_sample((2, len(lat_lons), 78)).astype(np.float32))
    y_ref = om.forecaster_forward(sd, model.encoder.graphs.as_oracle_dict(), feats)
    loss_ref = om.normalized_mse_loss(y_ref, target, lat_lons, var)
    criterion = gw.NormalizedMSELoss(lat_lons=lat_lons, feature_variance=var)
    model = model.to(DEV)
    out = model(feats.to(DEV))
    loss = criterion(out, target.to(DEV))
    assert not torch.isnan(loss) and not torch.isnan(out).any()
    assert _rel(out.cpu() - feats[..., :78], y_ref - feats[..., :78]) < 2e-4
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_example_script_runs():
    """examples/switch_from_reference.py: checkpoint round trip, two training steps with a decreasing loss, inference, rollout."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "switch_from_reference.py"), "--grid", "10", "--train-steps", "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    losses = [float(l.split("loss")[1]) for l in out.stdout.splitlines() if l.startswith("train step")]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    assert "finite: True" in out.stdout and "rollout steps: 3" in out.stdout
