"""RegionalForecaster / BoundaryNudgingLayer / LayerNorm on an output head through the C ABI,
against the golden vector produced by the reference's own files, the CPU oracle and its fp64 autograd.  The behavioural
tests follow the reference's ``tests/test_regional_forecast.py`` (at the default widths the HIP kernels are built for)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd import _lib  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_  # noqa: E402
from oracle import reference_math as om  # noqa: E402

from .test_gpu_backward import _check_param_grads, _rel  # noqa: E402
from .test_gpu_parity import GOLDEN_DIR, _close  # noqa: E402

DEV = "cuda:0"


def _uk_latlons():
    return [(51.5, -0.1), (52.0, 0.5), (53.0, -1.0), (54.0, -2.0), (50.0, -3.0)]


def _germany_latlons():
    return [(52.5, 13.4), (48.1, 11.6), (50.9, 6.9)]


@pytest.fixture(scope="module")
def small_model():
    """Default widths (the kernels' widths), two processor blocks to keep the module cheap."""
    model = gw.RegionalForecasterConfig(num_blocks=2).build()
    deterministic_fill_(model, seed=3)
    return model.to(DEV)


@pytest.mark.parametrize("n_out,hidden,rows", [(78, 128, 515), (24, 256, 130), (80, 128, 64), (1, 128, 33)])
def test_mlp_head_with_layernorm_forward_and_backward(n_out, hidden, rows):
    """node_decoder of the regional model: MLP(256 -> hidden -> hidden -> n_out) + LayerNorm(n_out) (regional_forecast.py:223-230)."""
    m = gw.MLP(256, n_out, hidden, 2, "LayerNorm")
    deterministic_fill_(m, seed=n_out)
    rs = np.random.RandomState(rows)
    x = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((rows, n_out)).astype(np.float32))
    ref = {"m." + k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    y_ref = om.mlp(ref, "m", xr)
    y_ref.backward(dy.double())
    m = m.to(DEV)
    with torch.no_grad():
        _close(m(x.to(DEV)), y_ref, what=f"head 256->{n_out} + LayerNorm (inference kernel)")
    if n_out == 1:
        return  # LayerNorm(1) is the constant beta: nothing to differentiate
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    _close(y, y_ref, what=f"head 256->{n_out} + LayerNorm (training forward)")
    y.backward(dy.to(DEV))
    worst = {}
    _check_param_grads(m, ref, "m.", worst)
    assert _rel(xd.grad, xr.grad) < 2e-3


@pytest.mark.parametrize("width,ld", [(78, 78), (78, 80), (200, 203), (1, 4)])
def test_layernorm_backward_narrow_widths(width, ld):
    rs = np.random.RandomState(width + ld)
    rows = 1031
    y = torch.from_numpy(rs.standard_normal((rows, ld)).astype(np.float32))
    dn = torch.from_numpy(rs.standard_normal((rows, width)).astype(np.float32))
    gamma = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(width)).astype(np.float32))
    yr = y[:, :width].double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = torch.zeros(width, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(yr, (width,), gr, br, 1e-5).backward(dn.double())
    yd, dnd, gd = y.to(DEV), dn.to(DEV), gamma.to(DEV)
    dy = torch.empty((rows, width), device=DEV)
    dg, db = torch.zeros(width, device=DEV), torch.zeros(width, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().gw_layernorm_backward(rows, width, dnd.data_ptr(), width, yd.data_ptr(), ld, gd.data_ptr(), dy.data_ptr(),
                                                width, dg.data_ptr(), db.data_ptr(), st), "gw_layernorm_backward")
    if width == 1:  # LayerNorm(1) is the constant beta: the input and gain gradients are exactly zero
        assert dy.abs().max().item() == 0.0 and dg.abs().max().item() < 1e-6 and _rel(db, br.grad) < 1e-4
        return
    assert _rel(dy, yr.grad) < 1e-4 and _rel(dg, gr.grad) < 1e-4 and _rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("feat,hidden,n", [(78, 64, 301), (12, 16, 5), (130, 200, 77)])
def test_boundary_nudging_layer_forward_and_backward(feat, hidden, n):
    """regional_forecast.py:44-132 as one forward and one backward kernel, against the oracle's autograd in fp64."""
    rs = np.random.RandomState(feat)
    lat_lons = [(float(a), float(b)) for a, b in zip(rs.uniform(40, 60, n), rs.uniform(-10, 20, n))]
    layer = gw.BoundaryNudgingLayer(feat, hidden)
    deterministic_fill_(layer, seed=4)
    regional = torch.from_numpy(rs.standard_normal((2, n, feat)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((2, n, feat)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((2, n, feat)).astype(np.float32))

    def oracle(params, reg, dtype):
        prior = om.nudging_weights(lat_lons).to(dtype).unsqueeze(0).expand(2, -1, -1)
        corr = om.mlp(params, "blend_mlp", torch.cat([reg, ctx.to(dtype), prior], dim=-1))
        alpha = torch.clamp(prior + corr, 0.0, 1.0)
        return (1 - alpha) * reg + alpha * ctx.to(dtype), alpha

    ref = {k: v.detach().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    rr = regional.double().requires_grad_(True)
    y_ref, alpha = oracle(ref, rr, torch.float64)
    frac_open = ((alpha > 0) & (alpha < 1)).double().mean().item()
    assert 0.05 < frac_open < 1.0, frac_open  # the case exercises both the clamped and the open regime
    y_ref.backward(dy.double())
    layer = layer.to(DEV)
    rd = regional.to(DEV).requires_grad_(True)
    y = layer(rd, ctx.to(DEV), lat_lons)
    _close(y, y_ref, rel=1e-5, what="nudging forward")
    y.backward(dy.to(DEV))
    worst = {}
    _check_param_grads(layer, ref, "", worst)
    assert _rel(rd.grad, rr.grad) < 2e-3
    with torch.no_grad():
        _close(layer(regional.to(DEV), ctx.to(DEV), lat_lons), y_ref, rel=1e-5, what="nudging forward (no_grad)")


def test_regional_forecaster_matches_reference_golden_and_oracle():
    """The golden vector was produced by the reference's own RegionalForecaster / DynamicGraphBuilder source files."""
    from .test_oracle import _regional_setup

    gold = np.load(GOLDEN_DIR + "/regional_eu_b2.npz")
    lat_lons, feats, ctx, g = _regional_setup()
    model = gw.RegionalForecasterConfig(enable_nudging=True).build()
    deterministic_fill_(model, seed=8)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    with torch.no_grad():
        y = model(feats.to(DEV), lat_lons)
        y_nudged = model(feats.to(DEV), lat_lons, global_context=ctx.to(DEV))
        y_again = model(feats.to(DEV), lat_lons)
    assert y.shape == (2, 768, 78) and not torch.isnan(y).any()
    # compare the predicted increment (the residual input would otherwise dominate the scale)
    res = feats[..., :78]
    _close(y.cpu() - res, torch.from_numpy(gold["y"]) - res, what="regional vs reference golden")
    _close(y.cpu() - res, om.regional_forward(p, g, feats, 78) - res, what="regional vs oracle")
    _close(y_nudged, torch.from_numpy(gold["y_nudged"]), what="regional + nudging vs reference golden")
    _close(y_again, y, rel=1e-6, what="cached graphs and embeddings")
    # a different batch size and a different region through the same module
    with torch.no_grad():
        y1 = model(feats[:1].to(DEV), lat_lons)
    _close(y1, y[:1], rel=1e-6, what="batch of one")


def test_different_regions_variable_lengths_and_residual(small_model):
    """tests/test_regional_forecast.py:45-88: shapes, no NaN, successive regions of different sizes."""
    model = small_model.eval()
    with torch.no_grad():
        out = model(torch.randn(2, 5, 102, device=DEV), _uk_latlons())
        assert out.shape == (2, 5, 78) and not torch.isnan(out).any()
        out_de = model(torch.randn(1, 3, 102, device=DEV), _germany_latlons())
        assert out_de.shape == (1, 3, 78) and not torch.isnan(out_de).any()
        # nudging disabled: global_context is ignored (:141-151)
        f = torch.randn(1, 5, 102, device=DEV)
        assert torch.equal(model(f, _uk_latlons(), global_context=torch.randn(1, 5, 78, device=DEV)), model(f, _uk_latlons()))
    with pytest.raises(RuntimeError, match="number of coordinates"):
        model(torch.randn(1, 4, 102, device=DEV), _uk_latlons())


def test_zero_parameters_give_the_residual_and_output_dim_override():
    """tests/test_regional_forecast.py:101-125."""
    model = gw.RegionalForecasterConfig(num_blocks=1, output_dim=6).build().to(DEV)
    features = torch.randn(1, 5, 102, device=DEV)
    with torch.no_grad():
        assert model(features, _uk_latlons()).shape == (1, 5, 6)
        for p in model.parameters():
            p.zero_()
        out = model(features, _uk_latlons())
    assert torch.allclose(out, features[..., :6], atol=1e-5)


def test_regional_backward_matches_oracle_autograd():
    """tests/test_regional_forecast.py:91-99,186-198 (gradients reach h3_embeddings, the encoders and the nudging layer) -
    here every parameter gradient is checked against the oracle in fp64."""
    rs = np.random.RandomState(17)
    n = 230
    lat_lons = [(float(a), float(b)) for a, b in zip(rs.uniform(35, 65, n), rs.uniform(-15, 30, n))]
    model = gw.RegionalForecasterConfig(num_blocks=2, enable_nudging=True).build()
    deterministic_fill_(model, seed=12)
    feats = torch.from_numpy(rs.standard_normal((2, n, 102)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((2, n, 78)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((2, n, 78)).astype(np.float32))
    enc, _, lat, h3_idx = model.graph_builder(lat_lons)
    g = {"enc_edge_index": enc.edge_index, "enc_edge_attr": enc.edge_attr, "lat_edge_index": lat.edge_index,
         "lat_edge_attr": lat.edge_attr, "h3_indices": h3_idx}
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    y_ref = om.regional_forward(ref, om.graphs_to_dtype(g, torch.float64), feats.double(), 78, global_context=ctx.double(), lat_lons=lat_lons)
    (y_ref * dy.double()).sum().backward()
    ref32 = {k: v.detach().float().requires_grad_(True) for k, v in ref.items()}
    (om.regional_forward(ref32, g, feats, 78, global_context=ctx, lat_lons=lat_lons) * dy).sum().backward()
    noise = max(_rel(ref32[k].grad, ref[k].grad) for k in ref)
    model = model.to(DEV).train()
    y = model(feats.to(DEV), lat_lons, global_context=ctx.to(DEV))
    _close(y, y_ref, what="regional training forward")
    (y * dy.to(DEV)).sum().backward()
    assert model.h3_embeddings.grad is not None and model.h3_embeddings.grad.abs().sum() > 0
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=max(2e-3, 4 * noise))
    k = max(worst, key=lambda name: worst[name][0])
    print(f"[backward] regional: {len(worst)} parameter gradients, worst max-rel {worst[k][0]:.2e} ({k}); oracle fp32-vs-fp64 {noise:.2e}")
    # nudging on but no context: same as the plain model (:154-162), and still differentiable
    out = model(feats.to(DEV), lat_lons, global_context=None)
    assert out.shape == (2, n, 78) and not torch.isnan(out).any()
    with torch.no_grad():
        assert not torch.allclose(out, model(feats.to(DEV), lat_lons, global_context=10.0 * ctx.to(DEV)))
