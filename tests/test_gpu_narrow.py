"""Widths other than the kernels' native 256 (the reference's own constructor defaults are 128, and its regional tests
build 32-wide models): the modules run zero-padded on the same kernels (``layers.MLP`` docstring).  Forward and backward
against the oracle; the RegionalForecaster cases are the reference's ``tests/test_regional_forecast.py`` at its own
``_small_config``."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons  # noqa: E402
from oracle import reference_math as om  # noqa: E402

from .test_gpu_backward import _check_param_grads, _rel  # noqa: E402
from .test_gpu_parity import _close  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("i,o,h,layers,norm", [(16, 128, 128, 2, "LayerNorm"), (102, 32, 32, 2, "LayerNorm"), (200, 64, 96, 3, "LayerNorm"),
                                                (32, 12, 32, 2, "LayerNorm"), (157, 1, 64, 1, None), (64, 100, 40, 1, "LayerNorm"),
                                                (256, 78, 32, 1, None)])
def test_mlp_any_width_forward_and_backward(i, o, h, layers, norm):
    """``MLP(in_dim)`` with the reference's defaults (128, graph_net_block.py:20-28) and other widths / depths."""
    m = gw.MLP(i, o, h, layers, norm)
    deterministic_fill_(m, seed=i + o)
    rs = np.random.RandomState(h)
    x = torch.from_numpy(rs.standard_normal((333, i)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((333, o)).astype(np.float32))
    ref = {"m." + k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    y_ref = om.mlp(ref, "m", xr)
    y_ref.backward(dy.double())
    m = m.to(DEV)
    with torch.no_grad():
        y = m(x.to(DEV))
    assert y.shape == (333, o)
    _close(y, y_ref, what=f"MLP {i}->{h}x{layers}->{o} (inference)")
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    _close(y, y_ref, what=f"MLP {i}->{h}x{layers}->{o} (training forward)")
    y.backward(dy.to(DEV))
    worst = {}
    if not (o == 1 and norm):
        _check_param_grads(m, ref, "m.", worst)
        assert _rel(xd.grad, xr.grad) < 2e-3


def test_graph_processor_reference_defaults_random_coo():
    """``GraphProcessor()`` as the reference constructs it by default: 128-wide nodes and edges (graph_net_block.py:234-244)."""
    gp = gw.GraphProcessor(mp_iterations=3)
    deterministic_fill_(gp, seed=4)
    ref = {"gp." + k: v.detach().double().requires_grad_(True) for k, v in gp.state_dict().items()}
    rs = np.random.RandomState(5)
    n, e = 150, 900
    x = torch.from_numpy(rs.standard_normal((n, 128)).astype(np.float32))
    ea = torch.from_numpy(rs.standard_normal((e, 128)).astype(np.float32))
    ei = torch.from_numpy(np.stack([rs.randint(0, n, size=e), np.where(rs.rand(e) < 0.2, 3, rs.randint(0, n, size=e))]).astype(np.int64))
    gx = torch.from_numpy(rs.standard_normal((n, 128)).astype(np.float32))
    ge = torch.from_numpy(rs.standard_normal((e, 128)).astype(np.float32))
    xr, er = x.double().requires_grad_(True), ea.double().requires_grad_(True)
    xo_r, eo_r = om.graph_processor(ref, "gp", xr, ei, er)
    ((xo_r * gx.double()).sum() + (eo_r * ge.double()).sum()).backward()
    gp = gp.to(DEV)
    with torch.no_grad():
        xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
    assert xo.shape == (n, 128) and eo.shape == (e, 128)
    _close(xo, xo_r, what="GraphProcessor(128) nodes")
    _close(eo, eo_r, what="GraphProcessor(128) edges")
    xd, ed = x.to(DEV).requires_grad_(True), ea.to(DEV).requires_grad_(True)
    xo, eo = gp(xd, ei.to(DEV), ed)
    ((xo * gx.to(DEV)).sum() + (eo * ge.to(DEV)).sum()).backward()
    worst = {}
    _check_param_grads(gp, ref, "gp.", worst)
    assert _rel(xd.grad, xr.grad) < 2e-3 and _rel(ed.grad, er.grad) < 2e-3


def test_narrow_forecaster_matches_oracle_forward_and_backward():
    """A forecaster with node 64 / edge 48 / hidden 96 and 40 / decoder hidden 32, single hidden layer in the edge MLPs."""
    lat_lons = regular_lat_lons(15.0)
    kw = dict(feature_dim=20, aux_dim=5, node_dim=64, edge_dim=48, num_blocks=2, hidden_dim_processor_node=96,
              hidden_dim_processor_edge=40, hidden_layers_processor_node=2, hidden_layers_processor_edge=1, hidden_dim_decoder=32,
              hidden_layers_decoder=2)
    model = gw.GraphWeatherForecaster(lat_lons, **kw)
    deterministic_fill_(model, seed=9)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    rs = np.random.RandomState(1)
    feats = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 25)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 20)).astype(np.float32))
    y_ref = om.forecaster_forward(ref, g64, feats.double(), feature_dim=20)
    (y_ref * dy.double()).sum().backward()
    model = model.to(DEV)
    with torch.no_grad():
        y = model(feats.to(DEV))
    res = feats[..., :20]
    _close(y.cpu() - res, y_ref.detach().float() - res, what="narrow forecaster (inference)")
    model.train()
    y = model(feats.to(DEV))
    _close(y.cpu() - res, y_ref.detach().float() - res, what="narrow forecaster (training forward)")
    (y * dy.to(DEV)).sum().backward()
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)
    # compositional API: tensors cross the module boundaries at their real widths (tests/test_model.py:106-119)
    with torch.no_grad():
        x, ei, ea = model.encoder(feats.to(DEV))
        assert x.shape[1] == 64 and ea.shape[1] == 48
        out = model.decoder(model.processor(x, ei, ea), feats.to(DEV)[..., :20])
    _close(out, y, rel=1e-5, what="narrow compositional")


# ---- the reference's tests/test_regional_forecast.py at its own small configuration ---------------------------------
def _small_config(**kw):
    return gw.RegionalForecasterConfig(feature_dim=12, aux_dim=4, node_dim=32, edge_dim=32, num_blocks=2, hidden_dim_processor_node=32,
                                       hidden_dim_processor_edge=32, hidden_dim_decoder=32, **kw)


def _uk_latlons():
    return [(51.5, -0.1), (52.0, 0.5), (53.0, -1.0), (54.0, -2.0), (50.0, -3.0)]


def _germany_latlons():
    return [(52.5, 13.4), (48.1, 11.6), (50.9, 6.9)]


def test_regional_small_config_like_the_reference_tests():
    model = _small_config().build().to(DEV)
    assert hasattr(model, "forward") and hasattr(model, "graph_builder") and hasattr(model, "h3_embeddings")
    out = model(torch.randn(2, 5, 16, device=DEV), _uk_latlons())  # test_forward_shape / test_no_nan_output
    assert out.shape == (2, 5, 12) and not torch.isnan(out).any()
    assert model(torch.randn(1, 3, 16, device=DEV), _germany_latlons()).shape == (1, 3, 12)  # test_different_coords_per_forward
    features = torch.randn(1, 5, 16, device=DEV)  # test_backward_pass
    model(features, _uk_latlons()).sum().backward()
    assert model.h3_embeddings.grad is not None
    assert any(p.grad is not None for p in model.node_encoder.parameters())
    cfg = _small_config()  # test_output_dim_override
    cfg.output_dim = 6
    assert cfg.build().to(DEV)(features, _uk_latlons()).shape == (1, 5, 6)
    with torch.no_grad():  # test_residual_connection
        for p in model.parameters():
            p.zero_()
    assert torch.allclose(model(features, _uk_latlons()), features[..., :12], atol=1e-5)


def test_regional_small_config_nudging_like_the_reference_tests_and_oracle():
    model = _small_config(enable_nudging=True, nudging_hidden_dim=16).build()
    deterministic_fill_(model, seed=2)
    lat_lons = _uk_latlons()
    rs = np.random.RandomState(4)
    feats = torch.from_numpy(rs.standard_normal((2, 5, 16)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((2, 5, 12)).astype(np.float32))
    enc, _, lat, h3_idx = model.graph_builder(lat_lons)
    g = {"enc_edge_index": enc.edge_index, "enc_edge_attr": enc.edge_attr, "lat_edge_index": lat.edge_index,
         "lat_edge_attr": lat.edge_attr, "h3_indices": h3_idx}
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    y_ref = om.regional_forward(ref, om.graphs_to_dtype(g, torch.float64), feats.double(), 12, global_context=ctx.double(), lat_lons=lat_lons)
    y_ref.sum().backward()
    model = model.to(DEV)
    out_no_ctx = model(feats.to(DEV), lat_lons, global_context=None)  # test_nudging_no_context_unchanged
    assert out_no_ctx.shape == (2, 5, 12) and not torch.isnan(out_no_ctx).any()
    y = model(feats.to(DEV), lat_lons, global_context=ctx.to(DEV))
    _close(y, y_ref, what="small regional model + nudging vs oracle")
    assert not torch.allclose(out_no_ctx, model(feats.to(DEV), lat_lons, global_context=10.0 * ctx.to(DEV)))  # test_nudging_changes_output
    y.sum().backward()  # test_nudging_backward_pass
    assert any(p.grad is not None for p in model.nudging.parameters())
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)
