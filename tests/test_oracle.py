"""The CPU oracle against (a) golden vectors produced by the reference's own source files,
(b) the reference executed live when /root/reference exists, (c) the invariants the reference's
tests state."""
import os

import numpy as np
import pytest
import torch

from graph_weather_amd.graphs import build_forecast_graphs
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features
from oracle import reference_math as om
from oracle.refload import reference_available

from .helpers import forecaster_param_shapes, make_params


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag,i,o,h,norm", [("node_enc", 102, 256, 256, True), ("edge_enc", 2, 256, 256, True),
                                            ("node_dec", 256, 78, 128, False)])
def test_mlp_matches_reference_golden(golden_dir, tag, i, o, h, norm):
    g = _load(golden_dir, f"mlp_{tag}.npz")
    shapes = {"m.model.0.weight": (h, i), "m.model.0.bias": (h,), "m.model.2.weight": (h, h), "m.model.2.bias": (h,),
              "m.model.4.weight": (o, h), "m.model.4.bias": (o,)}
    if norm:
        shapes.update({"m.model.5.weight": (o,), "m.model.5.bias": (o,)})
    # the golden generator filled a bare MLP, so its keys have no prefix: model.0.weight ...
    p = {"m." + k: v for k, v in make_params({k[2:]: s for k, s in shapes.items()}, seed=11).items()}
    y = om.mlp(p, "m", torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-5, atol=1e-5)


def test_graph_processor_random_coo_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "graph_processor_random.npz")
    shapes = {}
    for b in range(2):
        for kind, kin in (("edge_model.edge_mlp", 768), ("node_model.node_mlp", 512)):
            pre = f"blocks.{b}.{kind}.model"
            shapes.update({f"{pre}.0.weight": (256, kin), f"{pre}.0.bias": (256,), f"{pre}.2.weight": (256, 256),
                           f"{pre}.2.bias": (256,), f"{pre}.4.weight": (256, 256), f"{pre}.4.bias": (256,),
                           f"{pre}.5.weight": (256,), f"{pre}.5.bias": (256,)})
    p = {"gp." + k: v for k, v in make_params(shapes, seed=3).items()}
    rs = np.random.RandomState(123)
    x = torch.from_numpy(rs.standard_normal((500, 256)).astype(np.float32))
    ea = torch.from_numpy(rs.standard_normal((3000, 256)).astype(np.float32))
    ei = torch.from_numpy(rs.randint(0, 500, size=(2, 3000)).astype(np.int64))
    assert np.array_equal(ei.numpy(), g["edge_index"])
    xo, eo = om.graph_processor(p, "gp", x, ei, ea)
    np.testing.assert_allclose(xo.numpy(), g["x_out"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(eo[::5].numpy(), g["e_out_rows"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag,step,batch", [("10deg_b2", 10.0, 2), ("5deg_b1", 5.0, 1)])
def test_forecaster_matches_reference_golden(golden_dir, tag, step, batch):
    g = _load(golden_dir, f"forecaster_{tag}.npz")
    lat_lons = regular_lat_lons(step)
    graphs = build_forecast_graphs(lat_lons, 2)
    assert graphs.enc_edge_index.shape[1] == int(g["enc_num_edges"])
    assert graphs.lat_edge_index.shape[1] == int(g["lat_num_edges"]) == 41162
    assert graphs.dec_edge_index.shape[1] == int(g["dec_num_edges"])
    for name, ei in (("enc", graphs.enc_edge_index), ("lat", graphs.lat_edge_index), ("dec", graphs.dec_edge_index)):
        e = ei.numpy()
        w = np.arange(1, e.shape[1] + 1, dtype=np.int64)
        assert int(((e[0] * 31 + e[1] * 17) * w % 1000003).sum()) == int(g[name + "_index_checksum"])
    p = make_params(forecaster_param_shapes(graphs.num_mesh), seed=0)
    feats = seeded_features(batch, len(lat_lons), 102, seed=42)
    gd = graphs.as_oracle_dict()
    y, inter = om.forecaster_forward(p, gd, feats, return_intermediates=True)
    np.testing.assert_allclose(inter["enc_x"][::37].numpy(), g["enc_x_rows"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(inter["proc_x"][::37].numpy(), g["proc_x_rows"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=2e-4, atol=2e-4)
    # literal max(edge_index)+1 offsets == true node counts on these grids
    if batch > 1:
        y_lit = om.forecaster_forward(p, gd, feats, literal_offsets=True)
        assert torch.equal(y_lit, y)
    # shared-graph semantics == replicated semantics (tests/models/layers/test_efficient_batching.py:145)
    y_sh = om.forecaster_forward(p, gd, feats, shared=True)
    np.testing.assert_allclose(y_sh.numpy(), y.numpy(), rtol=1e-4, atol=1e-4)
    # loss
    rs = np.random.RandomState(7)
    target = torch.from_numpy(rs.random_sample(tuple(y.shape)).astype(np.float32))
    var = torch.from_numpy((rs.random_sample(78) + 0.5).astype(np.float32))
    y_gold = torch.from_numpy(g["out"])
    assert abs(om.normalized_mse_loss(y_gold, target, lat_lons).item() - float(g["loss"])) < 1e-6
    assert abs(om.normalized_mse_loss(y_gold, target, lat_lons, var, True).item() - float(g["loss_normalized"])) < 1e-6


def test_zero_parameters_give_residual_identity():
    """All parameters zero => out == features[..., :78] (pattern of tests/test_regional_forecast.py:113-125)."""
    lat_lons = regular_lat_lons(20.0)
    graphs = build_forecast_graphs(lat_lons, 2)
    p = {k: torch.zeros(s) for k, s in forecaster_param_shapes(graphs.num_mesh).items()}
    feats = seeded_features(2, len(lat_lons))
    y = om.forecaster_forward(p, graphs.as_oracle_dict(), feats)
    assert torch.equal(y, feats[..., :78])


def test_loss_closed_form():
    """tests/test_model.py:236-271: out**2/var == 1 everywhere => loss == mean of the cos-lat weight grid."""
    lat_lons = [(lat, lon) for lat in range(-90, 90, 5) for lon in range(0, 360, 5)]
    var = torch.rand(78) + 0.5
    pred = torch.sqrt(var)[None, None, :].expand(2, len(lat_lons), 78)
    target = torch.zeros_like(pred)
    loss = om.normalized_mse_loss(pred, target, lat_lons, var, normalize=True)
    w = np.cos(np.arange(-90, 90, 5) * np.pi / 180.0)
    assert abs(loss.item() - w.mean()) < 1e-4


@pytest.mark.reference
@pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")
def test_oracle_matches_live_reference_small():
    from oracle.refload import load_reference

    ns = load_reference()
    lat_lons = regular_lat_lons(30.0)
    model = ns.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=5)
    model.eval()
    feats = seeded_features(3, len(lat_lons), seed=9)
    with torch.no_grad():
        y_ref = model(feats)
    graphs = build_forecast_graphs(lat_lons, 2)
    assert torch.equal(model.encoder.graph.edge_index, graphs.enc_edge_index)
    assert torch.equal(model.encoder.latent_graph.edge_index, graphs.lat_edge_index)
    assert torch.equal(model.decoder.graph.edge_index, graphs.dec_edge_index)
    assert torch.allclose(model.decoder.graph.edge_attr, graphs.dec_edge_attr, atol=1e-7)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    assert set(p.keys()) == set(forecaster_param_shapes(graphs.num_mesh).keys())
    y = om.forecaster_forward(p, graphs.as_oracle_dict(), feats)
    assert torch.allclose(y, y_ref, rtol=1e-5, atol=1e-5)


def _assimilator_setup():
    """Shared by the oracle (CPU) and HIP (GPU) tests of GraphWeatherAssimilator: graphs, weights, inputs of the golden case."""
    from graph_weather_amd.graphs import build_forecast_graphs, build_latent_graph, build_observation_graph
    from oracle.gen_golden import assimilator_observations

    out_lat_lons = regular_lat_lons(10.0)
    llh, feats = assimilator_observations()
    obs_ei, obs_attr, _ = build_observation_graph(llh.numpy(), 2)
    lat_ei, lat_attr, _ = build_latent_graph(2)
    fg = build_forecast_graphs(out_lat_lons, 2)
    g = {"obs_edge_index": obs_ei, "obs_edge_attr": obs_attr, "lat_edge_index": lat_ei, "lat_edge_attr": lat_attr,
         "dec_edge_index": fg.dec_edge_index, "dec_edge_attr": fg.dec_edge_attr, "num_mesh": fg.num_mesh, "num_grid": fg.num_grid}
    return out_lat_lons, llh, feats, g


def test_assimilator_matches_reference_golden(golden_dir):
    """analysis.py:52-150 executed from the reference's own files (oracle/gen_golden.py: assimilator_case)."""
    import graph_weather_amd as gw

    gold = np.load(os.path.join(golden_dir, "assimilator_10deg.npz"))
    out_lat_lons, llh, feats, g = _assimilator_setup()
    assert np.array_equal(g["obs_edge_index"].numpy(), gold["obs_edge_index"])  # same graph as the reference built
    model = gw.GraphWeatherAssimilator(output_lat_lons=out_lat_lons, analysis_dim=24)  # module mirror: same state_dict keys
    deterministic_fill_(model, seed=6)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    y = om.assimilator_forward(p, g, feats, 24)
    assert y.shape == (1, 648, 24)
    err = (y - torch.from_numpy(gold["y"])).abs().max().item()
    assert err < 2e-5, err


def test_graphcast_wrapper_matches_reference_golden(golden_dir):
    """graphcast/model.py executed from the reference's own files (efficient_batching on and off agree there to 0.0):
    the oracle's forecaster composition with the input as residual reproduces it."""
    import graph_weather_amd as gw

    gold = np.load(os.path.join(golden_dir, "graphcast_10deg_b2.npz"))
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphCast(lat_lons, efficient_batching=True)  # module mirror: same state_dict keys as the reference class
    deterministic_fill_(model, seed=5)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    feats = seeded_features(2, len(lat_lons), 78, seed=9)
    y = om.forecaster_forward(p, model.encoder.graphs.as_oracle_dict(), feats, feature_dim=78)
    assert (y - torch.from_numpy(gold["y"])).abs().max().item() < 2e-5
    assert (y - torch.from_numpy(gold["y_efficient"])).abs().max().item() < 2e-5


def _regional_setup():
    """Shared by the oracle (CPU) and HIP (GPU) tests of RegionalForecaster: coordinates, inputs, graphs of the golden case."""
    from graph_weather_amd.regional import DynamicGraphBuilder
    from oracle.gen_golden import regional_inputs

    lat_lons, feats, ctx = regional_inputs()
    enc, dec, lat, h3_idx = DynamicGraphBuilder(2)(lat_lons)
    g = {"enc_edge_index": enc.edge_index, "enc_edge_attr": enc.edge_attr, "lat_edge_index": lat.edge_index,
         "lat_edge_attr": lat.edge_attr, "dec_edge_index": dec.edge_index, "h3_indices": h3_idx}
    return lat_lons, feats, ctx, g


def test_regional_forecaster_matches_reference_golden(golden_dir):
    """regional_forecast.py:135-298 and dynamic_graph_builder.py executed from the reference's own files
    (oracle/gen_golden.py: regional_case): same graphs from the vectorised builder, same outputs from the oracle, with and
    without boundary nudging."""
    import graph_weather_amd as gw

    gold = np.load(os.path.join(golden_dir, "regional_eu_b2.npz"))
    lat_lons, feats, ctx, g = _regional_setup()
    for name in ("enc_edge_index", "lat_edge_index", "dec_edge_index"):
        assert np.array_equal(g[name].numpy(), gold[name]), name
    assert list(gold["h3_indices"]) == g["h3_indices"]
    model = gw.RegionalForecasterConfig(enable_nudging=True).build()  # module mirror: same state_dict keys
    deterministic_fill_(model, seed=8)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    y = om.regional_forward(p, g, feats, 78)
    assert y.shape == (2, 768, 78)
    assert (y - torch.from_numpy(gold["y"])).abs().max().item() < 2e-5
    yn = om.regional_forward(p, g, feats, 78, global_context=ctx, lat_lons=lat_lons)
    assert (yn - torch.from_numpy(gold["y_nudged"])).abs().max().item() < 2e-5
    assert (yn - y).abs().mean().item() > 0.1  # the nudging layer does something


def test_chunked_oracle_equals_whole_tensor_oracle():
    """oracle/chunked.py (slab-wise, decoder on a row sample) is the same arithmetic as the whole-tensor oracle: all rows at
    10 degree with slabs far smaller than the graphs, and a row sample."""
    from oracle import chunked as oc

    lat_lons = regular_lat_lons(10.0)
    graphs = build_forecast_graphs(lat_lons, 2)
    p = make_params(forecaster_param_shapes(graphs.num_mesh), seed=3)
    feats = seeded_features(2, len(lat_lons), seed=4)
    g = graphs.as_oracle_dict()
    with torch.no_grad():
        y = om.forecaster_forward(p, g, feats)
    rows_all = torch.arange(len(lat_lons))
    y_all = oc.forecaster_rows(p, g, feats, rows_all, slab=1000)
    scale = (y - feats[..., :78]).abs().max().item()
    assert (y_all - y).abs().max().item() <= 2e-6 * scale
    rows = torch.from_numpy(np.random.RandomState(0).choice(len(lat_lons), size=50, replace=False)).long()
    y_s = oc.forecaster_rows(p, g, feats[1:], rows, slab=4096)
    assert (y_s[0] - y[1][rows]).abs().max().item() <= 2e-6 * scale
