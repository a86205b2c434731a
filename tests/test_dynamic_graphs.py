"""``DynamicGraphBuilder`` on CPU: reads like the reference's ``tests/test_dynamic_graph_builder.py``,
plus a comparison with the reference's own builder when ``/root/reference`` is present (this container only)."""
import numpy as np
import pytest
import torch

from graph_weather_amd.mesh import num_cells
from graph_weather_amd.regional import BoundaryNudgingLayer, DynamicGraphBuilder, RegionalForecasterConfig
from graph_weather_amd.utils import validate_lat_lons
from oracle.refload import reference_available


def _small_region():
    return [(float(lat), float(lon)) for lat in range(50, 55) for lon in range(-2, 3)]


def test_encoder_graph():
    builder = DynamicGraphBuilder(resolution=2)
    graph, h3_indices = builder.build_encoder_graph(_small_region())
    assert graph.edge_index.shape == (2, 25) and graph.edge_index.dtype == torch.long
    assert graph.edge_index[0].tolist() == list(range(25))
    assert graph.edge_index[1].min().item() >= 25
    assert graph.edge_attr.shape == (25, 2) and graph.edge_attr.abs().max() <= 1.0
    assert all(0 <= idx < num_cells(2) for idx in h3_indices)
    assert graph.edge_index[1].max().item() == 25 + len(h3_indices) - 1  # targets index the sorted unique cells


def test_decoder_graph():
    builder = DynamicGraphBuilder(resolution=2)
    lat_lons = _small_region()
    graph = builder.build_decoder_graph(lat_lons)
    # every coordinate receives from the disk-1 of its cell: 7 edges (6 in a pentagon cell) - 175 for this patch as in the
    # reference's test (tests/test_dynamic_graph_builder.py:43)
    assert graph.edge_index.shape == (2, 175) and graph.edge_index.dtype == torch.long
    assert graph.edge_attr.shape == (175, 2) and graph.edge_attr.abs().max() <= 1.0
    n_hood = int(graph.edge_index[0].max()) + 1
    assert graph.edge_index[1].min().item() == n_hood and graph.edge_index[1].max().item() == n_hood + 24


def test_latent_graph():
    builder = DynamicGraphBuilder(resolution=2)
    _, unique_cells, _ = builder._assign_h3_cells(_small_region())
    graph = builder.build_latent_graph(unique_cells)
    n = len(unique_cells)
    src, dst = graph.edge_index
    assert (src == dst).sum().item() == n  # one self loop per cell
    assert graph.edge_attr.shape == (graph.edge_index.shape[1], 2)
    pairs = set(zip(src.tolist(), dst.tolist()))
    assert all((b, a) in pairs for a, b in pairs)  # neighbourhood is symmetric
    assert int(src.max()) < n and int(dst.max()) < n
    # global ranks are accepted as cell ids too
    ranks = [builder.global_h3_map[c] for c in unique_cells]
    assert torch.equal(builder.build_latent_graph(ranks).edge_index, graph.edge_index)


def test_builder_caching():
    builder = DynamicGraphBuilder(resolution=2)
    lat_lons = _small_region()
    res1, res2 = builder(lat_lons), builder(lat_lons)
    assert all(res1[i] is res2[i] for i in range(4))
    res3 = builder([(0.0, 0.0), (1.0, 1.0)])
    assert res1[0] is not res3[0]


def test_validation_ranges():
    builder = DynamicGraphBuilder(resolution=2)
    with pytest.raises(ValueError, match="must not be empty"):
        builder([])
    with pytest.raises(ValueError, match="latitude"):
        builder([(91.0, 0.0)])
    with pytest.raises(ValueError, match="latitude"):
        builder([(-91.0, 0.0)])
    res = builder([(-90.0, 0.0), (90.0, 180.0)])
    assert res[0].edge_index.shape == (2, 2)
    with pytest.raises(ValueError, match="must not be empty"):
        validate_lat_lons([])
    validate_lat_lons([(0.0, 0.0), (45.0, 90.0)])


def test_native_plans_are_the_same_edges_destination_sorted():
    builder = DynamicGraphBuilder(resolution=2)
    rs = np.random.RandomState(3)
    lat_lons = [(float(a), float(b)) for a, b in zip(rs.uniform(30, 60, 400), rs.uniform(-20, 30, 400))]
    enc, _, lat, h3_idx = builder(lat_lons)
    enc_plan, lat_plan, dec_plan, rows = builder.native_plans(lat_lons, torch.device("cpu"))
    n, c = 400, len(h3_idx)
    assert rows.tolist() == h3_idx and (enc_plan.n_src, enc_plan.n_dst) == (n, c) and (dec_plan.n_src, dec_plan.n_dst) == (c, n)
    for plan, ei, off_s, off_d in ((enc_plan, enc.edge_index, 0, n), (lat_plan, lat.edge_index, 0, 0), (dec_plan, enc.edge_index.flip(0), n, 0)):
        assert torch.all(plan.dst[1:] >= plan.dst[:-1])
        assert torch.equal(plan.src.long(), ei[0][plan.perm] - off_s) and torch.equal(plan.dst.long(), ei[1][plan.perm] - off_d)
    assert torch.equal(dec_plan.edge_attr, enc.edge_attr)  # reversed encoder edges keep their attributes (regional_forecast.py:262)


def test_relaxation_weights_range():
    """tests/test_regional_forecast.py:175-183."""
    uk = [(51.5, -0.1), (52.0, 0.5), (53.0, -1.0), (54.0, -2.0), (50.0, -3.0)]
    w = BoundaryNudgingLayer._compute_relaxation_weights(uk, torch.device("cpu"))
    assert w.shape == (5, 1) and w.min() >= 0.0 and w.max() <= 1.0 and torch.isclose(w.max(), torch.tensor(1.0))


def test_config_build_and_state_dict_keys():
    """tests/test_regional_forecast.py:35-42 + the reference's key set (regional_forecast.py:135-232)."""
    model = RegionalForecasterConfig(enable_nudging=True).build()
    assert hasattr(model, "forward") and hasattr(model, "graph_builder") and hasattr(model, "h3_embeddings")
    keys = set(model.state_dict().keys())
    for k in ("h3_embeddings", "node_encoder.model.0.weight", "edge_encoder.model.5.bias", "latent_edge_encoder.model.4.weight",
              "encoder_gnn.blocks.0.edge_model.edge_mlp.model.0.weight", "processor.graph_processor.blocks.8.node_model.node_mlp.model.5.weight",
              "decoder_edge_encoder.model.2.bias", "decoder_gnn.blocks.0.node_model.node_mlp.model.4.bias",
              "node_decoder.model.5.weight", "nudging.blend_mlp.model.0.weight", "nudging.blend_mlp.model.2.bias"):
        assert k in keys, k
    assert model.state_dict()["node_decoder.model.5.weight"].shape == (78,)  # LayerNorm(output_dim) on the head
    assert model.state_dict()["nudging.blend_mlp.model.0.weight"].shape == (64, 157)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(1, 2, 102), [(0.0, 0.0), (1.0, 1.0)])


@pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")
def test_vectorised_builder_equals_the_reference_builder():
    """The reference's own DynamicGraphBuilder (Python loops over the h3 call surface) and the vectorised one give
    identical arrays; the reference model has the same state_dict keys and shapes."""
    from oracle.refload import load_reference

    ns = load_reference()
    rs = np.random.RandomState(5)
    regions = [_small_region(),
               [(float(a), float(b)) for a, b in zip(rs.uniform(-90, 90, 300), rs.uniform(-180, 180, 300))],
               [(-90.0, 0.0), (90.0, 180.0), (0.0, 0.0)]]
    ref, mine = ns.DynamicGraphBuilder(resolution=2), DynamicGraphBuilder(resolution=2)
    for lat_lons in regions:
        r, m = ref(lat_lons), mine(lat_lons)
        for a, b in zip(r[:3], m[:3]):
            assert torch.equal(a.edge_index, b.edge_index)
            assert torch.equal(a.edge_attr, b.edge_attr)
        assert list(r[3]) == list(m[3])
    ref_model = ns.RegionalForecasterConfig(enable_nudging=True).build()
    my_model = RegionalForecasterConfig(enable_nudging=True).build()
    assert {k: tuple(v.shape) for k, v in ref_model.state_dict().items()} == {k: tuple(v.shape) for k, v in my_model.state_dict().items()}
