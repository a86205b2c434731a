"""Host logic: init-time graph construction and the destination-sorted plans the kernels consume (CPU only)."""
import numpy as np
import torch

from graph_weather_amd.graphs import build_forecast_graphs, plan_from_coo
from graph_weather_amd.mesh import num_cells
from graph_weather_amd.utils import regular_lat_lons


def test_counts_the_reference_tests_pin():
    """tests/test_model.py:30-31,46-47: 5 882 mesh nodes and 41 162 directed latent edges at resolution 2."""
    g = build_forecast_graphs(regular_lat_lons(10.0), 2)
    assert g.num_mesh == 5882 == num_cells(2)
    assert g.lat_edge_index.shape == (2, 41162) and 41162 == 7 * 5882 - 12
    assert g.enc_edge_index.shape == (2, g.num_grid)  # one edge per grid node (encoder.py:86-104)
    # every grid node receives from the disk-1 of its cell: 7 edges, 6 for the 12 pentagon cells (assimilator_decoder.py:92-101)
    deg = np.bincount(g.dec_edge_index[1].numpy() - g.num_mesh, minlength=g.num_grid)
    assert set(np.unique(deg)) <= {6, 7} and deg.sum() == g.dec_edge_index.shape[1]
    # self loops: each latent node is its own neighbour once
    src, dst = g.lat_edge_index.numpy()
    assert (src == dst).sum() == g.num_mesh
    # edge attributes are [sin d, cos d] of a great-circle distance in radians
    for attr in (g.enc_edge_attr, g.lat_edge_attr, g.dec_edge_attr):
        assert torch.allclose((attr**2).sum(1), torch.ones(attr.shape[0]), atol=1e-5)


def test_encoder_and_decoder_use_reversed_rank_latent_uses_forward_rank():
    """encoder.py:80-84 / assimilator_decoder.py:72-77 vs encoder.py:262-263 - reproduced, not fixed."""
    g = build_forecast_graphs(regular_lat_lons(10.0), 2)
    G, M = g.num_grid, g.num_mesh
    enc_mesh_row = g.enc_edge_index[1].numpy() - G           # M-1-rank(cell of grid node i)
    dec_src, dec_dst = g.dec_edge_index.numpy()
    for i in (0, 17, G - 1):
        mine = dec_src[dec_dst == M + i]                       # reversed-rank ids of disk1(cell_i)
        assert enc_mesh_row[i] in mine                         # a cell is in its own disk
    # the latent graph indexes cells by forward rank: node r has a self loop (r, r)
    src, dst = g.lat_edge_index.numpy()
    assert np.array_equal(np.sort(src[src == dst]), np.arange(M))


def test_plans_are_destination_sorted_permutations_of_the_reference_order():
    g = build_forecast_graphs(regular_lat_lons(10.0), 2)
    for plan, ei, attr, off_s, off_d in ((g.enc_plan, g.enc_edge_index, g.enc_edge_attr, 0, g.num_grid),
                                         (g.lat_plan, g.lat_edge_index, g.lat_edge_attr, 0, 0),
                                         (g.dec_plan, g.dec_edge_index, g.dec_edge_attr, 0, g.num_mesh)):
        dst = plan.dst.numpy()
        assert np.all(np.diff(dst) >= 0)
        perm = plan.perm.numpy()
        assert np.array_equal(np.sort(perm), np.arange(plan.num_edges))
        assert np.array_equal(ei[0].numpy()[perm] - off_s, plan.src.numpy())
        assert np.array_equal(ei[1].numpy()[perm] - off_d, dst)
        assert torch.equal(attr[plan.perm], plan.edge_attr)
        assert plan.src.dtype == torch.int32 and plan.dst.dtype == torch.int32


def test_backward_index_structures():
    """dst_ptr / src_sorted / identity_ptr: the duals used by the backward pass (autograd.py)."""
    rs = np.random.RandomState(0)
    n_src, n_dst, e = 13, 9, 200
    src = rs.randint(0, n_src, size=e)
    dst = rs.randint(0, n_dst - 2, size=e)  # leaves two destinations without edges
    plan = plan_from_coo(src, dst, n_src, n_dst)
    ptr = plan.dst_ptr().numpy()
    assert ptr[0] == 0 and ptr[-1] == e and len(ptr) == n_dst + 1
    for n in range(n_dst):
        assert np.all(plan.dst.numpy()[ptr[n]:ptr[n + 1]] == n)
    perm, sptr = plan.src_sorted()
    perm, sptr = perm.numpy(), sptr.numpy()
    assert np.array_equal(np.sort(perm), np.arange(e)) and sptr[-1] == e
    for n in range(n_src):
        assert np.all(plan.src.numpy()[perm[sptr[n]:sptr[n + 1]]] == n)
    assert np.array_equal(plan.identity_ptr().numpy(), np.arange(e + 1))


def test_plan_from_coo_rejects_bad_indices_and_handles_empty():
    import pytest

    with pytest.raises(ValueError):
        plan_from_coo(np.array([0, 5]), np.array([0, 1]), 5, 2)
    p = plan_from_coo(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 3, 3)
    assert p.num_edges == 0 and p.dst_ptr().tolist() == [0, 0, 0, 0]
