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


def _check_seg_tiles(plan, st):
    """Invariants the segment-aligned kernels rely on (include/gw_amd.h: GW_EDGE_SEGMENT_TILES / _SPLIT)."""
    assert st.n_pad % 64 == 0 and st.src.shape[0] == st.dst.shape[0] == st.n_pad
    d = st.dst.numpy().reshape(-1, 64)
    valid = d >= 0
    assert (valid[:, :-1] >= valid[:, 1:]).all(), "padding columns sit at the end of a tile"
    assert valid[:, 0].all(), "no empty tile"
    assert (st.src.numpy()[~valid.reshape(-1)] == 0).all()
    # the padded list is the sorted list with gaps
    assert torch.equal(st.dst[st.pos], plan.dst) and torch.equal(st.src[st.pos], plan.src)
    assert (np.diff(st.pos.numpy()) > 0).all()
    # runs inside a tile are contiguous and destinations ascend through the padded list
    flat = d.reshape(-1)
    seen = flat[flat >= 0]
    assert (np.diff(seen) >= 0).all()
    slots = [len(np.unique(row[row >= 0])) for row in d]
    assert max(slots) == st.max_slots
    # a destination appears in more than one tile only as a split run: whole tiles, then the head of one more
    tiles_of = {}
    for t, row in enumerate(d):
        for v in np.unique(row[row >= 0]):
            tiles_of.setdefault(int(v), []).append(t)
    multi = {v: ts for v, ts in tiles_of.items() if len(ts) > 1}
    assert bool(multi) == st.split
    for v, ts in multi.items():
        assert ts == list(range(ts[0], ts[-1] + 1))
        for t in ts[:-1]:
            assert (d[t] == v).all()
        assert d[ts[-1]][0] == v
    assert st.complete == (len(tiles_of) == plan.n_dst)


def test_segment_aligned_tiles_of_the_forecaster_graphs():
    """GraphPlan.seg_tiles(): decoder (7 or 6 edges per grid node: 9 nodes per tile) and latent graph pack without splits; the
    encoder graph (polar mesh cells collect hundreds of grid nodes) only with split runs."""
    g = build_forecast_graphs(regular_lat_lons(2.0), 2)
    for plan in (g.dec_plan, g.lat_plan):
        st = plan.seg_tiles()
        assert st is not None and not st.split and st.complete and st.max_slots <= 16
        assert st.n_pad <= 1.05 * plan.num_edges + 64
        _check_seg_tiles(plan, st)
        assert plan.seg_tiles(split=True).n_pad == st.n_pad
    assert g.enc_plan.seg_tiles() is None
    st = g.enc_plan.seg_tiles(split=True)
    assert st is not None and st.split
    _check_seg_tiles(g.enc_plan, st)
    rows = torch.arange(g.enc_plan.num_edges, dtype=torch.float32)[:, None].expand(-1, 3).contiguous()
    padded = st.pad_rows(rows)
    assert torch.equal(padded[st.pos], rows) and padded.shape[0] == st.n_pad
    pb = st.pad_batched_rows(torch.cat([rows, rows + 0.5]), 2)
    assert torch.equal(pb.reshape(2, st.n_pad, 3)[1][st.pos], rows + 0.5)


def test_segment_aligned_tiles_of_random_graphs():
    rs = np.random.RandomState(3)
    for degrees in ([1, 2, 3], [0, 7, 6], [64, 1, 63], [0, 0, 200, 65, 2]):
        n_dst = 57
        deg = rs.choice(degrees, size=n_dst)
        deg[0] = max(deg[0], 1)
        dst = np.repeat(np.arange(n_dst), deg)
        src = rs.randint(0, 9, size=dst.size)
        plan = plan_from_coo(src, dst, 9, n_dst)
        long_run = int(deg.max()) > 64
        assert (plan.seg_tiles() is None) == long_run
        st = plan.seg_tiles(split=True)
        assert st.split == long_run
        _check_seg_tiles(plan, st)
    assert plan_from_coo(np.zeros(0, int), np.zeros(0, int), 3, 3).seg_tiles() is None
