"""graph_weather_amd/routes.py: every kernel-form decision of the message-passing path as a pure function, tested without a GPU -
against the boolean expressions the call sites of layers.py carried before (rounds 2-5) on an exhaustive grid of their inputs,
and against the C side's own table where the library re-checks eligibility."""
import itertools

import torch

from graph_weather_amd import _lib, routes
from graph_weather_amd.routes import BF16X3, MlpForm

DTYPES = (torch.float32, torch.bfloat16, BF16X3)


def _forms():
    for d, n_mid, lnw, norm in itertools.product(DTYPES, (1, 2), (0, 128), (True, False)):
        yield MlpForm(d, n_mid, lnw, norm)


def test_edge_route_table():
    for m, n_edges, train in itertools.product(_forms(), (0, 41162), (False, True)):
        r = routes.edge_route(m, n_edges, train)
        old_tiled = (not train) and m.dtype == torch.bfloat16 and m.n_mid == 1 and m.ln_width == 0 and m.has_norm and n_edges > 0
        assert (r == routes.EDGE_TILES_BF16) == old_tiled
        if train:
            assert r == routes.EDGE_AUTOGRAD
        elif m.dtype == torch.float32:
            assert r == routes.EDGE_ROWS_FP32
        elif m.dtype == BF16X3:
            assert r == routes.EDGE_ROWS_X3
        else:
            assert r in (routes.EDGE_TILES_BF16, routes.EDGE_ROWS_BF16)


def test_edge_hand_over_and_post_products():
    R = (routes.EDGE_AUTOGRAD, routes.EDGE_ROWS_FP32, routes.EDGE_ROWS_X3, routes.EDGE_TILES_BF16, routes.EDGE_ROWS_BF16, None)
    for r, need, nxt in itertools.product(R[:-1], (False, True), R):
        k = routes.edge_out_kind(r, need, nxt)
        old = "tiles" if (r == routes.EDGE_TILES_BF16 and need and nxt == routes.EDGE_TILES_BF16) else need
        assert k == old
    assert routes.post_products_half(routes.EDGE_TILES_BF16) and not routes.post_products_half(routes.EDGE_ROWS_X3)
    assert not routes.post_products_half(None)


def test_block_route_table():
    for e, nd, n_edges, wide, auto, det in itertools.product(_forms(), DTYPES, (0, 64800), (False, True), (False, True), (False, True)):
        r = routes.block_route(e, nd, n_edges, wide, auto, det)
        old_team = (not (wide or auto or det) and e.dtype == torch.bfloat16 and nd == torch.bfloat16 and e.n_mid == 1
                    and e.ln_width == 0 and e.has_norm and n_edges > 0)
        old_split = not (wide or auto) and e.dtype == BF16X3 and nd == BF16X3 and n_edges > 0
        assert (r == routes.BLOCK_TEAM) == old_team and (r == routes.BLOCK_SPLIT) == old_split
        assert r in (routes.BLOCK_TEAM, routes.BLOCK_SPLIT, routes.BLOCK_ROWS)


def test_stack_routes():
    ok = MlpForm(torch.bfloat16, 1, 0, True)
    blocks = [(ok, torch.bfloat16, False)] * 9
    assert routes.stack_on_segment_tiles(blocks, 41162, 16)
    assert not routes.stack_on_segment_tiles(blocks, 41162, 17) and not routes.stack_on_segment_tiles(blocks, 41162, None)
    assert not routes.stack_on_segment_tiles(blocks, 0, 16) and not routes.stack_on_segment_tiles([], 41162, 16)
    for bad in ((MlpForm(torch.bfloat16, 2, 0, True), torch.bfloat16, False), (MlpForm(torch.bfloat16, 1, 128, True), torch.bfloat16, False),
                (MlpForm(torch.bfloat16, 1, 0, False), torch.bfloat16, False), (ok, torch.float32, False), (ok, torch.bfloat16, True),
                (MlpForm(BF16X3, 1, 0, True), BF16X3, False)):
        assert not routes.stack_on_segment_tiles(blocks[:4] + [bad] + blocks[5:], 41162, 16)
    # per-sample edge features at block 0 never take the segment route
    for train, want, lo, shared in itertools.product((False, True), (False, True), (0, 3), (False, True)):
        assert routes.segment_route_allowed(train, want, lo, shared) == (not train and not want and lo == 0 and shared)
    assert routes.mesh_streams(0, [torch.float32] * 9, 2) == 2 and routes.mesh_streams(0, [BF16X3] * 9, 5) == 2
    assert routes.mesh_streams(0, [torch.float32] * 9, 1) == 1 and routes.mesh_streams(0, [torch.bfloat16] * 9, 16) == 1
    assert routes.mesh_streams(0, [torch.float32, torch.bfloat16], 4) == 1
    assert routes.mesh_streams(4, [torch.bfloat16], 2) == 2 and routes.mesh_streams(1, [torch.float32], 8) == 1


def test_node_update_form_equals_the_c_table():
    """csrc/gw_noders.hip: node_rs_groups, exported as gw_node_update_row_split_groups (ABI v17, pure host logic)."""
    L = _lib.lib()
    for n in list(range(0, 70)) + [4095, 4096, 4097, 5882, 8192, 8193, 11764, 12288, 12289, 47056, 129600, 1 << 31]:
        assert L.gw_node_update_row_split_groups(n) == routes.node_update_row_split_groups(n), n
    assert routes.node_update_row_split_groups(5882) == 2 and routes.node_update_row_split_groups(11764) == 3  # the mesh at batch 1, 2
    f = routes.node_update_form
    assert f(torch.float32, 11764, True, "raw", 1, 0, False) == "row_split_cg3"
    assert f(BF16X3, 5882, True, "proj", 1, 0, False) == "row_split_cg2" and f(BF16X3, 900, True, "zero", 1, 0, False) == "row_split_cg1"
    for args in ((torch.bfloat16, 5882, True, "raw", 1, 0, False), (torch.float32, 129600, True, "raw", 1, 0, False),
                 (torch.float32, 5882, False, "raw", 1, 0, False), (torch.float32, 5882, True, "other", 1, 0, False),
                 (torch.float32, 5882, True, "raw", 2, 0, False), (torch.float32, 5882, True, "raw", 1, 128, False),
                 (torch.float32, 5882, True, "raw", 1, 0, True)):
        assert f(*args) == routes.NODE_COLS64


def test_bench_reads_the_per_kernel_pmc_summaries_of_the_round():
    """bench.py attaches counter traffic to the gather-scatter stage (the processor's edge update) from the newest committed per-kernel
    PMC summary of the workload - a silent None there (renamed kernel, other grid size) would hide wasted re-reads: the lookup is
    checked against the files under profiles/ for both modes at the batch-2 grid (1 287 tiles x 256 threads)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    grid = ((2 * 41162 + 63) // 64) * 256
    for cfg, prefix in (("c2", "edge_kernel<true, 2>"), ("c2x3", "chainx3_kernel<8, true, 3, 16, 16, 1,")):
        traffic, name, busy = bench.pmc_kernel_traffic(cfg, (prefix,), grid)
        assert traffic is not None and name.endswith("_pmc_%s_all_kernels.json" % cfg), (cfg, name)
        algorithmic = 2 * (2 * 41162 * 256 * 4 + 2 * 5882 * 256 * 4)  # 96.3 MB per sample
        assert 1.0 <= traffic / algorithmic <= 2.5, (cfg, traffic / algorithmic)
        assert busy is None or 0.05 <= busy <= 1.0
    assert bench.pmc_kernel_traffic("c2", ("no_such_kernel<",), grid) == (None, None, None)
    total, d, name = bench.pmc_traffic("c2")
    assert total is not None and name.startswith("r") and 0.5 <= d.get("mfma_busy_frac", 0.0) <= 1.0
