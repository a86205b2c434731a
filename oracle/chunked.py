"""TEST INFRASTRUCTURE ONLY - the CPU oracle in slabs, for configurations whose whole-graph tensors do not fit a test.

``oracle/reference_math.py`` restates the reference forward on whole tensors: at BASELINE.json configs[4] (0.25 degree,
1 036 800 grid nodes, 7.25 M decoder edges) its decoder concat ``cat[x[row], x[col], e]`` alone is 22 GB
(graph_net_block.py:131-134).  This module evaluates THE SAME arithmetic - same functions of ``reference_math`` per
slab of rows / edges, shared-graph semantics (the reference's efficient-batching branches, encoder.py:168-196,
processor.py:106-122, assimilator_decoder.py:145-172, value-identical to the replicated branch by the reference's own
tests/models/layers/test_efficient_batching.py:53,91,145) - and the decoder only for a SAMPLE of grid rows: a grid row's
output depends on its own <= 7 incoming decoder edges and on the full processor output, nothing else
(assimilator_decoder.py:92-103: every decoder edge ends in a grid node; decoder.py:93).

Pinned by ``tests/test_oracle.py::test_chunked_oracle_equals_whole_tensor_oracle`` (all rows, 10 and 5 degree).
Only ``tests/`` may import this.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import reference_math as om

Tensor = torch.Tensor


def _mlp_slabs(p, prefix: str, x: Tensor, slab: int) -> Tensor:
    """``MLP.forward`` (graph_net_block.py:63-77) on row slabs."""
    return torch.cat([om.mlp(p, prefix, x[i:i + slab]) for i in range(0, x.shape[0], slab)], dim=0) if x.shape[0] else \
        om.mlp(p, prefix, x)


def _edge_block_slabs(p, prefix: str, x_src_tab: Tensor, x_dst_tab: Optional[Tensor], src: Tensor, dst: Tensor, e: Tensor,
                      n_dst: int, slab: int, want_edges: bool):
    """EdgeProcessor.forward + scatter_sum (graph_net_block.py:131-137, :188) in edge slabs: returns (agg [n_dst, D], e' or None).
    ``x_dst_tab`` None = destination rows are zeros (the decoder's lat/lon rows, assimilator_decoder.py:84,190-192)."""
    D = e.shape[1]
    agg = torch.zeros((n_dst, D), dtype=e.dtype)
    outs = []
    for i in range(0, e.shape[0], slab):
        s, d, ee = src[i:i + slab], dst[i:i + slab], e[i:i + slab]
        xs = x_src_tab[s]
        xd = x_dst_tab[d] if x_dst_tab is not None else torch.zeros((s.shape[0], xs.shape[1]), dtype=e.dtype)
        en = om.edge_processor(p, prefix + ".edge_model", xs, xd, ee)  # MLP(cat[src, dest, e]) + e
        agg.index_add_(0, d, en)  # == scatter_add_ over the destination column
        if want_edges:
            outs.append(en)
    return agg, (torch.cat(outs, dim=0) if want_edges else None)


def _node_update_slabs(p, prefix: str, x: Optional[Tensor], agg: Tensor, slab: int) -> Tensor:
    """NodeProcessor.forward after aggregation (graph_net_block.py:189-191); ``x`` None = zero rows."""
    outs = []
    for i in range(0, agg.shape[0], slab):
        a = agg[i:i + slab]
        xx = x[i:i + slab] if x is not None else torch.zeros_like(a)
        outs.append(om.mlp(p, prefix + ".node_model.node_mlp", torch.cat([xx, a], dim=-1)) + xx)
    return torch.cat(outs, dim=0)


def processor_output(p, g: dict, features_b: Tensor, slab: int = 1 << 16) -> Tensor:
    """Encoder (encoder.py:168-196) + Processor (processor.py:106-122) of ONE sample ``features_b`` [G, F] -> [M, D], in the
    reference's row order (mesh rows in reversed rank as the encoder emits them)."""
    G = features_b.shape[0]
    h3 = p["encoder.h3_nodes"]
    M = h3.shape[0]
    xg = _mlp_slabs(p, "encoder.node_encoder", features_b, slab)  # :205 grid rows
    xm = om.mlp(p, "encoder.node_encoder", h3)  # :205 mesh rows (same for every sample)
    e_enc = _mlp_slabs(p, "encoder.edge_encoder", g["enc_edge_attr"], slab)  # :206-208
    src, dst = g["enc_edge_index"][0], g["enc_edge_index"][1] - G  # targets G + mesh row -> mesh row
    # the encoder block: only mesh rows receive messages and only mesh rows are kept (:219-223); a grid row's node update
    # is computed and dropped by the reference
    agg, _ = _edge_block_slabs(p, "encoder.graph_processor.blocks.0", xg, xm, src, dst, e_enc, M, slab, False)
    x = _node_update_slabs(p, "encoder.graph_processor.blocks.0", xm, agg, slab)
    e = _mlp_slabs(p, "encoder.latent_edge_encoder", g["lat_edge_attr"], slab)  # :235-241
    ls, ld = g["lat_edge_index"][0], g["lat_edge_index"][1]
    i = 0
    while f"processor.graph_processor.blocks.{i}.edge_model.edge_mlp.model.0.weight" in p:  # graph_net_block.py:293-301
        pre = f"processor.graph_processor.blocks.{i}"
        agg, e = _edge_block_slabs(p, pre, x, x, ls, ld, e, M, slab, True)
        x = _node_update_slabs(p, pre, x, agg, slab)
        i += 1
    return x


def decoder_rows(p, g: dict, x_mesh: Tensor, start_rows: Tensor, rows: Tensor, slab: int = 1 << 16) -> Tensor:
    """Decoder (assimilator_decoder.py:145-172 + decoder.py:92-94) for the grid rows ``rows`` (int64 [R]) of one sample:
    ``x_mesh`` [M, D] = processor output, ``start_rows`` [R, F_out] = features[rows, :F_out].  Returns [R, F_out]."""
    M = x_mesh.shape[0]
    ei = g["dec_edge_index"]
    dst_grid = ei[1] - M
    G = int(g["num_grid"])
    slot = torch.full((G,), -1, dtype=torch.long)
    slot[rows] = torch.arange(rows.shape[0])
    sel = (slot[dst_grid] >= 0).nonzero(as_tuple=True)[0]  # decoder edges that end in a sampled row
    e = _mlp_slabs(p, "decoder.edge_encoder", g["dec_edge_attr"][sel], slab)  # :175-177 on the selected edges
    agg, _ = _edge_block_slabs(p, "decoder.graph_processor.blocks.0", x_mesh, None, ei[0][sel], slot[dst_grid[sel]], e,
                               rows.shape[0], slab, False)
    xg = _node_update_slabs(p, "decoder.graph_processor.blocks.0", None, agg, slab)  # grid rows enter as zeros (:190-192)
    return _mlp_slabs(p, "decoder.node_decoder", xg, slab) + start_rows  # :197, decoder.py:93


def forecaster_rows(p, g: dict, features: Tensor, rows: Tensor, feature_dim: int = 78, slab: int = 1 << 16) -> Tensor:
    """``GraphWeatherForecaster.forward`` (forecast.py:226-228) restricted to output rows ``rows``: [B, R, feature_dim]."""
    outs = []
    with torch.no_grad():
        for b in range(features.shape[0]):
            x = processor_output(p, g, features[b], slab)
            outs.append(decoder_rows(p, g, x, features[b][rows, :feature_dim], rows, slab))
    return torch.stack(outs, dim=0)
