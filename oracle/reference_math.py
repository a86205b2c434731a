"""TEST INFRASTRUCTURE ONLY - the CPU oracle.  Never imported by the product path.

A torch-only restatement of the arithmetic of the reference's
``GraphWeatherForecaster(lat_lons)(features)`` hot path (openclimatefix/graph_weather v1.0.140).
Every function cites the reference ``file:line`` it follows.  It works on a plain ``dict`` of
tensors keyed exactly like the reference ``state_dict`` and on plain graph arrays, so it needs
neither h3 nor torch_geometric nor torch_scatter.

Pinned (see ``tests/test_oracle_vs_reference.py`` and ``tests/golden``): outputs are compared
against the reference's *own source files executed in the build container* (``oracle/refload.py``:
third-party h3 / PyG / torch_scatter replaced by stand-ins of their public behaviour) and against
the golden vectors that run produced.  Graph *topology* parity with real h3 is unpinned
(SURVEY.md section 8c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# graph_net_block.py
# ----------------------------------------------------------------------------------------
def mlp(p: Params, prefix: str, x: Tensor) -> Tensor:
    """``MLP.forward`` - graph_net_block.py:45-61 (layer list) and :63-77 (forward).

    ``nn.Sequential`` indices: Linear at ``model.0, model.2, ...``, ReLU between, optional
    LayerNorm (eps 1e-5, affine) last.  The number of layers is read off the keys.
    """
    idx = 0
    lin = []
    while f"{prefix}.model.{idx}.weight" in p and p[f"{prefix}.model.{idx}.weight"].dim() == 2:
        lin.append(idx)
        idx += 2
    norm_idx = lin[-1] + 1
    has_norm = f"{prefix}.model.{norm_idx}.weight" in p
    h = x
    for n, i in enumerate(lin):
        h = F.linear(h, p[f"{prefix}.model.{i}.weight"], p[f"{prefix}.model.{i}.bias"])
        if n != len(lin) - 1:
            h = torch.relu(h)
    if has_norm:
        w = p[f"{prefix}.model.{norm_idx}.weight"]
        h = F.layer_norm(h, (w.shape[0],), w, p[f"{prefix}.model.{norm_idx}.bias"], 1e-5)
    return h


def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``torch_scatter.scatter_sum(edge_attr, col, dim=0, dim_size=N)`` - call site graph_net_block.py:188."""
    out = torch.zeros((dim_size, src.shape[1]), dtype=src.dtype)
    return out.scatter_add_(0, index.view(-1, 1).expand_as(src), src)


def edge_processor(p: Params, prefix: str, src: Tensor, dest: Tensor, edge_attr: Tensor) -> Tensor:
    """``EdgeProcessor.forward`` - graph_net_block.py:131-137: MLP(cat[src, dest, e]) + e."""
    out = torch.cat([src, dest, edge_attr], -1)
    out = mlp(p, prefix + ".edge_mlp", out)
    return out + edge_attr


def node_processor(p: Params, prefix: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
    """``NodeProcessor.forward`` - graph_net_block.py:184-193: MLP(cat[x, scatter_sum(e, col)]) + x."""
    col = edge_index[1]
    agg = scatter_sum(edge_attr, col, x.shape[0])
    out = torch.cat([x, agg], dim=-1)
    out = mlp(p, prefix + ".node_mlp", out)
    return out + x


def meta_block(p: Params, prefix: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tuple[Tensor, Tensor]:
    """PyG ``MetaLayer`` as built by graph_net_block.py:221-228: edge model first, then node model
    on the *updated* edge attributes; ``row`` = source, ``col`` = destination."""
    row, col = edge_index[0], edge_index[1]
    e = edge_processor(p, prefix + ".edge_model", x[row], x[col], edge_attr)
    xn = node_processor(p, prefix + ".node_model", x, edge_index, e)
    return xn, e


def graph_processor(p: Params, prefix: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tuple[Tensor, Tensor]:
    """``GraphProcessor.forward`` - graph_net_block.py:293-301: sequential blocks carrying x and e."""
    i = 0
    while f"{prefix}.blocks.{i}.edge_model.edge_mlp.model.0.weight" in p:
        x, edge_attr = meta_block(p, f"{prefix}.blocks.{i}", x, edge_index, edge_attr)
        i += 1
    return x, edge_attr


# ----------------------------------------------------------------------------------------
# encoder.py / processor.py / assimilator_decoder.py / decoder.py / forecast.py
# ----------------------------------------------------------------------------------------
def _replicate_index(edge_index: Tensor, batch: int, num_nodes: Optional[int]) -> Tensor:
    """encoder.py:212-218 / :226-234, assimilator_decoder.py:180-186.

    The reference offsets sample ``i`` by ``i*max(edge_index)+i``; this equals ``i*num_nodes`` whenever
    the highest node id is used by an edge (always true for the latent and decoder graphs, and for
    the encoder graph when the rank-0 cell holds a grid point - SURVEY.md appendix C.1).  Passing
    ``num_nodes=None`` reproduces the reference expression literally.
    """
    off = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    return torch.cat([edge_index + i * off for i in range(batch)], dim=1)


def encoder_forward(p: Params, g: dict, features: Tensor, literal_offsets: bool = False):
    """``Encoder.forward`` replicated branch - encoder.py:197-242.  ``g`` holds the graph arrays
    (``enc_edge_index`` [2,G] with targets ``G + (M-1-rank)``, ``enc_edge_attr`` [G,2],
    ``lat_edge_index`` [2,E_lat], ``lat_edge_attr`` [E_lat,2])."""
    B, G, _ = features.shape
    h3_nodes = p["encoder.h3_nodes"]
    M = h3_nodes.shape[0]
    feats = torch.cat([features, h3_nodes.unsqueeze(0).expand(B, -1, -1)], dim=1)  # :199-202
    feats = feats.reshape(B * (G + M), -1)  # :204
    out = mlp(p, "encoder.node_encoder", feats)  # :205
    edge_attr = mlp(p, "encoder.edge_encoder", g["enc_edge_attr"])  # :206-208
    edge_attr = edge_attr.repeat(B, 1)  # :210
    edge_index = _replicate_index(g["enc_edge_index"], B, None if literal_offsets else G + M)  # :212-218
    out, _ = graph_processor(p, "encoder.graph_processor", out, edge_index, edge_attr)  # :219
    out = out.reshape(B, G + M, -1)[:, G:, :].reshape(B * M, -1)  # :221-223
    lat_index = _replicate_index(g["lat_edge_index"], B, None if literal_offsets else M)  # :226-234
    lat_attr = mlp(p, "encoder.latent_edge_encoder", g["lat_edge_attr"].repeat(B, 1))  # :235-241
    return out, lat_index, lat_attr


def encoder_forward_shared(p: Params, g: dict, features: Tensor):
    """``Encoder.forward`` efficient-batching branch - encoder.py:168-196 (single shared graph)."""
    B, G, _ = features.shape
    outs = []
    edge_attr = mlp(p, "encoder.edge_encoder", g["enc_edge_attr"])
    for i in range(B):
        feat_i = torch.cat([features[i], p["encoder.h3_nodes"]], dim=0)
        out_i = mlp(p, "encoder.node_encoder", feat_i)
        out_i, _ = graph_processor(p, "encoder.graph_processor", out_i, g["enc_edge_index"], edge_attr)
        outs.append(out_i[G:])
    lat_attr = mlp(p, "encoder.latent_edge_encoder", g["lat_edge_attr"])
    return torch.cat(outs, dim=0), g["lat_edge_index"], lat_attr


def processor_forward(p: Params, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
    """``Processor.forward`` - processor.py:123-128 (thermalizer off): nodes only are returned."""
    out, _ = graph_processor(p, "processor.graph_processor", x, edge_index, edge_attr)
    return out


def processor_forward_shared(p: Params, x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: int) -> Tensor:
    """``Processor.forward`` efficient branch - processor.py:106-122."""
    n = x.shape[0] // batch
    outs = []
    for i in range(batch):
        o, _ = graph_processor(p, "processor.graph_processor", x[i * n:(i + 1) * n], edge_index, edge_attr)
        outs.append(o)
    return torch.cat(outs, dim=0)


def decoder_forward(p: Params, g: dict, processor_features: Tensor, start_features: Tensor,
                    literal_offsets: bool = False) -> Tensor:
    """``AssimilatorDecoder.forward`` replicated branch - assimilator_decoder.py:173-200, then the
    residual of ``Decoder.forward`` - decoder.py:92-94.  ``g['dec_edge_index']`` has sources
    ``M-1-rank(h)`` and targets ``M + i``."""
    B, G, _ = start_features.shape
    M = processor_features.shape[0] // B
    D = processor_features.shape[1]
    edge_attr = mlp(p, "decoder.edge_encoder", g["dec_edge_attr"]).repeat(B, 1)  # :175-178
    edge_index = _replicate_index(g["dec_edge_index"], B, None if literal_offsets else M + G)  # :180-186
    feats = processor_features.reshape(B, M, D)  # :189
    feats = torch.cat([feats, torch.zeros(B, G, D, dtype=feats.dtype)], dim=1).reshape(B * (M + G), D)  # :190-193
    out, _ = graph_processor(p, "decoder.graph_processor", feats, edge_index, edge_attr)  # :195
    out = mlp(p, "decoder.node_decoder", out)  # :197
    out = out.reshape(B, M + G, -1)[:, M:, :]  # :198-199
    return out + start_features  # decoder.py:93


def decoder_forward_shared(p: Params, g: dict, processor_features: Tensor, start_features: Tensor) -> Tensor:
    """``AssimilatorDecoder.forward`` efficient branch - assimilator_decoder.py:145-172."""
    B, G, _ = start_features.shape
    M = processor_features.shape[0] // B
    edge_attr = mlp(p, "decoder.edge_encoder", g["dec_edge_attr"])
    outs = []
    for i in range(B):
        feat_i = torch.cat([processor_features[i * M:(i + 1) * M],
                            torch.zeros(G, processor_features.shape[1], dtype=processor_features.dtype)], dim=0)
        out_i, _ = graph_processor(p, "decoder.graph_processor", feat_i, g["dec_edge_index"], edge_attr)
        outs.append(mlp(p, "decoder.node_decoder", out_i)[M:])
    return torch.stack(outs, dim=0) + start_features


def forecaster_forward(p: Params, g: dict, features: Tensor, feature_dim: int = 78, shared: bool = False,
                       literal_offsets: bool = False, return_intermediates: bool = False):
    """``GraphWeatherForecaster.forward`` - forecast.py:226-228 (constraint_type "none", no thermalizer)."""
    if shared:
        x, ei, ea = encoder_forward_shared(p, g, features)
        xp = processor_forward_shared(p, x, ei, ea, features.shape[0])
        y = decoder_forward_shared(p, g, xp, features[..., :feature_dim])
    else:
        x, ei, ea = encoder_forward(p, g, features, literal_offsets)
        xp = processor_forward(p, x, ei, ea)
        y = decoder_forward(p, g, xp, features[..., :feature_dim], literal_offsets)
    if return_intermediates:
        return y, {"enc_x": x, "lat_edge_attr": ea, "proc_x": xp}
    return y


def assimilator_forward(p: Params, g: dict, features: Tensor, analysis_dim: int) -> Tensor:
    """``GraphWeatherAssimilator.forward`` - analysis.py:136-150: AssimilatorEncoder.forward (assimilator_encoder.py:119-164),
    Processor, AssimilatorDecoder.forward.  ``g``: ``obs_edge_index`` [2, N] (targets N + (M-1-rank)), ``obs_edge_attr``
    [N, 3], ``lat_edge_index``, ``lat_edge_attr``, ``dec_edge_index``, ``dec_edge_attr``, ``num_mesh``, ``num_grid``.
    The encoder's mesh inputs are zeros (``h3_nodes`` is a plain zero tensor there, not a parameter, :80)."""
    B, N, F = features.shape
    M = g["num_mesh"]
    feats = torch.cat([features, torch.zeros((B, M, F), dtype=features.dtype)], dim=1).reshape(B * (N + M), F)  # :139-143
    out = mlp(p, "encoder.node_encoder", feats)  # :144
    edge_attr = mlp(p, "encoder.edge_encoder", g["obs_edge_attr"]).repeat(B, 1)  # :145-147
    edge_index = _replicate_index(g["obs_edge_index"], B, N + M)  # :149-152 (literal max+1 == N+M when the last cell is hit)
    out, _ = graph_processor(p, "encoder.graph_processor", out, edge_index, edge_attr)  # :153
    out = out.reshape(B, N + M, -1)[:, N:, :].reshape(B * M, -1)  # :155-157
    lat_index = _replicate_index(g["lat_edge_index"], B, M)
    lat_attr = mlp(p, "encoder.latent_edge_encoder", g["lat_edge_attr"].repeat(B, 1))  # :158-163
    xp = processor_forward(p, out, lat_index, lat_attr)
    # AssimilatorDecoder.forward (assimilator_decoder.py:145-200): as the forecaster's decoder without the residual add
    zeros = torch.zeros((B, g["num_grid"], analysis_dim), dtype=features.dtype)
    return decoder_forward(p, g, xp, zeros)


def nudging_weights(lat_lons) -> Tensor:
    """``BoundaryNudgingLayer._compute_relaxation_weights`` - regional_forecast.py:92-132 ([N, 1], float32 arithmetic)."""
    lats = torch.tensor([ll[0] for ll in lat_lons], dtype=torch.float32) * (np.pi / 180.0)
    lons = torch.tensor([ll[1] for ll in lat_lons], dtype=torch.float32) * (np.pi / 180.0)
    c_lat, c_lon = lats.mean(), lons.mean()
    a = torch.sin((lats - c_lat) / 2) ** 2 + torch.cos(lats) * torch.cos(c_lat) * torch.sin((lons - c_lon) / 2) ** 2
    dist = 2 * torch.asin(torch.sqrt(torch.clamp(a, 0.0, 1.0)))
    top = dist.max()
    return (dist / top if top > 0 else torch.zeros_like(dist)).unsqueeze(-1)


def regional_forward(p: Params, g: dict, features: Tensor, output_dim: int, global_context: Optional[Tensor] = None,
                     lat_lons=None) -> Tensor:
    """``RegionalForecaster.forward`` - regional_forecast.py:234-298.  ``g``: ``enc_edge_index`` [2, N] (coordinate i ->
    row N + index of its cell among the region's sorted unique cells), ``enc_edge_attr`` [N, 2], ``lat_edge_index`` /
    ``lat_edge_attr`` between those cells, ``h3_indices`` (rows of ``h3_embeddings``) - dynamic_graph_builder.py:41-130."""
    B, N, _ = features.shape
    regional_h3 = p["h3_embeddings"][torch.as_tensor(g["h3_indices"], dtype=torch.long)]  # :256
    enc_e = mlp(p, "edge_encoder", g["enc_edge_attr"])  # :258
    lat_e = mlp(p, "latent_edge_encoder", g["lat_edge_attr"])  # :259
    dec_index = g["enc_edge_index"].flip(0)  # :261 reversed encoder edges
    dec_e = mlp(p, "decoder_edge_encoder", g["enc_edge_attr"])  # :262
    outs = []
    for i in range(B):  # :264-281
        nodes = mlp(p, "node_encoder", torch.cat([features[i], regional_h3], dim=0))
        nodes, _ = graph_processor(p, "encoder_gnn", nodes, g["enc_edge_index"], enc_e)
        h3f = processor_forward(p, nodes[N:], g["lat_edge_index"], lat_e)
        dec_nodes = torch.cat([torch.zeros((N, h3f.shape[1]), dtype=features.dtype), h3f], dim=0)
        dec_nodes, _ = graph_processor(p, "decoder_gnn", dec_nodes, dec_index, dec_e)
        outs.append(mlp(p, "node_decoder", dec_nodes[:N]))
    out = torch.stack(outs, dim=0) + features[..., :output_dim]  # :283-284
    if global_context is not None and "nudging.blend_mlp.model.0.weight" in p:  # :287-289, :85-90
        prior = nudging_weights(lat_lons).to(features.dtype).unsqueeze(0).expand(B, -1, -1)
        corr = mlp(p, "nudging.blend_mlp", torch.cat([out, global_context, prior], dim=-1))
        alpha = torch.clamp(prior + corr, 0.0, 1.0)
        out = (1 - alpha) * out + alpha * global_context
    return out


# ----------------------------------------------------------------------------------------
# losses.py
# ----------------------------------------------------------------------------------------
def normalized_mse_loss(pred: Tensor, target: Tensor, lat_lons, feature_variance: Optional[Tensor] = None,
                        normalize: bool = False) -> Tensor:
    """``NormalizedMSELoss`` - losses.py:35-42 (weights) and :66-94 (forward, minus prints/asserts)."""
    unique_lats = sorted(set(lat for lat, _ in lat_lons))
    weights = torch.tensor([np.cos(lat * np.pi / 180.0) for lat in unique_lats], dtype=torch.float)
    out = (pred - target) ** 2
    if normalize:
        out = out / feature_variance
    out = out.mean(-1)
    B = out.shape[0]
    num_nodes = int(np.prod(out.shape[1:]))
    out = out.reshape(B, num_nodes)
    num_unique = weights.shape[0]
    num_lon = num_nodes // num_unique
    weight_grid = weights.unsqueeze(1).expand(num_unique, num_lon).reshape(1, num_nodes).expand(B, num_nodes)
    return (out * weight_grid.to(out.dtype)).mean()


# ----------------------------------------------------------------------------------------
# helpers shared by the tests
# ----------------------------------------------------------------------------------------
def to_dtype(p: Params, dtype) -> Params:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in p.items()}


def graphs_to_dtype(g: dict, dtype) -> dict:
    return {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in g.items()}
