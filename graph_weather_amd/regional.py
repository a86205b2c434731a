"""``RegionalForecaster`` / ``DynamicGraphBuilder`` / ``BoundaryNudgingLayer`` on the hot path's kernels
(reference ``graph_weather/models/regional_forecast.py:16-298`` and ``layers/dynamic_graph_builder.py:13-155``;
SURVEY.md 8f rows 3 and 4).

Same constructor arguments, attribute names and ``state_dict`` keys as the reference.  What differs is how it runs:

* graphs are built in vectorised numpy over the mesh provider (the reference walks Python loops of h3 calls per
  coordinate) and come with destination-sorted plans for the HIP kernels; ``DynamicGraphBuilder`` still returns the
  reference-format ``edge_index`` / ``edge_attr`` objects, cached by the identity of ``lat_lons`` like the reference;
* the forward runs once for the whole batch on shared graphs (the reference loops over batch elements,
  regional_forecast.py:264-281), with the layer-1 split, cached batch-independent embeddings in inference and the
  zero-row shortcuts of the global decoder (observation placeholders are zeros, :276-277);
* ``node_decoder`` carries ``LayerNorm(output_dim)`` here (regional_forecast.py:223-230 passes ``norm_type``): the head
  kernel normalises over its ``n_out`` real features.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import autograd as ag
from . import mesh as _mesh
from . import ops
from .graphs import GraphPlan, _sincos, plan_from_coo
from .layers import FEED_ZERO, Feed, GraphProcessor, KeyedCache, MLP, Processor, _autograd_on, _check_native_dims, _version_key
from .ops import Operand
from .utils import validate_lat_lons

__all__ = ["GraphData", "DynamicGraphBuilder", "BoundaryNudgingLayer", "RegionalForecasterConfig", "RegionalForecaster"]


class GraphData:
    """Attribute bag standing where the reference uses ``torch_geometric.data.Data`` (dynamic_graph_builder.py:8)."""

    def __init__(self, edge_index: torch.Tensor, edge_attr: torch.Tensor):
        self.edge_index, self.edge_attr = edge_index, edge_attr

    def to(self, device) -> "GraphData":
        self.edge_index, self.edge_attr = self.edge_index.to(device), self.edge_attr.to(device)
        return self


class DynamicGraphBuilder:
    """dynamic_graph_builder.py:13-155: encoder / decoder / latent graphs of an arbitrary set of coordinates."""

    def __init__(self, resolution: int = 2, provider=None):
        self.resolution = resolution
        self._provider = provider if provider is not None else _mesh.get_provider()
        self._vector = isinstance(self._provider, _mesh.H3Like)
        if self._vector:
            m = _mesh.get_mesh(resolution)
            self.all_h3 = [_mesh.H3Like._cid(resolution, i) for i in range(m.num)]  # already in sorted (rank) order
        else:  # pragma: no cover - real h3 is absent in this image
            h3 = self._provider
            self.all_h3 = sorted(h3.uncompact_cells(h3.get_res0_cells(), resolution))
        self.global_h3_map = {cell: i for i, cell in enumerate(self.all_h3)}
        self._prev_lat_lons = None
        self._cached_encoder_graph: Optional[GraphData] = None
        self._cached_decoder_graph: Optional[GraphData] = None
        self._cached_latent_graph: Optional[GraphData] = None
        self._cached_h3_indices: Optional[List[int]] = None
        self._native = None  # (enc_plan, lat_plan, dec_plan) of the cached coordinates, host side
        self._native_dev = {}
        self.generation = 0  # bumped whenever the cached graphs change: key of everything derived from them

    # ---- cell assignment -------------------------------------------------------------------------------------------
    def _ranks(self, lat_lons) -> np.ndarray:
        """Global rank (position in ``all_h3``) of the cell containing each coordinate."""
        ll = np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2)
        if self._vector:
            return _mesh.get_mesh(self.resolution).locate(ll[:, 0], ll[:, 1])
        h3 = self._provider  # pragma: no cover
        return np.array([self.global_h3_map[h3.latlng_to_cell(lat, lon, self.resolution)] for lat, lon in ll], dtype=np.int64)

    def _assign_h3_cells(self, lat_lons) -> Tuple[List[str], List[str], dict]:
        """dynamic_graph_builder.py:32-39."""
        ranks = self._ranks(lat_lons)
        h3_cells = [self.all_h3[r] for r in ranks]
        unique_cells = [self.all_h3[r] for r in np.unique(ranks)]  # ascending rank == sorted cell ids
        return h3_cells, unique_cells, {cell: i for i, cell in enumerate(unique_cells)}

    def _centres(self, ranks: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        if self._vector:
            m = _mesh.get_mesh(self.resolution)
            return m.lat[ranks], m.lon[ranks]
        h3 = self._provider  # pragma: no cover
        c = np.array([h3.cell_to_latlng(self.all_h3[r]) for r in ranks], dtype=np.float64).reshape(-1, 2)
        return c[:, 0], c[:, 1]

    def _disks(self, ranks: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(degree per requested cell, concatenated disk-1 members in the provider's grid_disk order) as global ranks."""
        if self._vector:
            ptr, idx = _mesh.get_mesh(self.resolution).disk1_csr()
            deg = np.diff(ptr)[ranks]
            start = np.repeat(ptr[ranks], deg)
            within = np.arange(int(deg.sum()), dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)
            return deg, idx[start + within]
        h3 = self._provider  # pragma: no cover
        disks = [[self.global_h3_map[h] for h in h3.grid_disk(self.all_h3[r], 1)] for r in ranks]
        return np.array([len(d) for d in disks], dtype=np.int64), np.array([h for d in disks for h in d], dtype=np.int64)

    # ---- the three graphs ------------------------------------------------------------------------------------------
    def _encoder_arrays(self, lat_lons):
        ll = np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2)
        ranks = self._ranks(lat_lons)
        uniq, inv = np.unique(ranks, return_inverse=True)
        clat, clon = self._centres(ranks)
        attr = _sincos(_mesh.haversine_rads(ll[:, 0], ll[:, 1], clat, clon))
        return ll, ranks, uniq, inv.astype(np.int64), attr

    def build_encoder_graph(self, lat_lons) -> Tuple[GraphData, List[int]]:
        """dynamic_graph_builder.py:41-68: one edge per coordinate, coordinate i -> row N + (index of its cell among the
        sorted unique cells); ``h3_indices`` = global ranks of those cells (rows of the embedding table)."""
        ll, _, uniq, inv, attr = self._encoder_arrays(lat_lons)
        n = ll.shape[0]
        edge_index = torch.from_numpy(np.stack([np.arange(n, dtype=np.int64), n + inv]))
        return GraphData(edge_index, torch.from_numpy(attr)), [int(u) for u in uniq]

    def build_decoder_graph(self, lat_lons) -> GraphData:
        """dynamic_graph_builder.py:70-100: every coordinate receives from the disk-1 of its cell; source rows index the
        sorted union of those disks."""
        ll = np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2)
        ranks = self._ranks(lat_lons)
        _, hood = self._disks(np.unique(ranks))
        hood = np.unique(hood)
        deg, h = self._disks(ranks)
        node = np.repeat(np.arange(ll.shape[0], dtype=np.int64), deg)
        hlat, hlon = self._centres(h)
        attr = _sincos(_mesh.haversine_rads(ll[node, 0], ll[node, 1], hlat, hlon))
        edge_index = torch.from_numpy(np.stack([np.searchsorted(hood, h).astype(np.int64), hood.size + node]))
        return GraphData(edge_index, torch.from_numpy(attr))

    def _latent_arrays(self, uniq: np.ndarray):
        lookup = {int(r): i for i, r in enumerate(uniq)}
        deg, h = self._disks(uniq)
        src_cell = np.repeat(uniq, deg)
        keep = np.isin(h, uniq)
        src_cell, h = src_cell[keep], h[keep]
        src = np.array([lookup[int(c)] for c in src_cell], dtype=np.int64)
        dst = np.array([lookup[int(c)] for c in h], dtype=np.int64)
        alat, alon = self._centres(src_cell)
        blat, blon = self._centres(h)
        return src, dst, _sincos(_mesh.haversine_rads(alat, alon, blat, blon))

    def build_latent_graph(self, unique_cells: Sequence) -> GraphData:
        """dynamic_graph_builder.py:102-130: disk-1 edges (self loops included) between the supplied cells only.
        ``unique_cells``: cell ids of the provider (or global ranks)."""
        uniq = np.array([c if isinstance(c, (int, np.integer)) else self.global_h3_map[c] for c in unique_cells], dtype=np.int64)
        src, dst, attr = self._latent_arrays(uniq)
        return GraphData(torch.from_numpy(np.stack([src, dst]).reshape(2, -1)), torch.from_numpy(attr.reshape(-1, 2)))

    def __call__(self, lat_lons) -> Tuple[GraphData, GraphData, GraphData, List[int]]:
        """dynamic_graph_builder.py:132-155 (cached on the identity of ``lat_lons``)."""
        if lat_lons is self._prev_lat_lons:
            return (self._cached_encoder_graph, self._cached_decoder_graph, self._cached_latent_graph, self._cached_h3_indices)
        validate_lat_lons(lat_lons)
        ll, _, uniq, inv, attr = self._encoder_arrays(lat_lons)
        n, c = ll.shape[0], uniq.size
        enc = GraphData(torch.from_numpy(np.stack([np.arange(n, dtype=np.int64), n + inv])), torch.from_numpy(attr))
        dec = self.build_decoder_graph(lat_lons)
        ls, ld, la = self._latent_arrays(uniq)
        lat = GraphData(torch.from_numpy(np.stack([ls, ld]).reshape(2, -1)), torch.from_numpy(la.reshape(-1, 2)))
        # plans of the native forward: encoder edges, the same edges reversed (regional_forecast.py:259-261), latent graph
        self._native = (plan_from_coo(np.arange(n), inv, n, c, attr), plan_from_coo(ls, ld, c, c, la.reshape(-1, 2)),
                        plan_from_coo(inv, np.arange(n), c, n, attr))
        self._native_dev = {}
        self.generation += 1
        self._prev_lat_lons = lat_lons
        self._cached_encoder_graph, self._cached_decoder_graph, self._cached_latent_graph = enc, dec, lat
        self._cached_h3_indices = [int(u) for u in uniq]
        return enc, dec, lat, self._cached_h3_indices

    def native_plans(self, lat_lons, device) -> Tuple[GraphPlan, GraphPlan, GraphPlan, torch.Tensor]:
        """(encoder plan, latent plan, reversed-encoder plan, embedding rows) on ``device`` for the cached coordinates."""
        self(lat_lons)
        key = str(device)
        if key not in self._native_dev:
            idx = torch.tensor(self._cached_h3_indices, dtype=torch.long, device=device)
            self._native_dev[key] = tuple(p.to(device) for p in self._native) + (idx,)
        return self._native_dev[key]


# ---------------------------------------------------------------------------------------------------------------------
class _NudgingFunction(torch.autograd.Function):
    """alpha = clamp(prior + blend_mlp(cat[regional, global, prior]), 0, 1); out = (1 - alpha) regional + alpha global
    (regional_forecast.py:85-90) as one kernel; backward = one kernel + the generic weight-gradient products."""

    @staticmethod
    def forward(ctx, mlp_input, feat, w1, b1, w2, b2):
        from . import _lib

        rows, hidden = int(mlp_input.shape[0]), int(w1.shape[0])
        out = torch.empty((rows, feat), dtype=torch.float32, device=mlp_input.device)
        w1t = w1.detach().t().contiguous()
        _lib.check(_lib.lib().gw_nudging_forward(rows, feat, hidden, mlp_input.data_ptr(), int(mlp_input.stride(0)), w1t.data_ptr(),
                                                 b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(),
                                                 torch.cuda.current_stream(out.device).cuda_stream), "gw_nudging_forward")
        ctx.feat = feat
        ctx.save_for_backward(mlp_input, w1, w1t, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib

        mlp_input, w1, w1t, b1, w2, b2 = ctx.saved_tensors
        rows, hidden, feat = int(mlp_input.shape[0]), int(w1.shape[0]), ctx.feat
        dev = mlp_input.device
        dout = dout.contiguous()
        d_in = torch.zeros_like(mlp_input)  # gradient reaches the regional columns only (context and prior are data)
        dz = torch.empty((rows, hidden), dtype=torch.float32, device=dev)
        hid = torch.empty((rows, hidden), dtype=torch.float32, device=dev)
        dcorr = torch.empty((rows, 1), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gw_nudging_backward(rows, feat, hidden, mlp_input.data_ptr(), int(mlp_input.stride(0)), w1.data_ptr(),
                                                  w1t.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), dout.data_ptr(),
                                                  d_in.data_ptr(), int(d_in.stride(0)), dz.data_ptr(), hid.data_ptr(),
                                                  dcorr.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "gw_nudging_backward")
        gw1, gb1 = torch.zeros_like(w1), torch.zeros_like(b1)
        gw2, gb2 = torch.zeros_like(w2), torch.zeros_like(b2)
        ag.gemm_tn_acc(dz, mlp_input, gw1, colsum=gb1)
        ag.gemm_tn_acc(dcorr, hid, gw2, colsum=gb2)
        return d_in, None, gw1, gb1, gw2, gb2


class BoundaryNudgingLayer(nn.Module):
    """regional_forecast.py:44-132."""

    def __init__(self, feature_dim: int, hidden_dim: int = 64):
        super().__init__()
        if feature_dim > 256 or hidden_dim > 256:
            raise NotImplementedError("graph_weather_amd: the nudging kernel handles feature_dim and hidden_dim up to 256")
        self.feature_dim = feature_dim
        self.blend_mlp = MLP(feature_dim * 2 + 1, 1, hidden_dim, 1, None)

    def forward(self, regional: torch.Tensor, global_context: torch.Tensor, lat_lons: list) -> torch.Tensor:
        """regional_forecast.py:69-90."""
        if not regional.is_cuda:
            raise RuntimeError("graph_weather_amd: tensors must be on a HIP device - there is no CPU path")
        B, N, F = (int(s) for s in regional.shape)
        alpha_prior = self._compute_relaxation_weights(lat_lons, regional.device)
        alpha_prior = alpha_prior.unsqueeze(0).expand(B, -1, -1)
        mlp_input = torch.cat([regional, global_context.to(regional.dtype), alpha_prior], dim=-1).reshape(B * N, 2 * F + 1)
        lin1, lin2 = self.blend_mlp.model[0], self.blend_mlp.model[2]
        out = _NudgingFunction.apply(mlp_input, F, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
        return out.reshape(B, N, F)

    @staticmethod
    def _compute_relaxation_weights(lat_lons: list, device) -> torch.Tensor:
        """regional_forecast.py:92-132: haversine distance from the region centroid, normalised to [0, 1] ([N, 1])."""
        ll = np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2)
        lats = torch.from_numpy(ll[:, 0].astype(np.float32)) * (math.pi / 180.0)
        lons = torch.from_numpy(ll[:, 1].astype(np.float32)) * (math.pi / 180.0)
        c_lat, c_lon = lats.mean(), lons.mean()
        a = torch.sin((lats - c_lat) / 2) ** 2 + torch.cos(lats) * torch.cos(c_lat) * torch.sin((lons - c_lon) / 2) ** 2
        dist = 2 * torch.asin(torch.sqrt(torch.clamp(a, 0.0, 1.0)))
        top = dist.max()
        weights = dist / top if top > 0 else torch.zeros_like(dist)
        return weights.unsqueeze(-1).to(device)


@dataclass
class RegionalForecasterConfig:
    """regional_forecast.py:16-41."""

    resolution: int = 2
    feature_dim: int = 78
    aux_dim: int = 24
    output_dim: Optional[int] = None
    node_dim: int = 256
    edge_dim: int = 256
    num_blocks: int = 9
    hidden_dim_processor_node: int = 256
    hidden_dim_processor_edge: int = 256
    hidden_layers_processor_node: int = 2
    hidden_layers_processor_edge: int = 2
    hidden_dim_decoder: int = 128
    hidden_layers_decoder: int = 2
    norm_type: str = "LayerNorm"
    use_checkpointing: bool = False
    enable_nudging: bool = False
    nudging_hidden_dim: int = 64

    def build(self) -> "RegionalForecaster":
        return RegionalForecaster(self)


class RegionalForecaster(nn.Module):
    """regional_forecast.py:135-298."""

    def __init__(self, config: RegionalForecasterConfig):
        super().__init__()
        c = config
        self.config = c
        input_dim = c.feature_dim + c.aux_dim
        output_dim = c.output_dim if c.output_dim is not None else c.feature_dim
        self.output_dim = output_dim
        self.nudging = BoundaryNudgingLayer(output_dim, c.nudging_hidden_dim) if c.enable_nudging else None
        self.graph_builder = DynamicGraphBuilder(resolution=c.resolution)
        self.h3_embeddings = nn.Parameter(torch.zeros(_mesh.num_cells(c.resolution), input_dim))
        mk = dict(norm_type=c.norm_type, use_checkpointing=c.use_checkpointing)
        self.node_encoder = MLP(input_dim, c.node_dim, c.hidden_dim_processor_node, c.hidden_layers_processor_node, **mk).as_table()
        self.edge_encoder = MLP(2, c.edge_dim, c.hidden_dim_processor_edge, c.hidden_layers_processor_edge, **mk).as_table()
        self.encoder_gnn = GraphProcessor(1, c.node_dim, c.edge_dim, c.hidden_dim_processor_node, c.hidden_dim_processor_edge,
                                          c.hidden_layers_processor_node, c.hidden_layers_processor_edge, c.norm_type,
                                          use_checkpointing=c.use_checkpointing)
        self.latent_edge_encoder = MLP(2, c.edge_dim, c.hidden_dim_processor_edge, c.hidden_layers_processor_edge, **mk).as_table()
        self.processor = Processor(input_dim=c.node_dim, edge_dim=c.edge_dim, num_blocks=c.num_blocks,
                                   hidden_dim_processor_edge=c.hidden_dim_processor_edge,
                                   hidden_layers_processor_node=c.hidden_layers_processor_node,
                                   hidden_dim_processor_node=c.hidden_dim_processor_node,
                                   hidden_layers_processor_edge=c.hidden_layers_processor_edge, mlp_norm_type=c.norm_type)
        self.decoder_edge_encoder = MLP(2, c.edge_dim, c.hidden_dim_processor_edge, c.hidden_layers_processor_edge, **mk).as_table()
        self.decoder_gnn = GraphProcessor(1, c.node_dim, c.edge_dim, c.hidden_dim_processor_node, c.hidden_dim_processor_edge,
                                          c.hidden_layers_processor_node, c.hidden_layers_processor_edge, c.norm_type,
                                          use_checkpointing=c.use_checkpointing)
        self.node_decoder = MLP(c.node_dim, output_dim, c.hidden_dim_decoder, c.hidden_layers_decoder, **mk)
        self._cache = KeyedCache()

    def _cached(self, name: str, params, token, fn):
        """Batch-independent tensors of the inference path: recomputed when a parameter or the coordinate set changes."""
        if _autograd_on(self):
            return fn()
        return self._cache.get(name, (_version_key(params), token), fn)

    def _is_wide(self) -> bool:
        from . import wide

        e, d = self.encoder_gnn.blocks[0], self.decoder_gnn.blocks[0]
        return (wide.is_wide(self.node_encoder, self.edge_encoder, self.latent_edge_encoder, self.decoder_edge_encoder, self.node_decoder,
                             e.edge_model.edge_mlp, e.node_model.node_mlp, d.edge_model.edge_mlp, d.node_model.node_mlp)
                or wide.processor_is_wide(self.processor.graph_processor))

    def _forward_wide(self, feats, B, N, enc_plan, lat_plan, dec_plan, rows):
        """The same forward for widths above 256, on the generic kernels of wide.py (regional_forecast.py:258-284)."""
        from . import wide

        C = enc_plan.n_dst
        xo = wide.mlp_rows(self.node_encoder, feats)
        xm = wide.mlp_rows(self.node_encoder, self.h3_embeddings[rows])
        e_enc = wide.mlp_rows(self.edge_encoder, enc_plan.edge_attr)
        x, _ = wide.block(self.encoder_gnn.blocks[0], enc_plan, B, xo, N, xm, 0, e_enc, 0)
        e_lat = wide.mlp_rows(self.latent_edge_encoder, lat_plan.edge_attr)
        x, _ = wide.run_blocks(self.processor.graph_processor, x, lat_plan, e_lat, True, B, False)
        e_dec = wide.mlp_rows(self.decoder_edge_encoder, dec_plan.edge_attr)
        xg, _ = wide.block(self.decoder_gnn.blocks[0], dec_plan, B, x.contiguous(), C, None, 0, e_dec, 0)
        return wide.mlp_rows(self.node_decoder, xg, residual=feats[:, :self.output_dim])

    def forward(self, features: torch.Tensor, lat_lons: list, global_context: Optional[torch.Tensor] = None) -> torch.Tensor:
        """regional_forecast.py:234-298."""
        if not features.is_cuda:
            raise RuntimeError("graph_weather_amd: features must be on a HIP device - there is no CPU path")
        if features.dtype != torch.float32:
            raise RuntimeError("graph_weather_amd: features must be float32")
        features = features.contiguous()
        B, N, F = (int(s) for s in features.shape)
        if N != len(lat_lons):
            raise RuntimeError("features and lat_lons disagree on the number of coordinates")
        dev = features.device
        enc_plan, lat_plan, dec_plan, rows = self.graph_builder.native_plans(lat_lons, dev)
        token = (self.graph_builder.generation, str(dev))
        C = enc_plan.n_dst
        train = _autograd_on(self, features)
        if F < self.output_dim:
            raise RuntimeError("graph_weather_amd: features need at least output_dim = %d channels for the residual "
                               "(regional_forecast.py:283-284)" % self.output_dim)
        feats = features.reshape(B * N, F)
        if self._is_wide():
            out = self._forward_wide(feats, B, N, enc_plan, lat_plan, dec_plan, rows).reshape(B, N, self.output_dim)
            if self.nudging is not None and global_context is not None:
                out = self.nudging(out, global_context, lat_lons)
            return out
        _check_native_dims(*self.encoder_gnn._dims)
        _check_native_dims(*self.decoder_gnn._dims)

        # ---- encode: coordinates + regional cells through the bipartite block (:258, :266-269) ----
        xo = self.node_encoder.run(feats, B * N, N)
        enc_params = list(self.node_encoder.parameters()) + [self.h3_embeddings]
        xm = self._cached("cells", enc_params, token,
                          lambda: self.node_encoder.table(self.h3_embeddings[rows] if train else self.h3_embeddings.detach()[rows]))
        e_enc = self._cached("enc_e", list(self.edge_encoder.parameters()), token, lambda: self.edge_encoder.table(enc_plan.edge_attr))
        blk = self.encoder_gnn.blocks[0]

        def enc_projections():
            if train:
                return (ag.project(blk.edge_model.edge_mlp, (1,), xm, C, C)[0], ag.project(blk.edge_model.edge_mlp, (2,), e_enc, N, N)[0],
                        ag.project(blk.node_model.node_mlp, (0,), xm, C, C)[0])
            pm_e, pm_n = blk.edge_model.edge_mlp.packed(), blk.node_model.node_mlp.packed()
            return (ops.project_forward([pm_e.w1[1]], Operand(xm, C, 256), C, C)[0],
                    ops.project_forward([pm_e.w1[2]], Operand(e_enc, N, 256), N, N)[0],
                    ops.project_forward([pm_n.w1[0]], Operand(xm, C, 256), C, C)[0])

        pd_xm, pe, px_xm = self._cached("enc_proj", enc_params + list(self.edge_encoder.parameters()) + list(blk.parameters()),
                                        token, enc_projections)
        x, _ = blk.run(B, enc_plan, Feed(xo, N, "raw"), Feed(pd_xm, 0, "proj"), Feed(pe, 0, "proj"), e_enc, 0,
                       Feed(px_xm, 0, "proj"), xm, 0, False, dev, tag="regional_encoder_edge")

        # ---- process: message passing between the regional cells (:259, :272) ----
        e_lat = self._cached("lat_e", list(self.latent_edge_encoder.parameters()), token,
                             lambda: self.latent_edge_encoder.table(lat_plan.edge_attr))
        x, _ = self.processor.graph_processor.run_plan(x, lat_plan, e_lat, True, B, False)

        # ---- decode: reversed encoder edges into zero placeholders (:262, :275-279), head, residual (:284) ----
        e_dec = self._cached("dec_e", list(self.decoder_edge_encoder.parameters()), token,
                             lambda: self.decoder_edge_encoder.table(dec_plan.edge_attr))
        dblk = self.decoder_gnn.blocks[0]
        mlp_e = dblk.edge_model.edge_mlp
        if train:
            ps = ag.project(mlp_e, (0,), x.contiguous(), B * C, C)[0]
            pe_d = ag.project(mlp_e, (2,), e_dec, N, N)[0]
        else:
            ps = ops.project_forward([mlp_e.packed().w1[0]], Operand(x.contiguous(), C, 256), B * C, C)[0]
            pe_d = self._cached("dec_pe", list(self.decoder_edge_encoder.parameters()) + list(dblk.parameters()), token,
                                lambda: ops.project_forward([mlp_e.packed().w1[2]], Operand(e_dec, N, 256), N, N)[0])
        xg, _ = dblk.run(B, dec_plan, Feed(ps, C, "proj"), FEED_ZERO, Feed(pe_d, 0, "proj"), e_dec, 0, FEED_ZERO, None, 0, False, dev,
                         tag="regional_decoder_edge")
        res = Operand(feats, N, self.output_dim)
        y = self.node_decoder.run(xg, B * N, N, residual=res)
        out = y.reshape(B, N, self.output_dim)

        # ---- boundary nudging (:287-289) ----
        if self.nudging is not None and global_context is not None:
            out = self.nudging(out, global_context, lat_lons)
        return out
