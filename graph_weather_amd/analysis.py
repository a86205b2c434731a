"""``GraphWeatherAssimilator`` / ``AssimilatorEncoder`` - analysis from raw observations on the same kernels
(reference ``graph_weather/models/analysis.py:52-150`` and ``layers/assimilator_encoder.py:36-242``).

Same constructor arguments and ``state_dict`` keys.  The observation -> mesh graph depends on where the observations
are, so it is built per distinct ``lat_lon_heights`` tensor (vectorised numpy instead of the reference's per-point h3
loops) and cached; everything after that is the forecaster's machinery: node / edge embeddings, one message-passing
block with the layer-1 split, the processor on the latent graph, ``AssimilatorDecoder``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch
from torch import nn

from . import ops
from .graphs import build_latent_graph, build_observation_graph
from .layers import (AssimilatorDecoder, Feed, GraphProcessor, KeyedCache, MLP, Processor, _autograd_on, _check_native_dims, _ver,
                     _version_key)
from .ops import Operand

try:
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass


class AssimilatorEncoder(nn.Module):
    """assimilator_encoder.py:36-242."""

    def __init__(self, resolution: int = 2, input_dim: int = 2, output_dim: int = 256, output_edge_dim: int = 256,
                 hidden_dim_processor_node: int = 256, hidden_dim_processor_edge: int = 256,
                 hidden_layers_processor_node: int = 2, hidden_layers_processor_edge: int = 2,
                 mlp_norm_type: str = "LayerNorm", use_checkpointing: bool = False):
        super().__init__()
        self.use_checkpointing = use_checkpointing
        self.output_dim, self.input_dim, self.resolution = output_dim, input_dim, resolution
        self.lat_edge_index, self.lat_edge_attr, self._lat_plan = build_latent_graph(resolution)
        self.num_h3 = self._lat_plan.n_dst
        # assimilator_encoder.py:80: a plain zero tensor, neither a parameter nor a buffer (so not in the state_dict)
        self.h3_nodes = torch.zeros((self.num_h3, input_dim), dtype=torch.float)
        self.output_edge_dim = output_edge_dim
        self.node_encoder = MLP(input_dim, output_dim, hidden_dim_processor_node, hidden_layers_processor_node, mlp_norm_type,
                                use_checkpointing).as_table()
        self.edge_encoder = MLP(3, output_edge_dim, hidden_dim_processor_edge, hidden_layers_processor_edge, mlp_norm_type,
                                use_checkpointing).as_table()  # [sin d, cos d, height]
        self.latent_edge_encoder = MLP(2, output_edge_dim, hidden_dim_processor_edge, hidden_layers_processor_edge,
                                       mlp_norm_type, use_checkpointing).as_table()
        self.graph_processor = GraphProcessor(1, output_dim, output_edge_dim, hidden_dim_processor_node,
                                              hidden_dim_processor_edge, hidden_layers_processor_node,
                                              hidden_layers_processor_edge, mlp_norm_type)
        self._dev = {}
        self._cache = KeyedCache()  # "obs" (observation graph per position tensor) + the batch-independent embeddings

    def _latent_plan(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self._lat_plan.to(device)
        return self._dev[key]

    def _observation_plan(self, lat_lon_heights: torch.Tensor, device):
        """assimilator_encoder.py:166-219, once per distinct position tensor."""
        llh = lat_lon_heights.reshape(-1, 3) if lat_lon_heights.dim() == 2 else lat_lon_heights[0]
        key = (lat_lon_heights.data_ptr(), _ver(lat_lon_heights), tuple(lat_lon_heights.shape), str(device))
        return self._cache.get("obs", key, lambda: build_observation_graph(llh.detach().cpu().numpy(), self.resolution)[2].to(device),
                               hold=lat_lon_heights)  # (the entry holds the tensor: its address stays taken)

    def _cached(self, name, params, fn):
        if _autograd_on(self):
            return fn()
        return self._cache.get(name, _version_key(params), fn)

    def latent_edge_embedding(self, plan) -> torch.Tensor:
        return self._cached("lat_e", list(self.latent_edge_encoder.parameters()), lambda: self.latent_edge_encoder.table(plan.edge_attr))

    def _is_wide(self) -> bool:
        from . import wide as wd

        b = self.graph_processor.blocks[0]
        return wd.is_wide(self.node_encoder, self.edge_encoder, self.latent_edge_encoder, b.edge_model.edge_mlp, b.node_model.node_mlp)

    def encode(self, features: torch.Tensor, lat_lon_heights: torch.Tensor, wide: bool = False) -> torch.Tensor:
        """assimilator_encoder.py:137-157 -> mesh node features [(B*M), D] (reversed-rank order).  ``wide``: take the generic
        kernels even if this encoder alone would fit the fused ones (another part of the model is wider than 256)."""
        if not features.is_cuda:
            raise RuntimeError("graph_weather_amd: features must be on a HIP device - there is no CPU path")
        B, N, F = (int(s) for s in features.shape)
        dev = features.device
        plan = self._observation_plan(lat_lon_heights, dev)
        if plan.n_src != N:
            raise RuntimeError("features and lat_lon_heights disagree on the number of observations")
        feats = features.contiguous().reshape(B * N, F)
        zeros_in = self.h3_nodes.to(dev)
        if wide or self._is_wide():  # widths above 256: the generic kernels of wide.py
            from . import wide as wd

            xo = wd.mlp_rows(self.node_encoder, feats)
            xm = wd.mlp_rows(self.node_encoder, zeros_in)
            e = wd.mlp_rows(self.edge_encoder, plan.edge_attr)
            x, _ = wd.block(self.graph_processor.blocks[0], plan, B, xo, N, xm, 0, e, 0)
            return x
        _check_native_dims(*self.graph_processor._dims)
        xo = self.node_encoder.table(feats)  # observation rows
        xm = self._cached("mesh", list(self.node_encoder.parameters()), lambda: self.node_encoder.table(zeros_in))
        e = self.edge_encoder.table(plan.edge_attr)  # depends on the observation positions: not cached across graphs
        blk = self.graph_processor.blocks[0]
        train = _autograd_on(self, features)
        M = self.num_h3
        if train:
            from . import autograd as ag

            pd_xm = ag.project(blk.edge_model.edge_mlp, (1,), xm, M, M)[0]
            pe = ag.project(blk.edge_model.edge_mlp, (2,), e, N, N)[0]
            px_xm = ag.project(blk.node_model.node_mlp, (0,), xm, M, M)[0]
        else:
            pm_e, pm_n = blk.edge_model.edge_mlp.packed(), blk.node_model.node_mlp.packed()
            pd_xm = ops.project_forward([pm_e.w1[1]], Operand(xm, M, 256), M, M)[0]
            pe = ops.project_forward([pm_e.w1[2]], Operand(e, N, 256), N, N)[0]
            px_xm = ops.project_forward([pm_n.w1[0]], Operand(xm, M, 256), M, M)[0]
        x, _ = blk.run(B, plan, Feed(xo, N, "raw"), Feed(pd_xm, 0, "proj"), Feed(pe, 0, "proj"), e, 0, Feed(px_xm, 0, "proj"), xm,
                       0, False, dev, tag="assimilator_encoder_edge")
        return x

    def forward(self, features: torch.Tensor, lat_lon_heights: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """assimilator_encoder.py:119-164: (mesh features, replicated latent edge_index, latent edge features) in
        reference order."""
        B = int(features.shape[0])
        x = self.encode(features, lat_lon_heights)
        plan = self._latent_plan(features.device)
        e_sorted = self.latent_edge_embedding(plan)
        e_ref = torch.empty_like(e_sorted)
        e_ref[plan.perm] = e_sorted
        if self.output_dim != 256:
            x = x[:, :self.output_dim]
        if self.output_edge_dim != 256:
            e_ref = e_ref[:, :self.output_edge_dim]
        ei = self.lat_edge_index.to(features.device)
        M = self.num_h3
        return x, torch.cat([ei + i * M for i in range(B)], dim=1), e_ref.repeat(B, 1)


@dataclass
class GraphWeatherAssimilatorConfig:
    """analysis.py:11-49."""

    output_lat_lons: list
    resolution: int = 2
    observation_dim: int = 2
    analysis_dim: int = 78
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

    def build(self) -> "GraphWeatherAssimilator":
        return GraphWeatherAssimilator(**self.__dict__)


class GraphWeatherAssimilator(nn.Module, PyTorchModelHubMixin):
    """analysis.py:52-150."""

    def __init__(self, output_lat_lons: list, resolution: int = 2, observation_dim: int = 2, analysis_dim: int = 78,
                 node_dim: int = 256, edge_dim: int = 256, num_blocks: int = 9, hidden_dim_processor_node: int = 256,
                 hidden_dim_processor_edge: int = 256, hidden_layers_processor_node: int = 2,
                 hidden_layers_processor_edge: int = 2, hidden_dim_decoder: int = 128, hidden_layers_decoder: int = 2,
                 norm_type: str = "LayerNorm", use_checkpointing: bool = False):
        super().__init__()
        self.encoder = AssimilatorEncoder(resolution=resolution, input_dim=observation_dim, output_dim=node_dim,
                                          output_edge_dim=edge_dim, hidden_dim_processor_edge=hidden_dim_processor_edge,
                                          hidden_layers_processor_node=hidden_layers_processor_node,
                                          hidden_dim_processor_node=hidden_dim_processor_node,
                                          hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type,
                                          use_checkpointing=use_checkpointing)
        self.processor = Processor(input_dim=node_dim, edge_dim=edge_dim, num_blocks=num_blocks,
                                   hidden_dim_processor_edge=hidden_dim_processor_edge,
                                   hidden_layers_processor_node=hidden_layers_processor_node,
                                   hidden_dim_processor_node=hidden_dim_processor_node,
                                   hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type)
        self.decoder = AssimilatorDecoder(lat_lons=[tuple(ll) for ll in output_lat_lons], resolution=resolution, input_dim=node_dim,
                                          output_dim=analysis_dim, output_edge_dim=edge_dim,
                                          hidden_dim_processor_edge=hidden_dim_processor_edge,
                                          hidden_layers_processor_node=hidden_layers_processor_node,
                                          hidden_dim_processor_node=hidden_dim_processor_node,
                                          hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type,
                                          hidden_dim_decoder=hidden_dim_decoder, hidden_layers_decoder=hidden_layers_decoder,
                                          use_checkpointing=use_checkpointing)

    def forward(self, features: torch.Tensor, obs_lat_lon_heights: torch.Tensor) -> torch.Tensor:
        """analysis.py:136-150; fused native path (shared graphs, data stays in the native layouts between the stages)."""
        from . import wide as wd

        B = int(features.shape[0])
        plan = self.encoder._latent_plan(features.device)
        if self.encoder._is_wide() or wd.processor_is_wide(self.processor.graph_processor) or wd.decoder_is_wide(self.decoder):
            x = self.encoder.encode(features, obs_lat_lon_heights, wide=True)
            e_lat = wd.mlp_rows(self.encoder.latent_edge_encoder, plan.edge_attr)
            x, _ = wd.run_blocks(self.processor.graph_processor, x, plan, e_lat, True, B, False)
            return wd.decode(self.decoder, x, B)
        x = self.encoder.encode(features, obs_lat_lon_heights)
        e_lat = self.encoder.latent_edge_embedding(plan)
        x, _ = self.processor.graph_processor.run_plan(x, plan, e_lat, True, B, False)
        return self.decoder.decode(x, B)
