"""``GraphWeatherForecaster`` - the drop-in model API of the hot path (reference ``graph_weather/models/forecast.py``).

Same constructor arguments, attributes (``encoder`` / ``processor`` / ``decoder``, ``grid_shape``, ``node_to_grid``),
``state_dict`` keys and ``forward(features[B, G, feature_dim+aux_dim], t=0) -> [B, G, output_dim]`` contract as
forecast.py:61-247; the arithmetic runs in the HIP kernels of libgw_amd.so.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .graphed import AutoGraphModule
from .graphs import TopologyRecord, build_forecast_graphs
from .layers import Decoder, Encoder, Processor, fused_forward, set_compute_dtype

try:  # forecast.py:8,61 - hub mixin gives save_pretrained / from_pretrained / push_to_hub
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass


@dataclass
class GraphWeatherForecasterConfig:
    """forecast.py:14-58."""

    lat_lons: list
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
    constraint_type: str = "none"
    use_thermalizer: bool = False

    def build(self) -> "GraphWeatherForecaster":
        return GraphWeatherForecaster(**self.__dict__)


class GraphWeatherForecaster(AutoGraphModule, TopologyRecord, torch.nn.Module, PyTorchModelHubMixin):
    """forecast.py:61-247 (constraint layer and thermalizer are optional extras outside the hot path)."""

    def __init__(self, lat_lons: list, resolution: int = 2, feature_dim: int = 78, aux_dim: int = 24,
                 output_dim: Optional[int] = None, node_dim: int = 256, edge_dim: int = 256, num_blocks: int = 9,
                 hidden_dim_processor_node: int = 256, hidden_dim_processor_edge: int = 256,
                 hidden_layers_processor_node: int = 2, hidden_layers_processor_edge: int = 2,
                 hidden_dim_decoder: int = 128, hidden_layers_decoder: int = 2, norm_type: str = "LayerNorm",
                 use_checkpointing: bool = False, constraint_type: str = "none", use_thermalizer: bool = False):
        super().__init__()
        if constraint_type != "none":
            raise NotImplementedError("PhysicalConstraintLayer (default 'none', forecast.py:82) is outside the hot path")
        self.feature_dim = feature_dim
        self.constraint_type = constraint_type
        self.use_thermalizer = use_thermalizer
        if output_dim is None:
            output_dim = self.feature_dim
        if output_dim != feature_dim:
            # decoder.py:93 adds features[..., :feature_dim] to a [B, G, output_dim] tensor: the reference fails there with a
            # shape error; fail at construction instead of reading the wrong residual columns
            raise RuntimeError("graph_weather_amd: output_dim (%d) must equal feature_dim (%d): the decoder adds the input's "
                               "first feature_dim channels to its output (decoder.py:93)" % (output_dim, feature_dim))
        self.output_dim = output_dim
        lat_lons = [tuple(ll) for ll in lat_lons]
        unique_lats = sorted(set(lat for lat, _ in lat_lons))
        unique_lons = sorted(set(lon for _, lon in lat_lons))
        self.grid_shape = (len(unique_lats), len(unique_lons))
        self.original_lat_lons = list(lat_lons)
        self._create_grid_mapping(unique_lats, unique_lons)
        graphs = build_forecast_graphs(lat_lons, resolution)  # built once, shared by encoder and decoder
        self.encoder = Encoder(lat_lons=lat_lons, resolution=resolution, input_dim=feature_dim + aux_dim,
                               output_dim=node_dim, output_edge_dim=edge_dim,
                               hidden_dim_processor_edge=hidden_dim_processor_edge,
                               hidden_layers_processor_node=hidden_layers_processor_node,
                               hidden_dim_processor_node=hidden_dim_processor_node,
                               hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type,
                               use_checkpointing=use_checkpointing, _graphs=graphs)
        self.processor = Processor(input_dim=node_dim, edge_dim=edge_dim, num_blocks=num_blocks,
                                   hidden_dim_processor_edge=hidden_dim_processor_edge,
                                   hidden_layers_processor_node=hidden_layers_processor_node,
                                   hidden_dim_processor_node=hidden_dim_processor_node,
                                   hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type,
                                   use_thermalizer=use_thermalizer)
        self.decoder = Decoder(lat_lons=lat_lons, resolution=resolution, input_dim=node_dim, output_dim=output_dim,
                               output_edge_dim=edge_dim, hidden_dim_processor_edge=hidden_dim_processor_edge,
                               hidden_layers_processor_node=hidden_layers_processor_node,
                               hidden_dim_processor_node=hidden_dim_processor_node,
                               hidden_layers_processor_edge=hidden_layers_processor_edge, mlp_norm_type=norm_type,
                               hidden_dim_decoder=hidden_dim_decoder, hidden_layers_decoder=hidden_layers_decoder,
                               use_checkpointing=use_checkpointing, _graphs=graphs)

    def set_compute_dtype(self, dtype) -> "GraphWeatherForecaster":
        """float32 (default), bfloat16 or "bf16x3" (split-operand) matrix products - see ``layers.set_compute_dtype``."""
        set_compute_dtype(self, dtype)
        return self

    def graphed(self, warmup: int = 3):
        """``fg = model.graphed(); y = fg(features)``: the inference forward replayed from one HIP graph (``graphed.ForwardGraph``;
        re-captured by itself when weights, input shape or compute dtype change).  Pays off where the step is launch-bound: the
        bf16x3 mode at small batch."""
        from .graphed import ForwardGraph

        return ForwardGraph(self, warmup=warmup)

    def set_deterministic(self, flag: bool = True) -> "GraphWeatherForecaster":
        """Bitwise reproducible inference forward - see ``layers.set_deterministic``."""
        from .layers import set_deterministic

        set_deterministic(self, flag)
        return self

    def _create_grid_mapping(self, unique_lats, unique_lons):
        """forecast.py:178-192 (vectorised; identical (row, col) pairs)."""
        lo_lat, hi_lat = min(unique_lats), max(unique_lats)
        lo_lon, hi_lon = min(unique_lons), max(unique_lons)
        nlat, nlon = len(unique_lats), len(unique_lons)
        self.node_to_grid = []
        for lat, lon in self.original_lat_lons:
            row = int((lat - lo_lat) / (hi_lat - lo_lat) * (nlat - 1)) if hi_lat > lo_lat else 0
            col = int((lon - lo_lon) / (hi_lon - lo_lon) * (nlon - 1)) if hi_lon > lo_lon else 0
            self.node_to_grid.append((row, col))

    def graph_to_grid(self, graph_tensor):
        """forecast.py:194-205: [B, N, C] -> [B, C, H, W]."""
        batch_size, num_nodes, features = graph_tensor.shape
        grid = torch.zeros(batch_size, features, *self.grid_shape)
        for node_idx, (row, col) in enumerate(self.node_to_grid):
            grid[..., row, col] = graph_tensor[..., node_idx, :]
        return grid

    def grid_to_graph(self, grid_tensor):
        """forecast.py:207-213: [B, C, H, W] -> [B, N, C]."""
        batch_size, features, H, W = grid_tensor.shape
        graph = torch.zeros(batch_size, H * W, features)
        for node_idx, (row, col) in enumerate(self.node_to_grid):
            graph[..., node_idx, :] = grid_tensor[..., row, col]
        return graph

    def forward(self, features: torch.Tensor, t: int = 0) -> torch.Tensor:
        """forecast.py:215-247 with constraint_type == "none".  Fused path: data stays in the native layouts
        (dst-sorted shared graph, cached batch-independent embeddings) between encoder, processor and decoder.  In eval() under
        no_grad() the call replays its own HIP graph from the third call of a shape on (graphed.AutoGraphModule)."""
        y = self._auto_graph_step(features)
        return y if y is not None else self._forward_eager(features)

    def _forward_eager(self, features: torch.Tensor) -> torch.Tensor:
        if not features.is_cuda:
            raise RuntimeError("graph_weather_amd: features must be on a HIP device - there is no CPU path")
        if features.dtype != torch.float32:
            raise RuntimeError("graph_weather_amd: features must be float32")
        if features.dim() != 3 or features.shape[2] < self.output_dim:
            raise RuntimeError("graph_weather_amd: features must be [B, nodes, >= %d channels]" % self.output_dim)
        features = features.contiguous()
        B, G = int(features.shape[0]), self.encoder.num_latlons
        return fused_forward(self.encoder, self.processor, self.decoder, features, features.reshape(B * G, features.shape[2]))
