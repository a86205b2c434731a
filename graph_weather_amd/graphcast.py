"""``GraphCast`` - the reference's encoder / processor / decoder wrapper with hierarchical checkpointing controls
(``graph_weather/models/graphcast/model.py:21-345``).

Same constructor arguments, sub-module names (``encoder`` / ``processor`` / ``decoder`` => same ``state_dict`` keys) and
setters.  The forward runs the native fused path (one shared destination-sorted graph for the whole batch - what the
reference calls ``efficient_batching`` - cached batch-independent embeddings, HIP kernels).  The checkpointing setters
select, under autograd, which segments keep no activations and are recomputed in the backward (``autograd.recompute``:
the segment's forward runs the inference kernels, its backward re-runs it with the activation saves) - the same
memory-for-recompute trade as ``torch.utils.checkpoint`` in graphcast/model.py:212-285.  Inference keeps nothing anyway.
"""
from __future__ import annotations

import torch

from . import autograd as ag
from . import wide
from .graphed import AutoGraphModule
from .graphs import TopologyRecord, build_forecast_graphs
from .layers import Decoder, Encoder, Processor, fused_forward


class GraphCast(AutoGraphModule, TopologyRecord, torch.nn.Module):
    def __init__(self, lat_lons: list, resolution: int = 2, input_dim: int = 78, output_dim: int = 78, hidden_dim: int = 256,
                 num_processor_blocks: int = 9, hidden_layers: int = 2, mlp_norm_type: str = "LayerNorm",
                 use_checkpointing: bool = False, efficient_batching: bool = False):
        super().__init__()
        self.lat_lons = lat_lons
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.efficient_batching = efficient_batching
        graphs = build_forecast_graphs([tuple(ll) for ll in lat_lons], resolution)
        self.encoder = Encoder(lat_lons=lat_lons, resolution=resolution, input_dim=input_dim, output_dim=hidden_dim,
                               output_edge_dim=hidden_dim, hidden_dim_processor_node=hidden_dim,
                               hidden_dim_processor_edge=hidden_dim, hidden_layers_processor_node=hidden_layers,
                               hidden_layers_processor_edge=hidden_layers, mlp_norm_type=mlp_norm_type,
                               use_checkpointing=use_checkpointing, efficient_batching=efficient_batching, _graphs=graphs)
        self.processor = Processor(input_dim=hidden_dim, edge_dim=hidden_dim, num_blocks=num_processor_blocks,
                                   hidden_dim_processor_node=hidden_dim, hidden_dim_processor_edge=hidden_dim,
                                   hidden_layers_processor_node=hidden_layers, hidden_layers_processor_edge=hidden_layers,
                                   mlp_norm_type=mlp_norm_type, use_checkpointing=use_checkpointing)
        self.decoder = Decoder(lat_lons=lat_lons, resolution=resolution, input_dim=hidden_dim, output_dim=output_dim,
                               hidden_dim_processor_node=hidden_dim, hidden_dim_processor_edge=hidden_dim,
                               hidden_layers_processor_node=hidden_layers, hidden_layers_processor_edge=hidden_layers,
                               mlp_norm_type=mlp_norm_type, hidden_dim_decoder=hidden_dim, hidden_layers_decoder=hidden_layers,
                               use_checkpointing=use_checkpointing, efficient_batching=efficient_batching, _graphs=graphs)
        self._checkpoint_model = False
        self._checkpoint_encoder = False
        self._checkpoint_processor_segments = 0
        self._checkpoint_decoder = False

    # ---- graphcast/model.py:120-175: hierarchical checkpointing controls ----
    def set_checkpoint_model(self, checkpoint_flag: bool):
        self._checkpoint_model = checkpoint_flag
        if checkpoint_flag:
            self._checkpoint_encoder = False
            self._checkpoint_processor_segments = 0
            self._checkpoint_decoder = False

    def set_checkpoint_encoder(self, checkpoint_flag: bool):
        self._checkpoint_encoder = checkpoint_flag

    def set_checkpoint_processor(self, checkpoint_segments: int):
        self._checkpoint_processor_segments = checkpoint_segments
        self.processor.set_checkpoint_segments(checkpoint_segments)

    def set_checkpoint_decoder(self, checkpoint_flag: bool):
        self._checkpoint_decoder = checkpoint_flag

    def _any_wide(self) -> bool:
        """Wide (layer-by-layer, unpadded rows) versus fused (256-float tables) is decided ONCE for the whole model, as
        ``layers.fused_forward`` does: the two paths hand different table layouts from stage to stage."""
        return (wide.encoder_is_wide(self.encoder) or wide.processor_is_wide(self.processor.graph_processor)
                or wide.decoder_is_wide(self.decoder))

    def _encode(self, features: torch.Tensor) -> torch.Tensor:
        if self._any_wide():
            return wide.encode(self.encoder, features)
        return self.encoder.encode(features)

    def _process(self, x: torch.Tensor, B: int, dev) -> torch.Tensor:
        _, lat_plan = self.encoder._plans(dev)
        if self._any_wide():
            return wide.run_blocks(self.processor.graph_processor, x, lat_plan, wide.latent_edges(self.encoder, lat_plan), True, B, False)[0]
        e_lat = self.encoder.latent_edge_embedding(lat_plan)
        return self.processor.graph_processor.run_plan(x, lat_plan, e_lat, True, B, False)[0]

    def _decode(self, x: torch.Tensor, features: torch.Tensor) -> torch.Tensor:
        B, G = int(features.shape[0]), self.encoder.num_latlons
        res = features.reshape(B * G, features.shape[2])
        if self._any_wide():
            return wide.decode(self.decoder, x, B, residual=res)
        return self.decoder.decode(x, B, residual=res)

    def _custom_forward(self, features: torch.Tensor) -> torch.Tensor:
        """graphcast/model.py:212-262 (hierarchical checkpointing: encoder / processor / decoder segments; the processor's
        -1 / N segments are handled inside ``GraphProcessor.run_plan``)."""
        B, dev = int(features.shape[0]), features.device
        grad = torch.is_grad_enabled()
        if not (grad and (self._checkpoint_encoder or self._checkpoint_decoder)):
            # no encoder / decoder segment to recompute: the forecaster's fused forward (inference: projection-free blocks, one
            # launch chain in native layouts; wide models through wide.forward; processor segments inside run_plan)
            G = self.encoder.num_latlons
            return fused_forward(self.encoder, self.processor, self.decoder, features, features.reshape(B * G, features.shape[2]))
        if grad and self._checkpoint_encoder:
            x = ag.recompute(lambda f: (self._encode(f),), (features,), self.encoder)[0]
        else:
            x = self._encode(features)
        x = self._process(x, B, dev)
        if grad and self._checkpoint_decoder:
            return ag.recompute(lambda x_, f: (self._decode(x_, f),), (x, features), self.decoder)[0]
        return self._decode(x, features)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        """graphcast/model.py:264-286: ``decoder(processor(encoder(features)), features)`` - the input itself is the residual,
        so ``input_dim`` must equal ``output_dim`` as in the reference (decoder.py:93).  In eval() under no_grad() the call
        replays its own HIP graph from the third call of a shape on (graphed.AutoGraphModule)."""
        y = self._auto_graph_step(features)
        return y if y is not None else self._forward_eager(features)

    def _forward_eager(self, features: torch.Tensor) -> torch.Tensor:
        if not features.is_cuda:
            raise RuntimeError("graph_weather_amd: features must be on a HIP device - there is no CPU path")
        if features.dim() != 3 or features.shape[2] != self.output_dim:
            raise RuntimeError("graph_weather_amd: GraphCast adds its input to its output (decoder.py:93): features must be "
                               "[B, nodes, output_dim = %d], got %s" % (self.output_dim, tuple(features.shape)))
        features = features.contiguous()
        if torch.is_grad_enabled() and self._checkpoint_model:  # graphcast/model.py:274-281: the whole model is one segment
            return ag.recompute(lambda f: (self._custom_forward(f),), (features,), self)[0]
        return self._custom_forward(features)


class GraphCastConfig:
    """graphcast/model.py:289-345: the five canned checkpointing strategies."""

    @staticmethod
    def no_checkpointing(model: GraphCast):
        model.set_checkpoint_model(False)
        model.set_checkpoint_encoder(False)
        model.set_checkpoint_processor(0)
        model.set_checkpoint_decoder(False)

    @staticmethod
    def full_checkpointing(model: GraphCast):
        model.set_checkpoint_model(True)

    @staticmethod
    def balanced_checkpointing(model: GraphCast):
        model.set_checkpoint_model(False)
        model.set_checkpoint_encoder(True)
        model.set_checkpoint_processor(-1)  # graphcast/model.py:318
        model.set_checkpoint_decoder(True)

    @staticmethod
    def processor_only_checkpointing(model: GraphCast):
        model.set_checkpoint_model(False)
        model.set_checkpoint_encoder(False)
        model.set_checkpoint_processor(-1)
        model.set_checkpoint_decoder(False)

    @staticmethod
    def fine_grained_checkpointing(model: GraphCast):
        model.set_checkpoint_model(False)
        model.set_checkpoint_encoder(False)
        model.set_checkpoint_processor(0)
        model.set_checkpoint_decoder(False)
