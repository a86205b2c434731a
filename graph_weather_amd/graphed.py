"""The inference forward as ONE HIP graph.

``model(features)`` of the forecaster is ~50 kernel launches (two per message-passing block, per-sample chains on two HIP
streams, a handful of allocator calls) issued from Python.  In fp32 at 1 degree the GPU work (7.2 ms) hides that; with
split-operand products (``"bf16x3"``, 3.3 ms of GPU work at batch 2) the step is bound by the host issuing launches and varies
with the host (3.6 - 4.4 ms measured across boxes of one pool).  A HIP graph of the whole forward - all launches of every
stream, their fork / join events and the allocator's buffers frozen into one object - replays with a single host call.

Nothing is traced or compiled: the graph holds exactly the launches the eager forward issued (same kernels, same arguments),
captured by ``torch.cuda.graph`` from the streams the C-ABI calls were enqueued on.  What a captured graph cannot follow is a
change of what those launches point at: new weights are re-packed into new buffers, a new input shape changes every launch.
``ForwardGraph`` therefore keys the capture on the input shape, the version counters of every parameter and the matrix-product
dtype of every MLP, and re-captures when the key changes (an optimizer step, ``load_state_dict``, ``set_compute_dtype``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class ForwardGraph:
    """``ForwardGraph(model)(features)`` == ``model(features)`` under ``torch.no_grad()``, replayed from a HIP graph.

    The returned tensor is the graph's own output buffer: it is overwritten by the next call (clone it to keep it), as the
    input is copied into the graph's own input buffer (a device-to-device copy of the batch, 53 MB at 1 degree / batch 2,
    enqueued in front of the replay - or write into ``.input`` directly and call with no argument)."""

    def __init__(self, model: nn.Module, warmup: int = 3):
        self.model = model
        self.warmup = int(warmup)
        self._key = None
        self._params = None
        self._mlps = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self.input: Optional[torch.Tensor] = None
        self.output: Optional[torch.Tensor] = None
        self.captures = 0

    def _state_key(self, shape, device, dtype) -> tuple:
        from .layers import MLP

        if self._params is None:  # (module structure is fixed after construction: walk it once)
            self._params = list(self.model.parameters())
            self._mlps = [m for m in self.model.modules() if isinstance(m, MLP)]
            self._dets = [m for m in self.model.modules() if hasattr(m, "deterministic")]
        versions = tuple((p.data_ptr(), -1 if p.is_inference() else p._version) for p in self._params)
        dtypes = tuple(str(m.compute_dtype) for m in self._mlps)
        det = tuple(bool(m.deterministic) for m in self._dets)
        return (tuple(shape), str(device), str(dtype), versions, dtypes, det)

    def _capture(self, features: torch.Tensor) -> None:
        from . import ops

        if ops.TIMER is not None:
            raise RuntimeError("graph_weather_amd: a kernel timer is active (HIP events cannot be recorded into a graph)")
        dev = features.device
        self._graph = None  # (release the previous graph's memory pool before building the next)
        self.input = torch.empty_like(features)
        self.input.copy_(features)
        # warm-up on a side stream: packed weights, cached embeddings, graph plans, side streams and every per-device kernel
        # attribute exist before the capture starts (nothing of that may happen inside it)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(max(1, self.warmup)):
                self.model(self.input)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            self.output = self.model(self.input)
        self._graph = g
        self.captures += 1

    def __call__(self, features: Optional[torch.Tensor] = None) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters()) and self.model.training:
            raise RuntimeError("graph_weather_amd: ForwardGraph replays the inference forward - call it on model.eval() "
                               "(training steps go through autograd, which a static graph cannot follow)")
        if features is None:
            if self._graph is None:
                raise RuntimeError("graph_weather_amd: ForwardGraph needs an input on its first call")
            features = self.input
        if not features.is_cuda:
            raise RuntimeError("graph_weather_amd: features must be on a HIP device - there is no CPU path")
        key = self._state_key(features.shape, features.device, features.dtype)
        if self._graph is None or key != self._key:
            self._capture(features)
            self._key = key
        elif features is not self.input:
            self.input.copy_(features)
        self._graph.replay()
        return self.output
