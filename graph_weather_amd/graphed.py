"""The inference forward as ONE HIP graph.

``model(features)`` of the forecaster is ~50 kernel launches (two per message-passing block, per-sample chains on two HIP
streams, a handful of allocator calls) issued from Python.  In fp32 at 1 degree the GPU work (6.9 ms) hides that; with
split-operand products (``"bf16x3"``, 3.2 ms of GPU work at batch 2) the step is bound by the host issuing launches and varies
with the host (3.6 - 4.4 ms measured across boxes of one pool).  A HIP graph of the whole forward - all launches of every
stream, their fork / join events and the allocator's buffers frozen into one object - replays with a single host call.

Nothing is traced or compiled: the graph holds exactly the launches the eager forward issued (same kernels, same arguments),
captured by ``torch.cuda.graph`` from the streams the C-ABI calls were enqueued on.  What a captured graph cannot follow is a
change of what those launches point at: new weights are re-packed into new buffers, a new input shape changes every launch.
``ForwardGraph`` therefore keys the capture on the input shape, the version counters of every parameter, the matrix-product
dtype of every MLP and every attribute that shapes the launch sequence (streams of the mesh stack, deterministic sums,
checkpoint segments, efficient batching), and re-captures when the key changes (an optimizer step, ``load_state_dict``,
``set_compute_dtype``, ``set_deterministic`` ...).  NOT detectable, here as in the eager path's per-weight-version caches
(``layers._ver``): edits through ``p.data`` and in-place edits of inference tensors (neither moves a version counter), and
modules / parameters swapped into the model after the first call (the module walk is cached) - call ``invalidate()`` then.

Input: the graph reads a fixed address.  By default that is the graph's own buffer and each call copies the batch into it (a
device-to-device copy, 53 MB at 1 degree / batch 2: ~20 us).  When the caller hands over the SAME buffer call after call (a
rollout writing step t's output into step t + 1's input, a serving loop with a staging buffer) the graph is captured on - or
re-captured once onto - that buffer and replays with no copy at all; a different buffer later sends it back to its own.

``AutoGraph`` is how ``GraphWeatherForecaster.forward`` uses this by itself: in ``eval()`` under ``torch.no_grad()``, for small
inputs, from the third call of one shape on the eager forward is replaced by the replay (output cloned out of the graph's buffer, so
the module keeps the semantics of an ordinary call); anything the capture cannot do switches it off for the model, loudly.
"""
from __future__ import annotations

import gc
import os
import warnings
import weakref
from typing import Optional

import torch
from torch import nn

# attributes that change which launches a forward issues (besides weights, dtypes and the input shape)
_SHAPING_ATTRS = ("streams", "deterministic", "checkpoint_segments", "efficient_batching", "use_checkpointing")


class ForwardGraph:
    """``ForwardGraph(model)(features)`` == ``model(features)`` under ``torch.no_grad()``, replayed from a HIP graph.

    The returned tensor is the graph's own output buffer: it is overwritten by the next call (clone it to keep it, or pass
    ``clone=True``).  ``.input`` is the buffer the graph reads: write into it and call with no argument to skip the copy, or
    keep passing one and the same buffer (see the module docstring: ``pin_after`` consecutive calls with one address re-capture
    the graph on that buffer)."""

    def __init__(self, model: nn.Module, warmup: int = 3, pin_after: int = 3, weak: bool = False):
        # ``weak`` (AutoGraph): the model owns this object, so this object must not own the model - a reference cycle would
        # leave the captured HIP graph to the cyclic garbage collector, which may run in the middle of ANOTHER capture, where
        # destroying a graph is "not permitted when stream is capturing" and the exception leaves a destructor (abort)
        self._model_ref = weakref.ref(model) if weak else None
        self._model = None if weak else model
        self.warmup = int(warmup)
        self.pin_after = int(pin_after)
        self._key = None
        self._params = None
        self._mlps = None
        self._shaping = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self.input: Optional[torch.Tensor] = None
        self.output: Optional[torch.Tensor] = None
        self.captures = 0
        self.pinned = False       # the graph reads the caller's buffer (held alive here) instead of its own
        self._pin_ok = True       # one unpin (a caller that alternates buffers) ends the attempts
        self._last_ptr = None
        self._same_ptr = 0

    @property
    def model(self) -> nn.Module:
        m = self._model if self._model_ref is None else self._model_ref()
        if m is None:
            raise RuntimeError("graph_weather_amd: the model of this ForwardGraph no longer exists")
        return m

    def invalidate(self) -> None:
        """Forget the capture and the cached module walk: the next call re-captures (after edits no version counter shows:
        ``p.data``, inference tensors, swapped modules)."""
        self._graph = None
        self._key = None
        self._params = None
        self.pinned = False

    def _eager(self, x: torch.Tensor) -> torch.Tensor:
        fwd = getattr(self.model, "_forward_eager", None)  # (models whose forward() may itself replay a graph: AutoGraph)
        return fwd(x) if fwd is not None else self.model(x)

    def _state_key(self, shape, device, dtype) -> tuple:
        from .layers import MLP

        if self._params is None:  # (module structure is fixed after construction: walk it once; invalidate() re-walks)
            self._params = list(self.model.parameters())
            mods = list(self.model.modules())
            self._mlps = [m for m in mods if isinstance(m, MLP)]
            self._shaping = [(m, a) for m in mods for a in _SHAPING_ATTRS if hasattr(m, a)]
        versions = tuple((p.data_ptr(), -1 if p.is_inference() else p._version) for p in self._params)
        dtypes = tuple(str(m.compute_dtype) for m in self._mlps)
        shaping = tuple(repr(getattr(m, a)) for m, a in self._shaping)
        return (tuple(shape), str(device), str(dtype), versions, dtypes, shaping)

    def _capture(self, features: torch.Tensor, pin: bool) -> None:
        from . import ops

        if ops.TIMER is not None:
            raise RuntimeError("graph_weather_amd: a kernel timer is active (HIP events cannot be recorded into a graph)")
        dev = features.device
        self._graph = None  # (release the previous graph's memory pool before building the next)
        # Everything below runs with inference mode switched OFF locally (and no_grad on): torch.cuda.graph keeps the default
        # generator's graph-safe seed / offset tensors alive across captures and updates them IN PLACE at the start of every
        # later capture - created under a caller's torch.inference_mode() they would be inference tensors, the next capture outside
        # inference mode would fail half-way ("Inplace update to inference tensor ...") and leave the generator in its
        # capturing state, after which every torch.rand on the device raises.  The same goes for the graph's own buffers.
        # ... and with the cyclic garbage collector held off: garbage that owns HIP graphs, events or streams (somebody's dropped
        # model, a finished ForwardGraph) must be destroyed BEFORE the capture starts, never inside it (see __init__).
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self._capture_locked(features, pin, dev)
        finally:
            if gc_was_on:
                gc.enable()
        self.captures += 1

    def _capture_locked(self, features: torch.Tensor, pin: bool, dev) -> None:
        with torch.inference_mode(False), torch.no_grad():
            if pin:
                self.input = features  # the caller's buffer; the reference keeps its memory mapped for the graph's lifetime
            else:
                self.input = torch.empty_like(features)
                self.input.copy_(features)
            self.pinned = pin
            # warm-up on the CURRENT stream: packed weights, cached embeddings, graph plans, side streams and every per-device
            # kernel attribute exist before the capture starts (nothing of that may happen inside it), and every cache entry
            # belongs to the stream that will read it later (a side-stream warm-up left them owned by a stream nobody
            # synchronises with)
            for _ in range(max(1, self.warmup)):
                self._eager(self.input)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.output = self._eager(self.input)
        self._graph = g

    def __call__(self, features: Optional[torch.Tensor] = None, clone: bool = False, pin_now: bool = False) -> torch.Tensor:
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
        ptr = features.data_ptr()
        self._same_ptr = self._same_ptr + 1 if ptr == self._last_ptr else 1
        self._last_ptr = ptr
        if self._graph is None or key != self._key:
            # (pin_now: the caller has already seen this buffer on its previous calls - AutoGraph counts them while it runs eager)
            keep = self.pinned and self.input is not None and ptr == self.input.data_ptr()
            self._capture(features, pin=(keep or (pin_now and self._pin_ok)) and features.is_contiguous())
            self._key = key
        elif ptr != self.input.data_ptr():
            if self.pinned:  # a pinned graph reads somebody's buffer: never write into that - back to an own buffer, for good
                self._pin_ok = False
                self._capture(features, pin=False)
            elif self._pin_ok and self.pin_after > 0 and self._same_ptr >= self.pin_after and features.is_contiguous():
                self._capture(features, pin=True)  # the caller keeps handing over one buffer: read it in place
            else:
                self.input.copy_(features)
        self._graph.replay()
        return self.output.clone() if clone else self.output


class AutoGraph:
    """The policy behind ``GraphWeatherForecaster.forward`` in eval mode (see the module docstring): ``step(features)`` returns
    the forecast - eagerly for the first ``after`` calls of a shape, from the graph afterwards - or None when this call has to
    take the ordinary path (grad mode, training, large input, timer or foreign capture active, switched off)."""

    MAX_INPUT_BYTES = 128 << 20  # 1 degree up to batch 4: where the forward is launch-bound; larger inputs are GPU-bound and
    # the graph's private pool (every intermediate of the forward, held for the graph's lifetime) is not worth it

    def __init__(self, model: nn.Module, after: int = 2):
        self._model_ref = weakref.ref(model)  # (the model owns this object: no cycle, see ForwardGraph.__init__)
        self.after = int(after)
        self.enabled = os.environ.get("GW_AUTO_GRAPH", "1") != "0"
        self._shape = None
        self._seen = 0
        self._ptr = None
        self._ptr_same = 0  # consecutive calls that handed over the same buffer
        self._fg: Optional[ForwardGraph] = None

    def usable(self, features: torch.Tensor) -> bool:
        from . import ops

        model = self._model_ref()
        return (self.enabled and model is not None and not model.training and not torch.is_grad_enabled() and features.is_cuda
                and features.numel() * features.element_size() <= self.MAX_INPUT_BYTES and ops.TIMER is None
                and not torch.cuda.is_current_stream_capturing())

    def step(self, features: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.usable(features):
            return None
        shape = (tuple(features.shape), features.dtype, features.device)
        if shape != self._shape:
            self._shape, self._seen = shape, 0
        self._seen += 1
        ptr = features.data_ptr()
        self._ptr_same = self._ptr_same + 1 if ptr == self._ptr else 1
        self._ptr = ptr
        if self._seen <= self.after:
            return None
        if self._fg is None:
            self._fg = ForwardGraph(self._model_ref(), warmup=1, weak=True)
        elif self._fg._graph is not None and self._fg._state_key(features.shape, features.device, features.dtype) != self._fg._key:
            # weights / dtype / flags changed under the graph: drop it and count afresh - a loop that alternates weight updates
            # and evaluation forwards must not pay a capture (three forwards' worth) per call
            self._fg._graph = None
            self._seen = 1
            return None
        try:
            # a buffer that came back on every call so far (a rollout's input, a staging buffer, a benchmark loop) is read in
            # place from the first capture on; fresh tensors per call get the graph's own input buffer and a copy per call
            return self._fg(features, clone=True, pin_now=self._ptr_same > self.after)
        except Exception as exc:  # a forward the capture cannot follow: say so once, stay eager from here on
            self.enabled = False
            self._fg = None
            warnings.warn("graph_weather_amd: the automatic HIP graph of the inference forward was switched off for this model "
                          "(%s: %s); the eager path is used" % (type(exc).__name__, exc))
            torch.cuda.synchronize(features.device)
            return None


class AutoGraphModule:
    """Mixin of the models whose eval forward may replay itself (``GraphWeatherForecaster``, ``GraphCast``): the class flag, the
    per-instance policy object kept out of copies and pickles, and the one call ``forward`` makes before its eager path
    (``_forward_eager``, which ``ForwardGraph`` calls for warm-up and capture)."""

    # eval() + torch.no_grad() + a small input: from the third call of a shape on the forward is replayed from one HIP graph
    # (same launches, output cloned out of the graph's buffer).  ``model.auto_graph = False`` or GW_AUTO_GRAPH=0: every call eager.
    auto_graph = True

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_auto", None)  # a captured HIP graph is neither copied nor pickled with the module
        return state

    def _auto_graph_step(self, features: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.auto_graph:
            return None
        auto = self.__dict__.get("_auto")
        if auto is None or auto._model_ref() is not self:
            # (a shallow copy of the module - nn.DataParallel's replicas copy __dict__ - must not replay the ORIGINAL's graph)
            auto = self.__dict__["_auto"] = AutoGraph(self)
        if not (features.is_cuda and features.dtype == torch.float32 and not self.training and not torch.is_grad_enabled()):
            return None
        return auto.step(features)
