"""Host-side mirror of the reference's module tree for the forecaster hot path.

Same class names, constructor arguments, attribute names and ``state_dict`` keys as
``graph_weather/models/layers/{graph_net_block,encoder,processor,decoder,assimilator_decoder}.py`` so reference
checkpoints load with ``strict=True`` - but ``forward`` enqueues the hand-written HIP kernels of
``libgw_amd.so`` (through ``graph_weather_amd.ops``) instead of ATen / torch_scatter / PyG ops.

Batching uses *shared-graph* semantics everywhere (one edge list for all batch elements; node tables
``[B, N, D]``): output-equivalent to the reference's replicated graph (its own efficient_batching tests,
``tests/models/layers/test_efficient_batching.py:53,91,145``) without building B copies of the graph,
re-encoding the batch-independent edge features B times, or running node MLPs on rows that are then dropped.

Under ``torch.no_grad()`` the inference kernels run; with autograd on, the same calls go through ``autograd.py`` (fp32).
Tensors must be fp32 on a HIP device.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from . import autograd as ag
from . import ops
from . import routes
from . import wide
from .autograd import OperandSpec
from .graphs import ForecastGraphs, GraphPlan, build_forecast_graphs
from .ops import BF16X3, Operand, PackedMLP

_NORMS = ["LayerNorm", "GraphNorm", "InstanceNorm", "BatchNorm", "MessageNorm"]


def _ver(t: torch.Tensor) -> int:
    """Version counter of a tensor for cache keys.  Tensors created under ``torch.inference_mode()`` carry no counter
    (reading ``_version`` raises): they get -1, i.e. the cache then trusts the tensor's identity (the caches hold the tensor
    object itself, so its address cannot be recycled) - an in-place change of an inference tensor between two calls is not
    detectable by anybody, autograd included."""
    return -1 if t.is_inference() else t._version


def _version_key(params) -> tuple:
    return tuple((p.data_ptr(), _ver(p), p.device) for p in params)


_DEVICE_STREAMS = {}


def device_stream(device, kind: str, index: int = 0) -> "torch.cuda.Stream":
    """The process-wide side stream ``(kind, index)`` of a device ("mesh" 0, 1, ...: per-sample chains of the mesh stack;
    "prefetch": cold rebuild of the decoder tables).  One small fixed set per device, shared by every model of the process: a
    stream per module instance (a bench run builds a dozen models) made the HIP runtime hand out dozens of streams, and which
    hardware queue the two chains of a stack landed on - i.e. whether they overlapped at all - came to depend on how many
    streams had been created before them (round 6: the batch-8 fp32 forward 24.9 ms or 27.3 ms on the same tree)."""
    key = (str(torch.device(device)), kind, int(index))
    st = _DEVICE_STREAMS.get(key)
    if st is None:
        st = _DEVICE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class KeyedCache:
    """The per-module caches of batch-independent tensors (edge / mesh embeddings, their layer-1 products, tile forms, graph
    plans): ``get(name, key, make, hold=None)`` returns the entry made for ``key`` or makes it.  Keys are built from
    ``_version_key`` (address + version counter of every parameter involved) and/or the address + version of an input tensor;
    an entry whose key contains a tensor's ADDRESS also ``hold``s that tensor - while the entry lives the address cannot be
    recycled for another tensor, and a different object at the same address is a miss.  One class for what used to be four
    hand-rolled tuples (``_cache``, ``_e0_cache``, ``_e0_seg_cache``, ``_plan_cache``)."""

    def __init__(self):
        self._entries = {}

    def get(self, name: str, key, make, hold=None):
        hit = self._entries.get(name)
        if hit is None or hit[0] != key or hit[2] is not hold:
            hit = (key, make(), hold)
            self._entries[name] = hit
        return hit[1]

    def fresh(self, name: str, key, hold=None) -> bool:
        hit = self._entries.get(name)
        return hit is not None and hit[0] == key and hit[2] is hold

    def clear(self) -> None:
        self._entries.clear()

    def __contains__(self, name: str) -> bool:
        return name in self._entries


def _autograd_on(module: nn.Module, *inputs: Optional[torch.Tensor]) -> bool:
    """True when the call must be differentiable: grad mode on and either the module has trainable parameters or one of the
    given input tensors requires grad (a frozen block downstream of a trainable one must still pass gradients through)."""
    on = torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters())
                                      or any(t is not None and t.requires_grad for t in inputs))
    if on:
        for m in module.modules():
            if isinstance(m, MLP) and m.compute_dtype == torch.bfloat16:
                raise NotImplementedError("graph_weather_amd: the backward pass is implemented for float32 and bf16x3 matrix "
                                          "products; call the bfloat16 mode under torch.no_grad() (inference)")
    return on


def _form(mlp: "MLP") -> routes.MlpForm:
    """routes.MlpForm of an MLP.  Only the bf16 routes read the packed shape facts: the other modes are told apart by dtype
    alone, so their weights are not packed just to answer a route question (host-only callers, CPU tests)."""
    if mlp.compute_dtype == torch.bfloat16:
        return routes.MlpForm.of(mlp)
    return routes.MlpForm(mlp.compute_dtype, -1, -1, False)


class Feed:
    """One input of a fused op: a row table, how many of its rows belong to one batch element (0 = shared by the
    batch) and how it enters layer 1: "raw" rows (matrix pass), "proj" rows already multiplied by their layer-1
    weight slice (gather-add), or "zero" (slice skipped)."""

    def __init__(self, tensor: Optional[torch.Tensor], rows_pb: int, mode: str):
        self.tensor, self.rows_pb, self.mode = tensor, rows_pb, mode

    def operand(self) -> Operand:
        if self.mode == "zero":
            return ops.ZERO
        if self.mode == "tiles":  # edge features as bf16 edge tiles (bf16 mode); rows_pb 0 = one tile set shared by the batch
            return Operand(self.tensor, self.rows_pb, 256, tiles=True)
        return Operand(self.tensor, self.rows_pb, 256, projected=(self.mode == "proj"))

    def spec(self) -> OperandSpec:
        return OperandSpec(self.mode, self.rows_pb)


FEED_ZERO = Feed(None, 0, "zero")


def set_deterministic(module: nn.Module, flag: bool = True) -> nn.Module:
    """Bitwise run-to-run reproducible forward (inference): the segment sums of every message-passing block under ``module``
    use per-tile carry records added in tile order instead of fp32 atomics, whose order is not fixed (include/gw_amd.h:
    GW_EDGE_DETERMINISTIC).  The reference's ``scatter_add_`` is order-nondeterministic on a GPU as well - this is an extra,
    at a small cost (the carry records, one more small launch per edge update; the bf16 path uses its 4-wave kernel)."""
    blocks = [m for m in module.modules() if isinstance(m, GraphNetBlock)]
    if flag:
        for m in blocks:
            mlp = m.edge_model.edge_mlp
            if mlp.compute_dtype == torch.float32 and (mlp._layout()[4] or mlp._norm() is None or len(mlp._linears()) != 3):
                # carry records exist in the hand-scheduled fp32 edge kernel (native 256 widths, LayerNorm, two hidden layers) and in
                # the bf16 kernels; zero-padded narrow blocks, norm_type=None and deeper MLPs run on the general fp32 kernel
                raise NotImplementedError("graph_weather_amd: deterministic segment sums are implemented for message-passing blocks "
                                          "of the native width (256) with LayerNorm and hidden_layers = 2 (float32), or bfloat16 "
                                          "matrix products; this block is %d -> %d wide with norm %s"
                                          % (mlp.hidden_dim, mlp.out_dim, "LayerNorm" if mlp._norm() is not None else "None"))
    for m in blocks:
        m.deterministic = bool(flag)
    return module


def set_compute_dtype(module: nn.Module, dtype) -> nn.Module:
    """Select the matrix-product dtype of every MLP under ``module``: ``torch.float32`` (default, the reference's
    arithmetic), ``torch.bfloat16`` (bf16 MFMA with fp32 accumulation; parameters, activations in HBM, LayerNorm,
    residuals and segment sums stay fp32) - what a reference user gets from ``torch.autocast(dtype=torch.bfloat16)`` - or
    ``"bf16x3"`` (``ops.BF16X3``): every product as three bf16 MFMAs on hi / lo operand pairs (16 significant bits per operand,
    fp32 accumulate, csrc/gw_split.hip) - the reference's fp32 outputs to a few 1e-5 of their scale, i.e. inside BASELINE.json's
    1e-3, at several times the fp32 matrix rate; every tensor in HBM stays fp32 rows.  bfloat16 is inference only; bf16x3 also
    trains (mixed precision: forward and the backward's input-gradient products on split operands, activation saves, weight-
    gradient GEMMs, LayerNorm / ReLU backward, master weights and the optimizer in fp32 - autograd.py)."""
    if isinstance(dtype, str):
        if dtype.lower() not in (BF16X3, "split"):
            raise RuntimeError("graph_weather_amd: compute dtype must be torch.float32, torch.bfloat16 or \"bf16x3\"")
        dtype = BF16X3
    elif dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("graph_weather_amd: compute dtype must be torch.float32, torch.bfloat16 or \"bf16x3\"")
    for m in module.modules():
        if isinstance(m, MLP):
            m.compute_dtype = dtype
        if isinstance(getattr(m, "_cache", None), KeyedCache):
            m._cache.clear()
    return module


class MLP(nn.Module):
    """``MLP`` - graph_net_block.py:17-77.  ``model`` is the same nn.Sequential layout (Linear/ReLU.../[LayerNorm]).

    The HIP kernels come in three shapes: hidden 256 -> 256 outputs (+ LayerNorm), and output heads of up to 80 features
    behind 128 or 256 hidden units.  An MLP of any other width up to 256 (the reference's own defaults are 128,
    graph_net_block.py:20-28,234-244) runs on those kernels zero-padded: padded hidden units are relu(0) = 0, padded
    outputs are 0, LayerNorm statistics span the real ``out_dim`` features only (``gw_mlp_weights.ln_width``), and an MLP
    with a single hidden layer gets an identity middle layer (relu(I h) = h for h >= 0) - all exact.  MLPs whose output
    feeds a 256-wide node / edge table (``as_table``) keep the padded layout between kernels; ``forward`` slices it off."""

    def __init__(self, in_dim: int, out_dim: int = 128, hidden_dim: int = 128, hidden_layers: int = 2,
                 norm_type: Optional[str] = "LayerNorm", use_checkpointing: bool = False):
        super().__init__()
        # graph_net_block.py:73-74: under autograd, keep no activations of this MLP and recompute them in the backward
        # (autograd.recompute); inference keeps nothing anyway
        self.use_checkpointing = use_checkpointing
        layers: List[nn.Module] = [nn.Linear(in_dim, hidden_dim), nn.ReLU()]
        for _ in range(hidden_layers - 1):
            layers += [nn.Linear(hidden_dim, hidden_dim), nn.ReLU()]
        layers.append(nn.Linear(hidden_dim, out_dim))
        if norm_type is not None:
            assert norm_type in _NORMS
            norm_layer = getattr(nn, norm_type)  # only LayerNorm exists in torch.nn, as in the reference
            layers.append(norm_layer(out_dim))
        self.model = nn.Sequential(*layers)
        self.in_dim, self.out_dim, self.hidden_dim = in_dim, out_dim, hidden_dim
        self._packed: Optional[PackedMLP] = None
        self._packed_key = None
        self._splits: Tuple[Tuple[int, int], ...] = ((0, in_dim),)
        self._table = False
        # dtype of the matrix products: float32 (fp32 MFMA, bitwise an fmaf chain) or bfloat16 (bf16 MFMA, fp32
        # accumulate); parameters and activations stay fp32 either way.  Set through set_compute_dtype().
        self.compute_dtype = torch.float32

    def set_input_splits(self, splits) -> None:
        """Column slices of layer 1 fed by separate operands (``cat`` order of the reference)."""
        self._splits = tuple(splits)
        self._packed = None

    def as_table(self) -> "MLP":
        """The output of this MLP is a node / edge table consumed by the message-passing kernels (256 floats per row)."""
        self._table = True
        self._packed = None
        return self

    def _linears(self):
        return [m for m in self.model if isinstance(m, nn.Linear)]

    def _norm(self):
        return self.model[-1] if isinstance(self.model[-1], nn.LayerNorm) else None

    # ---- mapping onto a kernel variant -----------------------------------------------------------------------------
    def _layout(self):
        """(hidden units, outputs, layer-1 slice widths, identity layer inserted?, any padding?) of the kernel variant."""
        hid, out = self.hidden_dim, self.out_dim
        if hid > 256 or out > 256:
            raise NotImplementedError("graph_weather_amd: the HIP kernels are built for MLP widths up to 256")
        if self._table or out > 80:
            H, O = 256, 256
        else:
            H, O = (128 if hid <= 128 else 256), out
        single = len(self._splits) == 1
        kp = []
        for lo, hi in self._splits:
            k = hi - lo
            if k > 256:
                raise NotImplementedError("graph_weather_amd: MLP inputs wider than 256 features are not implemented")
            kp.append(k if (single and O == 256 and k <= 112) else 256)
        identity = len(self._linears()) == 2
        padded = identity or hid != H or O != out or any(k != hi - lo for k, (lo, hi) in zip(kp, self._splits))
        return H, O, tuple(kp), identity, padded

    def native_k(self) -> int:
        return self._layout()[2][0]

    def native_out(self) -> int:
        return self._layout()[1]

    def native_splits(self) -> Tuple[Tuple[int, int], ...]:
        kp = self._layout()[2]
        offs = [0]
        for k in kp:
            offs.append(offs[-1] + k)
        return tuple((offs[i], offs[i + 1]) for i in range(len(kp)))

    def native_key(self) -> tuple:
        return _version_key(list(self.model.parameters()))

    def native_params(self) -> List[torch.Tensor]:
        """Parameters in the shapes the kernels and the backward products work on, in ``model.parameters()`` order.  With
        kernel-native widths these are the parameters themselves; otherwise zero-padded copies made with differentiable
        torch ops (pad / cat = data movement), so autograd hands the slices of their gradients back to the parameters."""
        H, O, kp, identity, padded = self._layout()
        if not padded:
            return list(self.model.parameters())
        import torch.nn.functional as F

        lin, norm = self._linears(), self._norm()
        hid, out = self.hidden_dim, self.out_dim
        parts = [F.pad(lin[0].weight[:, lo:hi], (0, k - (hi - lo))) for (lo, hi), k in zip(self._splits, kp)]
        ps = [F.pad(torch.cat(parts, dim=1), (0, 0, 0, H - hid)), F.pad(lin[0].bias, (0, H - hid))]
        for m in lin[1:-1]:
            ps += [F.pad(m.weight, (0, H - hid, 0, H - hid)), F.pad(m.bias, (0, H - hid))]
        if identity:
            w = lin[0].weight
            ps += [torch.eye(H, dtype=w.dtype, device=w.device), torch.zeros(H, dtype=w.dtype, device=w.device)]
        ps += [F.pad(lin[-1].weight, (0, H - hid, 0, O - out)), F.pad(lin[-1].bias, (0, O - out))]
        if norm is not None:
            ps += [F.pad(norm.weight, (0, O - out)), F.pad(norm.bias, (0, O - out))]
        return ps

    def packed(self) -> PackedMLP:
        norm = self._norm()
        key = (self.native_key(), self.compute_dtype)
        if self._packed is None or key != self._packed_key:
            if not self.model[0].weight.is_cuda:
                raise RuntimeError("graph_weather_amd: module parameters must be on a HIP device (no CPU path exists)")
            if norm is not None and abs(norm.eps - 1e-5) > 0:
                raise RuntimeError("graph_weather_amd: LayerNorm eps must be 1e-5")
            with torch.no_grad():
                ps = [p.detach() for p in self.native_params()]
            n_lin = (len(ps) - (2 if norm is not None else 0)) // 2
            self._packed = PackedMLP([ps[2 * i] for i in range(n_lin)], [ps[2 * i + 1] for i in range(n_lin)],
                                     (ps[-2], ps[-1]) if norm is not None else None, self.native_splits(), self.compute_dtype,
                                     ln_width=self.out_dim if norm is not None else 0)
            self._packed_key = key
        return self._packed

    def run(self, x2: torch.Tensor, n_rows: int, rows_per_batch: int, residual: Optional[Operand] = None) -> torch.Tensor:
        """Rows through the kernel variant: [n_rows, in_dim] -> [n_rows, native_out()] (the padded table layout for
        narrow ``as_table`` MLPs); differentiable when grad mode is on."""
        if wide.is_wide(self):  # widths above 256: layer by layer on the generic kernels (wide.py), unpadded rows
            return wide.mlp_rows(self, x2, None if residual is None else residual.tensor[:, :self.out_dim])
        k = self.native_k()
        if k > x2.shape[1]:
            x2 = torch.nn.functional.pad(x2, (0, k - x2.shape[1]))  # inputs of 113..255 features: zero columns
        if residual is not None and self.native_out() != self.out_dim:
            raise NotImplementedError("graph_weather_amd: a fused residual needs an output head of at most 80 features")
        if _autograd_on(self, x2, None if residual is None else residual.tensor):
            if self.use_checkpointing:  # graph_net_block.py:73-74: keep nothing, recompute this MLP in the backward
                if residual is None:
                    return ag.recompute(lambda x_: (ag.mlp_rows(self, x_, n_rows, rows_per_batch),), (x2,), self)[0]
                return ag.recompute(lambda x_, r_: (ag.mlp_rows(self, x_, n_rows, rows_per_batch,
                                                               residual_op=Operand(r_, residual.rows_per_batch, residual.k)),),
                                    (x2, residual.tensor), self)[0]
            return ag.mlp_rows(self, x2, n_rows, rows_per_batch, residual_op=residual)
        return ops.mlp_forward(self.packed(), Operand(x2, rows_per_batch, k), n_rows, rows_per_batch, residual=residual)

    def table(self, x: torch.Tensor) -> torch.Tensor:
        """``forward`` without slicing the padding off: [rows, in_dim] -> [rows, 256] for ``as_table`` MLPs."""
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        n = int(x2.shape[0])
        return self.run(x2, n, max(n, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """graph_net_block.py:63-77."""
        lead = x.shape[:-1]
        if wide.is_wide(self):
            return wide.mlp_rows(self, x.reshape(-1, x.shape[-1]).contiguous()).reshape(*lead, self.out_dim)
        y = self.table(x)
        if y.shape[1] != self.out_dim:
            y = y[:, :self.out_dim]
        return y.reshape(*lead, self.out_dim)


class EdgeProcessor(nn.Module):
    """``EdgeProcessor`` - graph_net_block.py:87-137 (parameter container; arithmetic in gw_edge_update_forward)."""

    def __init__(self, in_dim_node=128, in_dim_edge=128, hidden_dim=128, hidden_layers=2, norm_type="LayerNorm"):
        super().__init__()
        self.edge_mlp = MLP(2 * in_dim_node + in_dim_edge, in_dim_edge, hidden_dim, hidden_layers, norm_type).as_table()
        self.edge_mlp.set_input_splits(((0, in_dim_node), (in_dim_node, 2 * in_dim_node),
                                        (2 * in_dim_node, 2 * in_dim_node + in_dim_edge)))


class NodeProcessor(nn.Module):
    """``NodeProcessor`` - graph_net_block.py:140-193."""

    def __init__(self, in_dim_node=128, in_dim_edge=128, hidden_dim=128, hidden_layers=2, norm_type="LayerNorm"):
        super().__init__()
        self.node_mlp = MLP(in_dim_node + in_dim_edge, in_dim_node, hidden_dim, hidden_layers, norm_type).as_table()
        self.node_mlp.set_input_splits(((0, in_dim_node), (in_dim_node, in_dim_node + in_dim_edge)))


class GraphNetBlock(nn.Module):
    """Stands where the reference puts PyG ``MetaLayer`` (graph_net_block.py:221-228): attributes
    ``edge_model`` / ``node_model`` give the same state_dict keys."""

    def __init__(self, edge_model: EdgeProcessor, node_model: NodeProcessor):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model
        self.deterministic = False  # set_deterministic(): reproducible segment sums

    def run(self, batch: int, plan: GraphPlan, x_src: Feed, x_dst: Feed, e_in: Feed, e_res: torch.Tensor, e_res_rows_pb: int,
            x_node: Feed, x_res: Optional[torch.Tensor], x_res_rows_pb: int, want_edges: bool, device,
            tag: Optional[str] = None, agg_zeroed: Optional[torch.Tensor] = None, post_w=None, post_zero: bool = False,
            post_half: bool = False, head=None, seg=None, agg_acc: Optional[torch.Tensor] = None):
        """One message-passing block on a shared graph: e' (optional), x' for all ``batch * plan.n_dst`` rows.
        Inputs may be raw rows, rows pre-multiplied by their layer-1 weight slice, or zeros (see ``Feed``).
        ``agg_zeroed``: an aggregate buffer the caller has already had zero-filled (by the projection launch).
        ``post_w`` (inference): packed layer-1 slices of the NEXT block's edge MLP - the node update multiplies the new rows by
        them in the same launch (and zero-fills the next aggregate with ``post_zero``); returns (x', e', products, next_agg).
        ``post_half``: those products as fp16 rows (bf16 mode, when their consumer is the bf16 edge update with resident weights:
        the products are gathered once per incident edge, so their bytes dominate what the edge update reads).
        ``seg`` (bf16 inference, no residual, no e'): the plan's segment-aligned tiles (``GraphPlan.seg_tiles()``) - the edge update
        runs on the padded edge list (batch-shared per-edge operands must be in padded order, ``SegTiles.pad_rows``), writes the
        aggregate with plain stores as bf16 rows in the K order the node update's matrix product reads (a quarter of the bytes
        of the fp32 round trip; no zero fill when every destination has an edge).  ``seg`` with ``agg_acc`` (a processor block): the
        edge update adds its segment sums onto ``agg_acc`` in place - the previous block's aggregate (include/gw_amd.h:
        GW_EDGE_SEGMENT_TILES); edge tiles then cover the padded list, and no next aggregate is zero-filled."""
        n_dst, n_edges = plan.n_dst, plan.num_edges
        if _autograd_on(self, x_src.tensor, x_dst.tensor, e_in.tensor, e_res, x_node.tensor, x_res):
            agg, e_out = ag.edge_update(self.edge_model.edge_mlp, plan, batch, (x_src.spec(), x_dst.spec(), e_in.spec()),
                                        want_edges, x_src.tensor, x_dst.tensor, e_in.tensor, e_res, e_res_rows_pb)
            x_new = ag.node_update(self.node_model.node_mlp, batch * n_dst, n_dst, x_node.spec(), x_node.tensor, x_res,
                                   x_res_rows_pb, agg)
            return x_new, e_out
        if seg is not None and agg_acc is not None:
            if want_edges not in (False, "tiles"):
                raise RuntimeError("graph_weather_amd: a processor block on segment-aligned tiles hands e' over as edge tiles")
            agg = agg_acc
            n_edges = seg.n_pad  # (the tile buffers cover the padded list)
        elif seg is not None:
            if want_edges or e_res is not None:
                raise RuntimeError("graph_weather_amd: segment-aligned tiles come without residual and without e'")
            if seg.split:  # pieces of long runs meet in fp32 atomics
                agg = agg_zeroed if agg_zeroed is not None else torch.zeros((batch * n_dst, 256), dtype=torch.float32, device=device)
            else:
                agg = (torch.empty if seg.complete else torch.zeros)((batch * n_dst, 256), dtype=torch.bfloat16, device=device)
        else:
            agg = agg_zeroed if agg_zeroed is not None else torch.zeros((batch * n_dst, 256), dtype=torch.float32, device=device)
        if want_edges == "tiles":  # e' stays in the kernels' own bf16 tile format for the next block
            e_out = torch.empty(ops.edge_tiles_bytes(batch, n_edges), dtype=torch.uint8, device=device)
        else:
            e_out = torch.empty((batch * n_edges, 256), dtype=torch.float32, device=device) if want_edges else None
        res_op = ops.ZERO if e_res is None else Operand(e_res, e_res_rows_pb, 256, tiles=(e_res.dtype == torch.uint8))
        ops.edge_update_forward(self.edge_model.edge_mlp.packed(), batch, plan.src if seg is None else seg.src,
                                plan.dst if seg is None else seg.dst, x_src.operand(), x_dst.operand(),
                                e_in.operand(), res_op, n_dst, agg, e_out, tag=tag, deterministic=self.deterministic,
                                segment_tiles=seg is not None, segment_split=seg is not None and seg.split and agg_acc is None)
        res_x = ops.ZERO if x_res is None else Operand(x_res, x_res_rows_pb, 256)
        if head is not None:
            # (bf16 inference, decoder) the node update and the output head that follows it in one launch: ``head`` = (packed
            # head MLP, residual operand or None); returns the head's output instead of the new node rows
            if x_res is not None:
                raise RuntimeError("graph_weather_amd: the fused node update + head has no node residual (decoder rows are zeros)")
            y = ops.node_update_head_forward(self.node_model.node_mlp.packed(), head[0], batch * n_dst, n_dst, x_node.operand(),
                                             Operand(agg, n_dst, 256), head[1])
            return y, e_out
        if post_w is not None:
            next_agg = torch.empty((batch * n_dst, 256), dtype=torch.float32, device=device) if post_zero else None
            if seg is not None and agg_acc is not None and not post_zero:
                next_agg = None
            x_new, posts = ops.node_update_forward(self.node_model.node_mlp.packed(), batch * n_dst, n_dst, x_node.operand(), res_x,
                                                   Operand(agg, n_dst, 256), post_w=post_w, zero_rows=next_agg, post_half=post_half)
            return x_new, e_out, posts, next_agg
        x_new = ops.node_update_forward(self.node_model.node_mlp.packed(), batch * n_dst, n_dst, x_node.operand(), res_x,
                                        Operand(agg, n_dst, 256))
        return x_new, e_out

    def params_key(self) -> tuple:
        return _version_key(list(self.parameters()))


def build_graph_processor_block(in_dim_node=128, in_dim_edge=128, hidden_dim_node=128, hidden_dim_edge=128,
                                hidden_layers_node=2, hidden_layers_edge=2, norm_type="LayerNorm") -> nn.Module:
    """graph_net_block.py:196-228."""
    return GraphNetBlock(
        edge_model=EdgeProcessor(in_dim_node, in_dim_edge, hidden_dim_edge, hidden_layers_edge, norm_type),
        node_model=NodeProcessor(in_dim_node, in_dim_edge, hidden_dim_node, hidden_layers_node, norm_type),
    )


def _check_native_dims(in_dim_node, in_dim_edge, hidden_dim_node, hidden_dim_edge, norm_type):
    if max(in_dim_node, in_dim_edge, hidden_dim_node, hidden_dim_edge) > 256 or norm_type not in ("LayerNorm", None):
        raise NotImplementedError(
            "graph_weather_amd: the fused HIP message-passing kernels handle node / edge / hidden widths up to 256 (narrower "
            "models run zero-padded to 256) with LayerNorm or no norm (the only norm_type values torch.nn resolves, "
            "graph_net_block.py:50-59).  Wider models run on the generic kernels of graph_weather_amd/wide.py - this call was "
            "reached with a table the fused kernels cannot take")


def _pad256(t: torch.Tensor) -> torch.Tensor:
    """A [rows, width <= 256] tensor in the 256-float table layout of the kernels (zero columns appended)."""
    t = t.contiguous()
    if t.shape[1] > 256:
        raise RuntimeError("graph_weather_amd: a %d-wide table reached the 256-wide kernels" % int(t.shape[1]))
    return t if t.shape[1] == 256 else torch.nn.functional.pad(t, (0, 256 - t.shape[1]))


class GraphProcessor(nn.Module):
    """``GraphProcessor`` - graph_net_block.py:231-301."""

    def __init__(self, mp_iterations=15, in_dim_node=128, in_dim_edge=128, hidden_dim_node=128, hidden_dim_edge=128,
                 hidden_layers_node=2, hidden_layers_edge=2, norm_type="LayerNorm", use_checkpointing=False):
        super().__init__()
        self.use_checkpointing = use_checkpointing
        self._dims = (in_dim_node, in_dim_edge, hidden_dim_node, hidden_dim_edge, norm_type)
        self.blocks = nn.ModuleList()
        for _ in range(mp_iterations):
            self.blocks.append(build_graph_processor_block(in_dim_node, in_dim_edge, hidden_dim_node, hidden_dim_edge,
                                                           hidden_layers_node, hidden_layers_edge, norm_type))
        self.checkpoint_segments = 0  # processor.py:70-81, set through Processor.set_checkpoint_segments
        self.streams = 0  # HIP streams of the fused inference forward: 0 = automatic (see forward_streams), 1 = one stream
        self._cache = KeyedCache()  # "plan" (user COO graph), "e0_pe" / "e0_tiles" / "e0_seg" (block 0's batch-shared edge features)

    # -- native path: shared dst-sorted plan, node table [batch*n, 256], edge features in sorted order ----------
    def run_plan(self, x: torch.Tensor, plan: GraphPlan, e: torch.Tensor, e_shared: bool, batch: int,
                 want_edges: bool, pre_proj=None, tail_w=None, tail_half: bool = False):
        """Layer 1 of every edge MLP is split (cat[x_s, x_d, e].W1^T = x_s.Ws^T + x_d.Wd^T + e.We^T): the node
        products are computed once per node (shared by its ~7 incident edges) and gathered per edge; when the
        incoming edge features are batch independent (first block after the encoder) their product is cached.

        Under autograd, ``use_checkpointing`` (graph_net_block.py:294-297: one checkpoint per block) and
        ``checkpoint_segments`` (processor.py:70-81: -1 = the whole stack, N > 0 = every N blocks) select recomputation:
        the segment's forward runs the inference kernels and keeps only its inputs; its backward re-runs it with the
        activation saves (autograd.recompute).

        Inference: the layer-1 node products of block i + 1 are made by the node update of block i (``GraphNetBlock.run``
        ``post_w``); ``pre_proj`` = (P_s, P_d, zeroed aggregate) of the first block when the caller's previous launch made
        them, ``tail_w`` = packed slices to multiply the final node rows with (returns (x, e, products) then)."""
        if wide.processor_is_wide(self):
            x, e_out = wide.run_blocks(self, x, plan, e, e_shared, batch, want_edges)
            return (x, e_out) if tail_w is None else (x, e_out, None)
        _check_native_dims(*self._dims)
        nb = len(self.blocks)
        seg = 0
        if _autograd_on(self, x, e):
            if self.checkpoint_segments == -1:
                seg = nb
            elif self.checkpoint_segments > 0:
                seg = int(self.checkpoint_segments)
            elif self.use_checkpointing:
                seg = 1
        if seg <= 0 or nb == 0:
            x, e_cur, _, tail = self._run_blocks(0, nb, x, e, e_shared, plan, batch, want_edges, pre_proj, tail_w, tail_half)
            if tail_w is not None:
                return x, (e_cur if want_edges else None), tail
            return x, (e_cur if want_edges else None)
        e_cur, shared = e, e_shared
        for lo in range(0, nb, seg):
            hi = min(nb, lo + seg)
            need_e = want_edges or hi < nb

            def fn(x_, e_, lo=lo, hi=hi, shared=shared, need_e=need_e):
                xo, eo, _, _ = self._run_blocks(lo, hi, x_, e_, shared, plan, batch, need_e)
                return (xo, eo) if need_e else (xo,)

            outs = ag.recompute(fn, (x, e_cur), nn.ModuleList(list(self.blocks)[lo:hi]))
            x = outs[0]
            if need_e:
                e_cur, shared = outs[1], False
        if tail_w is not None:  # (checkpointed training path: the caller's projection stays separate)
            return x, (e_cur if want_edges else None), None
        return x, (e_cur if want_edges else None)

    def forward_streams(self, batch: int) -> int:
        """Streams the fused inference forward runs this stack on: ``self.streams`` if set, else 2 for fp32 / bf16x3 matrix products
        and batch >= 2 (their mesh-sized launches of 64-column workgroups leave workgroup slots idle in the last round; the bf16
        kernels are persistent and occupy every CU by themselves), else 1."""
        return routes.mesh_streams(self.streams, [b.edge_model.edge_mlp.compute_dtype for b in self.blocks], batch)

    def side_streams(self, device, n: int):
        return [device_stream(device, "mesh", i) for i in range(n)]

    def prepare_shared(self, e: torch.Tensor, plan: GraphPlan) -> None:
        """Everything the per-sample chains share, made on the current stream: packed weights of every block and the cached
        product / tile form of the batch-independent edge features of the first block."""
        for blk in self.blocks:
            blk.edge_model.edge_mlp.packed()
            blk.node_model.node_mlp.packed()
        if len(self.blocks):
            # (on the stream of the caller, before the per-sample chains fork: whichever of the two cached forms block 0 reads)
            seg = self._seg_for(plan)
            if seg is not None:
                self._shared_e0_seg(self.blocks[0], e, seg)
            else:
                self._shared_e0(self.blocks[0], e, plan.num_edges)

    def _seg_for(self, plan: GraphPlan):
        """The plan's segment-aligned tiles when the whole stack can run on them (bf16 inference with the resident kernels in
        every block, atomics mode, at most 16 destinations per tile: csrc/gw_edge16p.hip), else None."""
        if len(self.blocks) == 0 or plan.num_edges == 0:
            return None
        if any(b.edge_model.edge_mlp.compute_dtype != torch.bfloat16 for b in self.blocks):  # (no packing of other modes' weights)
            return None
        forms = [(_form(b.edge_model.edge_mlp), b.node_model.node_mlp.compute_dtype, bool(b.deterministic)) for b in self.blocks]
        if not routes.stack_on_segment_tiles(forms, plan.num_edges, 0):  # (everything but the tiling itself, which is built lazily)
            return None
        seg = plan.seg_tiles()
        return seg if routes.stack_on_segment_tiles(forms, plan.num_edges, None if seg is None else seg.max_slots) else None

    def _shared_e0_seg(self, blk, e_cur: torch.Tensor, seg):
        """(We . e in padded order, e as one shared set of bf16 edge tiles over the padded list) of batch-independent edge
        features on segment-aligned tiles, cached per (e, weights)."""
        key = (e_cur.data_ptr(), _ver(e_cur), blk.params_key(), seg.n_pad)

        def make():
            pe = self._shared_e0(blk, e_cur, int(e_cur.shape[0]), tiles=False)[1]
            return seg.pad_rows(pe), ops.edge_rows_to_tiles(seg.pad_rows(e_cur), 1, seg.n_pad, seg.n_pad)

        return self._cache.get("e0_seg", key, make, hold=e_cur)

    def _shared_e0(self, blk, e_cur: torch.Tensor, n_edges: int, tiles: bool = True):
        """(key, We . e, e, [e as one shared set of bf16 edge tiles]) of batch-independent edge features, cached per (e, weights).
        ``tiles=False``: the caller reads only the product (the segment-tile route keeps its own padded tile set)."""
        mlp_e = blk.edge_model.edge_mlp
        pm_e = mlp_e.packed()
        key = (e_cur.data_ptr(), _ver(e_cur), blk.params_key())
        pe = self._cache.get("e0_pe", key, lambda: ops.project_forward([pm_e.w1[2]], Operand(e_cur, n_edges, 256), n_edges, n_edges)[0],
                             hold=e_cur)
        out = (key, pe, e_cur)
        if tiles and routes.resident_bf16(routes.MlpForm.of(mlp_e, pm_e), n_edges):
            # the residual of the resident bf16 kernel: e as one shared tile set
            out = out + (self._cache.get("e0_tiles", key, lambda: ops.edge_rows_to_tiles(e_cur, 1, n_edges, n_edges), hold=e_cur),)
        return out

    def _run_blocks(self, lo: int, hi: int, x: torch.Tensor, e: torch.Tensor, e_shared: bool, plan: GraphPlan, batch: int,
                    want_edges: bool, pre_proj=None, tail_w=None, tail_half: bool = False):
        """Blocks [lo, hi) of the stack; returns (x, e, e_shared, tail products) after them (e of the last block only if
        ``want_edges``)."""
        n, n_edges = plan.n_dst, plan.num_edges
        e_cur, shared = e, e_shared
        train = _autograd_on(self, x, e)
        carried = None if train else pre_proj  # (P_s, P_d, zeroed aggregate) made by the previous node update
        tail = None
        # segment-aligned tiles: the running aggregate of that route starts from the cached segment sums of the BATCH-SHARED edge
        # features of block 0 (csrc/gw_edge16p.hip: agg += sum(LN(.)) on top of sum(e)); per-sample edge features handed in by a
        # caller (Processor.forward without efficient batching) take the tile route without segment alignment
        seg = self._seg_for(plan) if routes.segment_route_allowed(train, bool(want_edges), lo, shared) else None

        def route_of(j: int) -> Optional[str]:  # the edge-update route of block j of this call (None past the stack)
            if j >= len(self.blocks):
                return None
            m = self.blocks[j].edge_model.edge_mlp
            return routes.EDGE_AUTOGRAD if train else routes.edge_route(_form(m), n_edges, False)
        for i in range(lo, hi):
            blk = self.blocks[i]
            last = i == hi - 1
            mlp_e = blk.edge_model.edge_mlp
            agg_buf = None
            pm_e = None
            if train:
                ps, pd = ag.project(mlp_e, (0, 1), x, batch * n, n)
            else:
                pm_e = mlp_e.packed()
                if carried is not None:
                    ps, pd, agg_buf = carried
                    carried = None
                else:
                    if mlp_e.compute_dtype == torch.float32:  # the projection launch also zero-fills this block's aggregate
                        agg_buf = torch.empty((batch * n, 256), dtype=torch.float32, device=x.device)
                    ps, pd = ops.project_forward([pm_e.w1[0], pm_e.w1[1]], Operand(x, n, 256), batch * n, n, zero_rows=agg_buf)
            # bf16 mode (inference): between blocks the per-sample edge features live as bf16 "edge tiles" - the MFMA B-operand
            # order the next block's layer-1 product consumes directly (csrc/gw_edge16.hip); only what crosses the API is rows
            route = route_of(i)
            tiled = route == routes.EDGE_TILES_BF16
            if shared:
                if train:
                    pe = ag.project(mlp_e, (2,), e_cur, n_edges, n_edges)[0]
                elif seg is not None:
                    pe = self._shared_e0_seg(blk, e_cur, seg)[0]
                else:
                    pe = self._shared_e0(blk, e_cur, n_edges)[1]
                e_in = Feed(pe, 0, "proj")
            elif tiled:
                if e_cur.dtype != torch.uint8:  # per-sample rows handed over by a caller: into the tile format once
                    e_cur = ops.edge_rows_to_tiles(e_cur.contiguous(), batch, n_edges, n_edges)
                e_in = Feed(e_cur, n_edges, "tiles")
            else:
                e_in = Feed(e_cur, n_edges, "raw")
            need_e = want_edges or not last
            # e' stays in the bf16 tile format only when the block that consumes it reads tiles too (same compute dtype and
            # shape: set_compute_dtype applied to a sub-module can leave neighbours in different modes)
            out_kind = routes.edge_out_kind(route, bool(need_e), route_of(i + 1) if not last else None)
            if seg is not None and shared:
                e_res = self._shared_e0_seg(blk, e_cur, seg)[1]
            elif shared and tiled:
                e_res = self._shared_e0(blk, e_cur, n_edges)[3]
            else:
                e_res = e_cur
            # what the node update of this block also produces (inference): the next block's layer-1 node products
            post_w, post_zero, post_half = None, False, False
            if not train:
                pm_n = blk.node_model.node_mlp.packed()
                if i + 1 < len(self.blocks):
                    nxt_mlp = self.blocks[i + 1].edge_model.edge_mlp
                    nxt = nxt_mlp.packed()
                    if nxt.weight_dtype == pm_n.weight_dtype and i + 1 < hi:
                        # (segment-aligned stack: the next block adds onto this block's aggregate - nothing to zero-fill)
                        post_w, post_zero = [nxt.w1[0], nxt.w1[1]], seg is None
                        # block i + 1 >= 1 reads per-sample edge tiles, i.e. runs its layer 1 in the bf16 layer-1 kernel, which
                        # gathers these products once per edge: hand them over as fp16 rows
                        post_half = routes.post_products_half(route_of(i + 1))
                elif tail_w is not None and all(w_.dtype == pm_n.w_out.dtype for w_ in tail_w):
                    post_w = list(tail_w)
                    post_half = bool(tail_half)
            if seg is not None and agg_buf is None:
                agg_buf = torch.zeros((batch * n, 256), dtype=torch.float32, device=x.device)
            res = blk.run(batch, plan, Feed(ps, n, "proj"), Feed(pd, n, "proj"), e_in, e_res, 0 if shared else n_edges,
                          Feed(x, n, "raw"), x, n, out_kind, x.device, tag="processor_edge", agg_zeroed=agg_buf,
                          post_w=post_w, post_zero=post_zero, post_half=post_half, seg=seg, agg_acc=agg_buf if seg is not None else None)
            x, e_new = res[0], res[1]
            if post_w is not None:
                if post_zero:
                    carried = (res[2][0], res[2][1], res[3])
                elif seg is not None and i + 1 < hi:
                    carried = (res[2][0], res[2][1], agg_buf)  # the running aggregate goes on to the next block
                else:
                    tail = res[2]
            if e_new is not None:
                e_cur, shared = e_new, False
        return x, e_cur, shared, tail

    def _plan_for(self, edge_index: torch.Tensor, num_nodes: int) -> GraphPlan:
        key = (edge_index.data_ptr(), tuple(edge_index.shape), _ver(edge_index), num_nodes)

        def make():
            if edge_index.dim() != 2 or edge_index.shape[0] != 2:
                raise RuntimeError("edge_index must be [2, E] in COO format")
            if edge_index.numel() and (int(edge_index.min()) < 0 or int(edge_index.max()) >= num_nodes):
                raise IndexError("edge_index refers to nodes outside x")
            dst_sorted, perm = torch.sort(edge_index[1], stable=True)
            return GraphPlan(num_nodes, num_nodes, edge_index[0][perm].to(torch.int32).contiguous(),
                             dst_sorted.to(torch.int32).contiguous(), perm, None)

        return self._cache.get("plan", key, make, hold=edge_index)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """graph_net_block.py:279-301 for an arbitrary COO graph (the reference's random-graph test,
        tests/models/test_gradient_checkpointing.py:62-86): edges are dst-sorted once per edge_index tensor."""
        plan = self._plan_for(edge_index, int(x.shape[0]))
        dn, de = int(x.shape[1]), int(edge_attr.shape[1])
        if wide.processor_is_wide(self):
            x_out, e_out = wide.run_blocks(self, x.contiguous(), plan, edge_attr[plan.perm].contiguous(), False, 1, True)
            e_ref = torch.empty_like(e_out)
            e_ref[plan.perm] = e_out
            return x_out, e_ref
        e_sorted = _pad256(edge_attr)[plan.perm].contiguous()
        x_out, e_out = self.run_plan(_pad256(x), plan, e_sorted, False, 1, True)
        e_ref = torch.empty_like(e_out)
        e_ref[plan.perm] = e_out
        return (x_out if dn == 256 else x_out[:, :dn]), (e_ref if de == 256 else e_ref[:, :de])


class Encoder(nn.Module):
    """``Encoder`` - encoder.py:36-268."""

    def __init__(self, lat_lons: list, resolution: int = 2, input_dim: int = 78, output_dim: int = 256,
                 output_edge_dim: int = 256, hidden_dim_processor_node=256, hidden_dim_processor_edge=256,
                 hidden_layers_processor_node=2, hidden_layers_processor_edge=2, mlp_norm_type="LayerNorm",
                 use_checkpointing: bool = False, efficient_batching: bool = False,
                 _graphs: Optional[ForecastGraphs] = None):
        super().__init__()
        self.use_checkpointing = use_checkpointing
        self.efficient_batching = efficient_batching
        self.output_dim = output_dim
        self.num_latlons = len(lat_lons)
        self.graphs = _graphs if _graphs is not None else build_forecast_graphs(lat_lons, resolution)
        self.num_h3 = self.graphs.num_mesh
        # encoder.py:112-114 - zero-initialised learnable mesh-node inputs
        self.h3_nodes = nn.Parameter(torch.zeros((self.num_h3, input_dim), dtype=torch.float))
        self.output_edge_dim = output_edge_dim
        self.node_encoder = MLP(input_dim, output_dim, hidden_dim_processor_node, hidden_layers_processor_node,
                                mlp_norm_type, use_checkpointing).as_table()
        self.edge_encoder = MLP(2, output_edge_dim, hidden_dim_processor_edge, hidden_layers_processor_edge,
                                mlp_norm_type, use_checkpointing).as_table()
        self.latent_edge_encoder = MLP(2, output_edge_dim, hidden_dim_processor_edge, hidden_layers_processor_edge,
                                       mlp_norm_type, use_checkpointing).as_table()
        self.graph_processor = GraphProcessor(1, output_dim, output_edge_dim, hidden_dim_processor_node,
                                              hidden_dim_processor_edge, hidden_layers_processor_node,
                                              hidden_layers_processor_edge, mlp_norm_type, use_checkpointing)
        self._dev_plans = {}
        self._cache = KeyedCache()

    # plans live outside state_dict (like encoder.graph / encoder.latent_graph, encoder.py:107,109)
    def _plans(self, device):
        key = str(device)
        if key not in self._dev_plans:
            self._dev_plans[key] = (self.graphs.enc_plan.to(device), self.graphs.lat_plan.to(device))
        return self._dev_plans[key]

    def _cached(self, name: str, params, fn):
        if _autograd_on(self):  # training: batch-independent embeddings are part of the graph, recomputed every step
            return fn()
        return self._cache.get(name, _version_key(params), fn)

    def mesh_embedding(self) -> torch.Tensor:
        """node_encoder(h3_nodes): batch independent (encoder.py:199-205 recomputes it for every sample)."""
        ps = list(self.node_encoder.parameters()) + [self.h3_nodes]
        return self._cached("mesh", ps, lambda: self.node_encoder.table(self.h3_nodes if _autograd_on(self) else self.h3_nodes.detach()))

    def encoder_edge_embedding(self, plan: GraphPlan) -> torch.Tensor:
        return self._cached("enc_e", list(self.edge_encoder.parameters()), lambda: self.edge_encoder.table(plan.edge_attr))

    def latent_edge_embedding(self, plan: GraphPlan) -> torch.Tensor:
        """latent_edge_encoder(attr) once on [E_lat, 2] in dst-sorted order (encoder.py:235-241 repeats B times)."""
        return self._cached("lat_e", list(self.latent_edge_encoder.parameters()),
                            lambda: self.latent_edge_encoder.table(plan.edge_attr))

    def encode(self, features: torch.Tensor, post_w=None):
        """encoder.py:199-223 -> mesh node features [(B*M), D] (batch-major, reversed-rank mesh order).  ``post_w`` (inference,
        fused forward): packed layer-1 slices of the first processor block - returns (x, products, zeroed aggregate)."""
        if features.dim() != 3 or features.shape[1] != self.num_latlons:
            raise RuntimeError("features must be [B, %d, input_dim]" % self.num_latlons)
        B, G, F = (int(s) for s in features.shape)
        if wide.encoder_is_wide(self):
            return wide.encode(self, features)
        feats = features.contiguous().reshape(B * G, F)
        enc_plan, _ = self._plans(features.device)
        team = self.team_path(features)
        x3 = self.split_path(features)
        ne = self.node_encoder
        fused_ps = None
        if ((team and ne.compute_dtype == torch.bfloat16) or (x3 and ne.compute_dtype == BF16X3)) and not ne._layout()[4] \
                and 32 < ne.native_k() <= 128:
            # node encoder and the x[row] product of the edge MLP's layer 1 in ONE launch: the grid rows themselves are never
            # written (the encoder drops them, encoder.py:219-223) - only Ws . node_encoder(features) (bf16 mode: as fp16 rows)
            pm_e0 = self.graph_processor.blocks[0].edge_model.edge_mlp.packed()
            k = ne.native_k()
            x2 = feats if k <= F else torch.nn.functional.pad(feats, (0, k - F))
            fused_ps = ops.mlp_post_forward(ne.packed(), Operand(x2, G, k), B * G, G, [pm_e0.w1[0]], post_half=team)[1][0]
            xg = None
        else:
            xg = ne.run(feats, B * G, G)  # grid rows only
        xm = self.mesh_embedding()
        e = self.encoder_edge_embedding(enc_plan)
        blk = self.graph_processor.blocks[0]
        _check_native_dims(*self.graph_processor._dims)
        pd_xm, pe, px_xm, e = self._static_projections(blk, xm, e)
        x_src, e_res = Feed(xg, G, "raw"), e
        seg = None
        if team or x3:
            # 16-bit inference (bf16: the team-pipelined edge kernel, csrc/gw_edge16t.hip; bf16x3: csrc/gw_split.hip): every operand
            # of the edge MLP's layer 1 enters as a product - Ws.xg is made once per grid node here (one edge per grid node: the same
            # matrix work the raw operand would cost inside the edge kernel; bf16 mode: handed over as fp16 rows) - and, e' being
            # dropped (encoder.py:219) and e batch independent, the residual leaves the kernel: sum(LN(.) + e) = sum(LN(.)) + S, and
            # Wa.S joins the cached node-update product of the mesh rows.
            pm_e = blk.edge_model.edge_mlp.packed()
            if fused_ps is None:
                fused_ps = ops.project_forward([pm_e.w1[0]], Operand(xg, G, 256), B * G, G, out_half=team)[0]
            x_src = Feed(fused_ps, G, "proj")
            px_xm = self._team_node_product(blk, enc_plan, e, px_xm)
            e_res = None
        if team:
            seg = enc_plan.seg_tiles(split=True)  # (a polar mesh cell collects hundreds of grid nodes: runs split over tiles)
            if seg is not None:
                pe = self._cached("enc_pe_pad", list(self.parameters()), lambda: seg.pad_rows(pe))
        if post_w is not None:
            x, _, posts, agg0 = blk.run(B, enc_plan, x_src, Feed(pd_xm, 0, "proj"), Feed(pe, 0, "proj"), e_res, 0,
                                        Feed(px_xm, 0, "proj"), xm, 0, False, features.device, tag="encoder_edge",
                                        post_w=post_w, post_zero=True, seg=seg)
            return x, posts, agg0
        x, _ = blk.run(B, enc_plan, x_src, Feed(pd_xm, 0, "proj"), Feed(pe, 0, "proj"), e_res, 0,
                       Feed(px_xm, 0, "proj"), xm, 0, False, features.device, tag="encoder_edge", seg=seg)
        return x

    def team_path(self, features: Optional[torch.Tensor] = None) -> bool:
        """Inference in bf16 with everything the team-pipelined edge kernel needs (see ``AssimilatorDecoder.team_path``).
        ``features``: the call's input - an input that requires grad keeps the call on the differentiable path (which raises
        for bf16: no silent drop of d/d(features) through the non-differentiable fused launches)."""
        return self._block_route(features) == routes.BLOCK_TEAM

    def _block_route(self, features: Optional[torch.Tensor] = None) -> str:
        """routes.block_route of the encoder's bipartite block for this call."""
        blk = self.graph_processor.blocks[0]
        mlp_e, mlp_n = blk.edge_model.edge_mlp, blk.node_model.node_mlp
        is_wide = wide.encoder_is_wide(self)
        auto = False if is_wide else _autograd_on(self, features)
        if is_wide or auto:
            return routes.BLOCK_ROWS
        return routes.block_route(_form(mlp_e), mlp_n.compute_dtype, self.graphs.enc_plan.num_edges, False, False,
                                  bool(blk.deterministic))

    def split_path(self, features: Optional[torch.Tensor] = None) -> bool:
        """Inference with bf16x3 (split-operand) products in both MLPs of the encoder block: the edge update then runs with every
        layer-1 operand projected and without residual, like the bf16 team path, on fp32 rows (csrc/gw_split.hip)."""
        return self._block_route(features) == routes.BLOCK_SPLIT

    def _team_node_product(self, blk, plan: GraphPlan, e: torch.Tensor, px_xm: torch.Tensor) -> torch.Tensor:
        """Wx.xm + Wa.S with S[dst] = the sum of the (batch-independent) edge features e over the destination's edges: the
        node-update layer-1 operand of the mesh rows when the edge kernel adds no residual.  Cached per weight version."""
        def make():
            pm_n = blk.node_model.node_mlp.packed()
            n_e = int(e.shape[0])
            e_sum = ag.segment_sum_rows(e, n_e, 1, 1, plan.n_dst, plan.dst_ptr(), None)
            pa = ops.project_forward([pm_n.w1[1]], Operand(e_sum, plan.n_dst, 256), plan.n_dst, plan.n_dst)[0]
            return wide.add_rows(px_xm, pa)

        return self._cached("enc_proj_team", list(self.parameters()), make)

    def _static_projections(self, blk, xm: torch.Tensor, e: torch.Tensor):
        """Batch-independent layer-1 products of the encoder block (mesh rows are the same for every sample,
        encoder.py:199-204): Wd.xm (edge MLP), We.e (edge MLP), Wx.xm (node MLP) - cached per weight version; and e itself (under
        autograd: as an output of the node that made We.e)."""
        ps = list(self.parameters())

        def make():
            M, G = int(xm.shape[0]), int(e.shape[0])
            if _autograd_on(self):
                pd_xm = ag.project(blk.edge_model.edge_mlp, (1,), xm, M, M)[0]
                # (e comes back as an output of the same autograd node: it is also the block's residual, and the two gradients of
                # the [E, 256] table then leave that node as one - autograd.ProjectFunction)
                pe, e_same = ag.project(blk.edge_model.edge_mlp, (2,), e, G, G, passthrough=True)
                px_xm = ag.project(blk.node_model.node_mlp, (0,), xm, M, M)[0]
                return pd_xm, pe, px_xm, e_same
            pm_e = blk.edge_model.edge_mlp.packed()
            pm_n = blk.node_model.node_mlp.packed()
            pd_xm = ops.project_forward([pm_e.w1[1]], Operand(xm, M, 256), M, M)[0]
            pe = ops.project_forward([pm_e.w1[2]], Operand(e, G, 256), G, G)[0]
            px_xm = ops.project_forward([pm_n.w1[0]], Operand(xm, M, 256), M, M)[0]
            return pd_xm, pe, px_xm, e

        return self._cached("enc_proj", ps, make)

    def forward(self, features: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """encoder.py:153-242.  Returns reference-order tensors: replicated graph by default, the single shared
        graph with ``efficient_batching`` (encoder.py:190-196)."""
        B = int(features.shape[0])
        x = self.encode(features)
        _, lat_plan = self._plans(features.device)
        e_sorted = self.latent_edge_embedding(lat_plan)
        e_ref = torch.empty_like(e_sorted)
        e_ref[lat_plan.perm] = e_sorted
        if x.shape[1] != self.output_dim:
            x = x[:, :self.output_dim]
        if e_ref.shape[1] != self.output_edge_dim:
            e_ref = e_ref[:, :self.output_edge_dim]
        ei = self.graphs.lat_edge_index.to(features.device)
        if self.efficient_batching:
            return x, ei, e_ref
        M = self.num_h3
        ei_rep = torch.cat([ei + i * M for i in range(B)], dim=1)  # max(edge_index)+1 == M (self loops), encoder.py:229
        return x, ei_rep, e_ref.repeat(B, 1)


class Processor(nn.Module):
    """``Processor`` - processor.py:17-128 (thermalizer not part of the hot path)."""

    def __init__(self, input_dim: int = 256, edge_dim: int = 256, num_blocks: int = 9, hidden_dim_processor_node: int = 256,
                 hidden_dim_processor_edge: int = 256, hidden_layers_processor_node: int = 2,
                 hidden_layers_processor_edge: int = 2, mlp_norm_type: str = "LayerNorm", use_thermalizer: bool = False,
                 use_checkpointing: bool = False):
        super().__init__()
        if use_thermalizer:
            raise NotImplementedError("the thermalizer (reference default off, forecast.py:83) is outside the hot path")
        self.input_dim = input_dim
        self.use_thermalizer = use_thermalizer
        self.checkpoint_segments = 0
        self.graph_processor = GraphProcessor(num_blocks, input_dim, edge_dim, hidden_dim_processor_node,
                                              hidden_dim_processor_edge, hidden_layers_processor_node,
                                              hidden_layers_processor_edge, mlp_norm_type, use_checkpointing)

    def set_checkpoint_segments(self, checkpoint_segments: int):
        """processor.py:70-81: 0 = per-block checkpointing as configured by ``use_checkpointing``; -1 = the whole processor
        is one recomputed segment; N > 0 = one segment per N blocks (the reference documents N > 0 as not yet
        implemented - here it is).  Takes effect under autograd (``GraphProcessor.run_plan``)."""
        self.checkpoint_segments = checkpoint_segments
        self.graph_processor.checkpoint_segments = checkpoint_segments

    def forward(self, x: torch.Tensor, edge_index, edge_attr, t: int = 0, batch_size: int = None,
                efficient_batching: bool = False) -> torch.Tensor:
        """processor.py:83-128."""
        dn = int(x.shape[1])
        if wide.processor_is_wide(self.graph_processor):
            x, edge_attr = x.contiguous(), edge_attr.contiguous()
        else:
            x, edge_attr = _pad256(x), _pad256(edge_attr)
        if efficient_batching and batch_size is not None and batch_size > 1:
            n = int(x.shape[0]) // batch_size
            plan = self.graph_processor._plan_for(edge_index, n)
            e_sorted = edge_attr[plan.perm].contiguous()
            out, _ = self.graph_processor.run_plan(x, plan, e_sorted, True, batch_size, False)
        else:
            plan = self.graph_processor._plan_for(edge_index, int(x.shape[0]))
            e_sorted = edge_attr[plan.perm].contiguous()
            out, _ = self.graph_processor.run_plan(x, plan, e_sorted, False, 1, False)
        return out if out.shape[1] == dn else out[:, :dn]


class AssimilatorDecoder(nn.Module):
    """``AssimilatorDecoder`` - assimilator_decoder.py:26-200."""

    def __init__(self, lat_lons: list, resolution: int = 2, input_dim: int = 256, output_dim: int = 78,
                 output_edge_dim: int = 256, hidden_dim_processor_node: int = 256, hidden_dim_processor_edge: int = 256,
                 hidden_layers_processor_node: int = 2, hidden_layers_processor_edge: int = 2,
                 mlp_norm_type: str = "LayerNorm", hidden_dim_decoder: int = 128, hidden_layers_decoder: int = 2,
                 use_checkpointing: bool = False, efficient_batching: bool = False,
                 _graphs: Optional[ForecastGraphs] = None):
        super().__init__()
        self.use_checkpointing = use_checkpointing
        self.efficient_batching = efficient_batching
        self.num_latlons = len(lat_lons)
        self.graphs = _graphs if _graphs is not None else build_forecast_graphs(lat_lons, resolution)
        self.num_h3 = self.graphs.num_mesh
        self.output_dim = output_dim
        # assimilator_decoder.py:108-110: hidden_layers of the edge encoder is the literal 2
        self.edge_encoder = MLP(2, output_edge_dim, hidden_dim_processor_edge, 2, mlp_norm_type, use_checkpointing).as_table()
        self.graph_processor = GraphProcessor(mp_iterations=1, in_dim_node=input_dim, in_dim_edge=output_edge_dim,
                                              hidden_dim_node=hidden_dim_processor_node,
                                              hidden_dim_edge=hidden_dim_processor_edge,
                                              hidden_layers_node=hidden_layers_processor_node,
                                              hidden_layers_edge=hidden_layers_processor_edge, norm_type=mlp_norm_type,
                                              use_checkpointing=use_checkpointing)
        self.node_decoder = MLP(input_dim, output_dim, hidden_dim_decoder, hidden_layers_decoder, None, use_checkpointing)
        self._dev_plans = {}
        self._cache = KeyedCache()

    def _plan(self, device) -> GraphPlan:
        key = str(device)
        if key not in self._dev_plans:
            self._dev_plans[key] = self.graphs.dec_plan.to(device)
        return self._dev_plans[key]

    def edge_embedding(self, plan: GraphPlan) -> torch.Tensor:
        if _autograd_on(self):
            return self.edge_encoder.table(plan.edge_attr)
        return self._cache.get("dec_e", _version_key(list(self.edge_encoder.parameters())), lambda: self.edge_encoder.table(plan.edge_attr))

    def prefetch_tables(self, dev) -> Optional["torch.cuda.Stream"]:
        """Cold forward (weights changed since the tables were made), inference, fp32 / bf16x3: the decoder's two big batch-
        independent tables - the edge embedding ``edge_encoder(attr)`` on its ~7 G rows and its layer-1 product ``We . e``
        (assimilator_decoder.py:175-177 recomputes the embedding on every forward) - are rebuilt on a SIDE stream while encoder and
        processor run on the caller's: they are first read by the decoder's edge update, ~4 ms later.  Returns that stream (the
        caller joins it in front of ``decode``) or None when everything is fresh / the path does not apply.  Measured
        (scripts/probes/cold_parts_probe.py): the decoder's share of the cold step's extra time is 1.7 of 3.6 ms in fp32."""
        if wide.decoder_is_wide(self) or _autograd_on(self) or len(self.graph_processor.blocks) == 0:
            return None
        blk = self.graph_processor.blocks[0]
        mlp_e = blk.edge_model.edge_mlp
        if mlp_e.compute_dtype == torch.bfloat16 or self.edge_encoder.compute_dtype == torch.bfloat16:
            return None  # (the frozen bf16 mode keeps its own table formats and is left as it is)
        key_e = _version_key(list(self.edge_encoder.parameters()))
        key = _version_key(list(self.edge_encoder.parameters()) + list(blk.parameters()))
        if self._cache.fresh("dec_e", key_e) and self._cache.fresh("dec_pe", key):
            return None
        plan = self._plan(dev)
        n_e = plan.num_edges
        if n_e == 0:
            return None
        main = torch.cuda.current_stream(dev)
        side = device_stream(dev, "prefetch")
        side.wait_stream(main)  # (ordered behind everything that may still read the tables - and packed weights - being replaced)
        with torch.cuda.stream(side):
            e = self.edge_embedding(plan)
            pm_e = mlp_e.packed()
            pe = self._cache.get("dec_pe", key, lambda: ops.project_forward([pm_e.w1[2]], Operand(e, n_e, 256), n_e, n_e)[0])
        for t in (e, pe):
            t.record_stream(main)  # made on the side stream, read on the caller's for as long as the cache entry lives
        return side

    def team_path(self) -> bool:
        """Inference in bf16 with everything the team-pipelined edge kernel needs (csrc/gw_edge16t.hip): one middle layer,
        LayerNorm over all 256 features, atomics mode, node and edge MLP of the block in the same dtype.  Then the decoder's edge
        update runs without residual (the sums of e enter the node update) and takes its layer-1 node products as fp16 rows."""
        return self._block_route() == routes.BLOCK_TEAM

    def _block_route(self) -> str:
        """routes.block_route of the decoder's bipartite block."""
        blk = self.graph_processor.blocks[0]
        mlp_e, mlp_n = blk.edge_model.edge_mlp, blk.node_model.node_mlp
        is_wide = wide.decoder_is_wide(self)
        auto = False if is_wide else _autograd_on(self)
        if is_wide or auto:
            return routes.BLOCK_ROWS
        return routes.block_route(_form(mlp_e), mlp_n.compute_dtype, self.graphs.dec_plan.num_edges, False, False,
                                  bool(blk.deterministic))

    def split_path(self) -> bool:
        """Inference with bf16x3 (split-operand) products in both MLPs of the decoder block (see ``Encoder.split_path``)."""
        return self._block_route() == routes.BLOCK_SPLIT

    def decode(self, processor_features: torch.Tensor, batch_size: int,
               residual: Optional[torch.Tensor] = None, ps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """assimilator_decoder.py:173-200 (+ decoder.py:93 when ``residual`` [B*G, ld] is given).  ``ps`` (inference, fused
        forward): the layer-1 product of the mesh rows with this decoder's edge MLP, made by the processor's last launch."""
        B, G, M = batch_size, self.num_latlons, self.num_h3
        if processor_features.shape[0] != B * M:
            raise RuntimeError("processor_features must have batch*num_h3 rows")
        dev = processor_features.device
        if wide.decoder_is_wide(self):
            return wide.decode(self, processor_features, B, residual=residual)
        processor_features = _pad256(processor_features)
        plan = self._plan(dev)
        e = self.edge_embedding(plan)
        blk = self.graph_processor.blocks[0]
        _check_native_dims(*self.graph_processor._dims)
        # lat/lon rows are zeros (assimilator_decoder.py:84,190-192): x_dst = 0, node input [0 | agg], residual 0.
        # Layer 1 of the edge MLP is then relu(Ws.x[src] + (We.e + b)): a gather-add of a per-mesh-node product and a
        # cached batch-independent per-edge product - no matrix work per edge in layer 1.
        train = _autograd_on(self, processor_features)
        mlp_e = blk.edge_model.edge_mlp
        n_e = plan.num_edges
        if train:
            ps = ag.project(mlp_e, (0,), processor_features.contiguous(), B * M, M)[0]
            pe, e = ag.project(mlp_e, (2,), e, n_e, n_e, passthrough=True)  # (e is the block's residual too: see Encoder._static_projections)
        else:
            pm_e = mlp_e.packed()
            team = self.team_path()
            if ps is None:
                ps = ops.project_forward([pm_e.w1[0]], Operand(processor_features.contiguous(), M, 256), B * M, M, out_half=team)[0]
            elif ps.dtype == torch.float16 and not team:
                raise RuntimeError("graph_weather_amd: fp16 layer-1 products were handed to a decoder that cannot take them")
            key = _version_key(list(self.edge_encoder.parameters()) + list(blk.parameters()))
            e_rows = e
            pe = self._cache.get("dec_pe", key, lambda: ops.project_forward([pm_e.w1[2]], Operand(e_rows, n_e, 256), n_e, n_e)[0])
            seg = plan.seg_tiles() if team else None
            if seg is not None:  # the per-edge product in the padded order of the segment-aligned tiles
                pe_rows = pe
                pe = self._cache.get("dec_pe_pad", key, lambda: seg.pad_rows(pe_rows))
            x_node = FEED_ZERO
            x3 = self.split_path()
            if x3 or routes.resident_bf16(_form(mlp_e), n_e):
                pm_n = blk.node_model.node_mlp.packed()
                if not (team or x3):
                    # residual of the resident bf16 kernel: the cached edge embedding as one shared set of bf16 edge tiles
                    e = self._cache.get("dec_e_tiles", key, lambda: ops.edge_rows_to_tiles(e_rows, 1, n_e, n_e))
                else:
                    # e' itself is dropped (assimilator_decoder.py:195) and e is the same for every sample, so
                    #   agg = sum(LN(.) + e) = sum(LN(.)) + S,  S[dst] = sum of e over the destination's edges (batch independent),
                    # and layer 1 of the node update (graph_net_block.py:189, x == 0) is Wa.agg = Wa.sum(LN(.)) + Wa.S: the edge
                    # kernel adds no residual at all, and Wa.S enters the node update as a cached, batch-shared PROJECTED operand
                    # in the slot of the all-zero x rows.
                    def make_e_sum():
                        e_sum = ag.segment_sum_rows(e_rows, n_e, 1, 1, plan.n_dst, plan.dst_ptr(), None)
                        # (bf16 mode: fp16 rows, like every layer-1 product of that path: the node update adds them to its fp32
                        # accumulator and rounds the sum to bf16; bf16x3: fp32 rows)
                        return ops.project_forward([pm_n.w1[1]], Operand(e_sum, plan.n_dst, 256), plan.n_dst, plan.n_dst, out_half=team)[0]

                    x_node = Feed(self._cache.get("dec_e_sum", key, make_e_sum), 0, "proj")
                    e = None
        res = None
        if residual is not None:
            if residual.dim() != 2 or residual.shape[0] != B * G or residual.shape[1] < self.output_dim:
                raise RuntimeError("graph_weather_amd: the residual (start features) must have batch*num_latlons rows of at least "
                                   "output_dim = %d features, got %s" % (self.output_dim, tuple(residual.shape)))
            res = Operand(residual, G, self.output_dim)
        # node update + node_decoder (+ residual) in one launch wherever the head has the fused kernel's shape (256 -> 128 -> 128 ->
        # <= 80 features, no norm, same matrix-product dtype as the node MLP): the [B*G, 256] grid-row table between them - 133 MB
        # at 1 degree, batch 2 - is never written or read (fp32: round 6; bf16 / bf16x3: round 3 / 5)
        head = None
        if not train:
            nd = self.node_decoder
            if nd.compute_dtype == blk.node_model.node_mlp.compute_dtype and not wide.is_wide(nd) and not nd._layout()[4]:
                pm_h = nd.packed()
                pm_nn = blk.node_model.node_mlp.packed()
                if (pm_h.hidden == 128 and pm_h.n_mid == 1 and pm_h.n_out <= 80 and pm_h.gamma is None and pm_nn.n_mid == 1
                        and pm_nn.gamma is not None and pm_nn.ln_width == 0):
                    head = (pm_h, res)
        if not train and x_node is not FEED_ZERO:
            out, _ = blk.run(B, plan, Feed(ps, M, "proj"), FEED_ZERO, Feed(pe, 0, "proj"), None, 0, x_node, None, 0, False, dev,
                             tag="decoder_edge", head=head, seg=seg)
            if head is not None:
                return out.reshape(B, G, self.output_dim)
            xg = out
        elif head is not None and mlp_e.compute_dtype == torch.float32:
            out, _ = blk.run(B, plan, Feed(ps, M, "proj"), FEED_ZERO, Feed(pe, 0, "proj"), e, 0, FEED_ZERO, None, 0, False, dev,
                             tag="decoder_edge", head=head)
            return out.reshape(B, G, self.output_dim)
        else:
            xg, _ = blk.run(B, plan, Feed(ps, M, "proj"), FEED_ZERO, Feed(pe, 0, "proj"), e, 0, FEED_ZERO, None, 0, False, dev,
                            tag="decoder_edge")
        y = self.node_decoder.run(xg, B * G, G, residual=res)
        if y.shape[1] != self.output_dim:
            y = y[:, :self.output_dim]
        return y.reshape(B, G, self.output_dim)

    def forward(self, processor_features: torch.Tensor, batch_size: int) -> torch.Tensor:
        return self.decode(processor_features, batch_size)


def fused_forward(encoder: "Encoder", processor: "Processor", decoder: "AssimilatorDecoder", features: torch.Tensor,
                  residual: torch.Tensor) -> torch.Tensor:
    """encoder -> processor -> decoder of the forecaster / GraphCast wrapper in native layouts.  Inference: every node update
    also makes the layer-1 node products (and zero-fills the aggregate) of the block that follows it - the encoder's for the
    first processor block, each processor block's for the next, the last one's for the decoder - so no projection launch runs
    between blocks.  Under autograd the blocks keep their separate differentiable projections."""
    B = int(features.shape[0])
    gp = processor.graph_processor
    if wide.encoder_is_wide(encoder) or wide.processor_is_wide(gp) or wide.decoder_is_wide(decoder):
        return wide.forward(encoder, processor, decoder, features, residual)
    _, lat_plan = encoder._plans(features.device)
    e_lat = encoder.latent_edge_embedding(lat_plan)
    fuse = (not _autograd_on(encoder, features) and not _autograd_on(gp) and not _autograd_on(decoder) and len(gp.blocks) > 0
            and gp.checkpoint_segments == 0)
    if fuse:
        enc_n = encoder.graph_processor.blocks[0].node_model.node_mlp.packed()
        first = gp.blocks[0].edge_model.edge_mlp.packed()
        dec_e = decoder.graph_processor.blocks[0].edge_model.edge_mlp.packed()
        last_n = gp.blocks[-1].node_model.node_mlp.packed()
        fuse = first.weight_dtype == enc_n.weight_dtype and dec_e.weight_dtype == last_n.weight_dtype
    if not fuse:
        x = encoder.encode(features)
        x, _ = gp.run_plan(x, lat_plan, e_lat, True, B, False)
        return decoder.decode(x, B, residual=residual)
    dec_prefetch = decoder.prefetch_tables(features.device)  # cold forward: the decoder's big tables on a side stream

    def join_prefetch():  # ... joined in front of decode
        if dec_prefetch is not None:
            torch.cuda.current_stream(features.device).wait_stream(dec_prefetch)

    x, posts, agg0 = encoder.encode(features, post_w=[first.w1[0], first.w1[1]])
    n_streams = gp.forward_streams(B)
    if n_streams <= 1:
        x, _, tail = gp.run_plan(x, lat_plan, e_lat, True, B, False, pre_proj=(posts[0], posts[1], agg0), tail_w=[dec_e.w1[0]],
                                 tail_half=decoder.team_path())
        join_prefetch()
        return decoder.decode(x, B, residual=residual, ps=None if tail is None else tail[0])
    # The mesh stack as independent per-sample chains on separate HIP streams (batch elements never interact,
    # encoder.py:212-218).  One batched launch of a mesh-sized kernel fills the chip unevenly - 1 287 64-column tiles on 512
    # workgroup slots are 2.5 rounds that cost 3, a node update has 184 workgroups for 256 CUs - and each launch waits for the
    # previous one; with the samples on different streams one chain's tail round and small launches run beside the other
    # chain's full rounds.  Everything shared (packed weights, cached products) is made on the main stream before the fork.
    dev = features.device
    M = encoder.num_h3
    main = torch.cuda.current_stream(dev)
    gp.prepare_shared(e_lat, lat_plan)
    streams = gp.side_streams(dev, n_streams)
    base, rem = divmod(B, n_streams)
    xs, tails, lo = [], [], 0
    for i, st in enumerate(streams):
        hi = lo + base + (1 if i < rem else 0)
        st.wait_stream(main)
        with torch.cuda.stream(st):
            pre = (posts[0][lo * M:hi * M], posts[1][lo * M:hi * M], agg0[lo * M:hi * M])
            xo, _, tl = gp.run_plan(x[lo * M:hi * M], lat_plan, e_lat, True, hi - lo, False, pre_proj=pre, tail_w=[dec_e.w1[0]],
                                    tail_half=decoder.team_path())
        xs.append(xo)
        tails.append(tl[0])
        lo = hi
    for st in streams:
        main.wait_stream(st)
        for t in (x, posts[0], posts[1], agg0):
            t.record_stream(st)  # allocated on the main stream, read on a side stream
    for t in xs + tails:
        t.record_stream(main)  # allocated on a side stream, read on the main stream
    join_prefetch()
    return decoder.decode(torch.cat(xs), B, residual=residual, ps=torch.cat(tails))


class Decoder(AssimilatorDecoder):
    """``Decoder`` - decoder.py:22-94."""

    def forward(self, processor_features: torch.Tensor, start_features: torch.Tensor, t: int = 0) -> torch.Tensor:
        B = int(start_features.shape[0])
        sf = start_features
        if sf.dim() != 3 or sf.shape[1] != self.num_latlons or sf.shape[2] != self.output_dim:
            raise RuntimeError("start_features must be [B, %d, %d]" % (self.num_latlons, self.output_dim))
        # fused residual: read start_features in place when it is a leading-channel view of a contiguous tensor
        base = None
        if sf.stride(2) == 1 and sf.stride(0) == sf.shape[1] * sf.stride(1) and sf.stride(1) >= sf.shape[2]:
            try:
                base = torch.as_strided(sf, (B * sf.shape[1], sf.stride(1)), (sf.stride(1), 1))
            except RuntimeError:
                base = None
        if base is None:
            base = sf.contiguous().reshape(B * sf.shape[1], sf.shape[2])
        return self.decode(processor_features, B, residual=base)
