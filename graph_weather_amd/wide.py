"""Message passing for models WIDER than 256 features.

The fused HIP kernels (``csrc/gw_kernels.hip``, ``gw_edge.hip``, ``gw_edge16.hip``) are built around 256-float rows; the
reference's own training script, however, constructs the forecaster with node / edge / hidden widths of 1024
(``train/run.py:493-497``).  Such models run here: the same modules, the same ``state_dict`` keys, every ``nn.Linear`` one
fp32-MFMA GEMM with bias and ReLU in its epilogue (``gw_linear_forward``), LayerNorm, the ``x[row]`` / ``x[col]`` gathers
(MetaLayer, graph_net_block.py:221-228) and the ``scatter_sum`` (:188) one kernel each at any width (``csrc/gw_wide.hip``),
forward and backward.  Nothing is fused across layers and the ``cat`` of graph_net_block.py:133 / :189 is materialised as in
the reference: this is the coverage path, not the tuned one.

What is kept from the native path: one shared destination-sorted graph plan for all batch elements (no replicated graph),
batch-independent embeddings computed once, layer-1 weight columns of operands that are identically zero (the decoder's
grid rows, assimilator_decoder.py:84,190-192) skipped, and segment sums that walk a CSR in one fixed order (bitwise
reproducible, no atomics).  PyTorch moves data (``cat``, slices, views) and links the autograd nodes; no torch arithmetic
op touches an activation.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _lib
from .graphs import GraphPlan
from .ops import on_device_of


def _st(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _L():
    return _lib.lib()


def _rows(t: torch.Tensor, name: str) -> torch.Tensor:
    """A 2-D fp32 HIP tensor whose rows are contiguous (any row stride)."""
    if not t.is_cuda:
        raise RuntimeError(f"graph_weather_amd: {name} must live on a HIP device (no CPU path exists)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"graph_weather_amd: {name} must be float32, got {t.dtype}")
    if t.dim() != 2:
        raise RuntimeError(f"graph_weather_amd: {name} must be 2-D")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


# ---------------------------------------------------------------------------------------------------------------------
# kernel wrappers
# ---------------------------------------------------------------------------------------------------------------------
def linear_forward(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """act(x @ w.T + bias): x [rows, k], w [n, k] (nn.Linear.weight)."""
    x, w = _rows(x, "x"), _rows(w, "weight")
    rows, k, n = int(x.shape[0]), int(x.shape[1]), int(w.shape[0])
    if int(w.shape[1]) != k:
        raise RuntimeError("graph_weather_amd: Linear expects %d input features, got %d" % (int(w.shape[1]), k))
    out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    with on_device_of(out):
        _lib.check(_L().gw_linear_forward(rows, k, n, x.data_ptr(), _ld(x), w.data_ptr(), _ld(w),
                                          None if bias is None else bias.contiguous().data_ptr(), 1 if relu else 0, out.data_ptr(), n,
                                          _st(out)), "gw_linear_forward")
    return out


def layernorm_forward(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, res: Optional[torch.Tensor]) -> torch.Tensor:
    y = _rows(y, "y")
    rows, width = int(y.shape[0]), int(y.shape[1])
    if res is not None:
        res = _rows(res, "residual")
    out = torch.empty((rows, width), dtype=torch.float32, device=y.device)
    with on_device_of(out):
        _lib.check(_L().gw_layernorm_forward(rows, width, y.data_ptr(), _ld(y), gamma.contiguous().data_ptr(), beta.contiguous().data_ptr(),
                                             None if res is None else res.data_ptr(), 0 if res is None else _ld(res), out.data_ptr(),
                                             width, _st(out)), "gw_layernorm_forward")
    return out


def add_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = _rows(a, "a"), _rows(b, "b")
    rows, width = int(a.shape[0]), int(a.shape[1])
    out = torch.empty((rows, width), dtype=torch.float32, device=a.device)
    with on_device_of(out):
        _lib.check(_L().gw_add_rows(rows, width, a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), out.data_ptr(), width, _st(out)),
                   "gw_add_rows")
    return out


def gather_rows(table: torch.Tensor, rows_pb: int, idx: Optional[torch.Tensor], batch: int, n_idx: int) -> torch.Tensor:
    """out[b, i] = table[b * rows_pb + idx[i]] (idx None: identity; rows_pb 0: one table shared by the batch)."""
    table = _rows(table, "table")
    width = int(table.shape[1])
    out = torch.empty((batch * n_idx, width), dtype=torch.float32, device=table.device)
    with on_device_of(out):
        _lib.check(_L().gw_gather_rows_wide(batch, n_idx, width, table.data_ptr(), _ld(table), rows_pb,
                                            None if idx is None else idx.data_ptr(), out.data_ptr(), width, _st(out)),
                   "gw_gather_rows_wide")
    return out


def segment_sum_rows(rows: torch.Tensor, rows_pb_in: int, batch: int, batch_out: int, n_seg: int, ptr: torch.Tensor,
                     perm: Optional[torch.Tensor]) -> torch.Tensor:
    rows = _rows(rows, "rows")
    width = int(rows.shape[1])
    out = torch.empty((batch_out * n_seg, width), dtype=torch.float32, device=rows.device)
    with on_device_of(out):
        _lib.check(_L().gw_segment_sum_rows_wide(batch, batch_out, n_seg, width, rows.data_ptr(), _ld(rows), rows_pb_in,
                                                 None if perm is None else perm.data_ptr(), ptr.data_ptr(), out.data_ptr(), width,
                                                 _st(out)), "gw_segment_sum_rows_wide")
    return out


def _relu_mask(dh: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    dh, h = _rows(dh, "dh"), _rows(h, "h")
    rows, width = int(dh.shape[0]), int(dh.shape[1])
    dz = torch.empty((rows, width), dtype=torch.float32, device=dh.device)
    with on_device_of(dz):
        _lib.check(_L().gw_relu_backward(rows, width, dh.data_ptr(), _ld(dh), h.data_ptr(), _ld(h), dz.data_ptr(), width, None, _st(dz)),
                   "gw_relu_backward")
    return dz


# ---------------------------------------------------------------------------------------------------------------------
# autograd nodes
# ---------------------------------------------------------------------------------------------------------------------
class _Linear(Function):
    """nn.Linear (+ nn.ReLU).  Backward: dz = dout * (out > 0); dx = dz @ W (the same GEMM kernel on W^T); dW += dz^T @ x and
    db += column sums of dz in one TN GEMM (gw_gemm_f32)."""

    @staticmethod
    def forward(ctx, x, w, b, relu: bool):
        out = linear_forward(x, w, b, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .autograd import gemm_tn_acc

        x, w, out = ctx.saved_tensors
        dz = _relu_mask(dout, out) if ctx.relu else _rows(dout, "dout")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_forward(dz, w.detach().t().contiguous(), None, False)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.zeros_like(w)
            db = torch.zeros((int(w.shape[0]),), dtype=torch.float32, device=w.device)
            with on_device_of(dw):
                gemm_tn_acc(dz, _rows(x, "x"), dw, colsum=db)
        return dx, dw, db, None


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, res):
        out = layernorm_forward(y, gamma, beta, res)
        ctx.has_res = res is not None
        ctx.save_for_backward(y, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .autograd import layernorm_backward

        y, gamma = ctx.saved_tensors
        dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
        dn = dout.contiguous()
        with on_device_of(dn):
            dy = layernorm_backward(dn, _rows(y, "y"), gamma.contiguous(), dgamma, dbeta)
        return dy, dgamma, dbeta, (dout if ctx.has_res else None)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return add_rows(a, b)

    @staticmethod
    def backward(ctx, dout):
        return dout, dout


class _Gather(Function):
    """out[b, i] = table[b, idx[i]]; backward = segment sum over the positions that read each table row (``back`` = (perm, ptr):
    positions grouped by table row)."""

    @staticmethod
    def forward(ctx, table, idx, batch: int, rows_pb: int, n_idx: int, back):
        ctx.meta = (batch, rows_pb, n_idx)
        ctx.back = back
        return gather_rows(table, rows_pb, idx, batch, n_idx)

    @staticmethod
    def backward(ctx, dout):
        batch, rows_pb, n_idx = ctx.meta
        perm, ptr = ctx.back
        n_seg = int(ptr.numel()) - 1
        dtable = segment_sum_rows(dout.contiguous(), n_idx, batch, batch if rows_pb > 0 else 1, n_seg, ptr, perm)
        return dtable, None, None, None, None, None


class _SegmentSum(Function):
    """scatter_sum over the destination-sorted edges (graph_net_block.py:188); backward = gather by destination."""

    @staticmethod
    def forward(ctx, rows, plan: GraphPlan, batch: int):
        ctx.plan, ctx.batch = plan, batch
        return segment_sum_rows(rows, plan.num_edges, batch, batch, plan.n_dst, plan.dst_ptr(), None)

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        return gather_rows(dout.contiguous(), plan.n_dst, plan.dst, ctx.batch, plan.num_edges), None, None


def _ident_ptr(plan: GraphPlan, n: int) -> torch.Tensor:
    cache = plan.__dict__.setdefault("_wide_ident", {})
    if n not in cache:
        cache[n] = torch.arange(n + 1, dtype=torch.int32, device=plan.src.device)
    return cache[n]


def _broadcast(table: torch.Tensor, batch: int, plan: GraphPlan) -> torch.Tensor:
    """A batch-shared table repeated for every batch element (its gradient is summed over the batch)."""
    n = int(table.shape[0])
    return _Gather.apply(table, None, batch, 0, n, (None, _ident_ptr(plan, n)))


# ---------------------------------------------------------------------------------------------------------------------
# MLP, block, encoder / processor / decoder
# ---------------------------------------------------------------------------------------------------------------------
def is_wide(*mlps) -> bool:
    """True if any of the MLPs has a width the fused kernels do not take."""
    for m in mlps:
        if m.hidden_dim > 256 or m.out_dim > 256 or any(hi - lo > 256 for lo, hi in m._splits):
            return True
    return False


def mlp_rows(mlp, x2: torch.Tensor, residual: Optional[torch.Tensor] = None,
             w1_cols: Optional[Sequence[Tuple[int, int]]] = None) -> torch.Tensor:
    """graph_net_block.py:45-61,63-77 on rows: [rows, in] -> [rows, out_dim] (+ residual).  ``w1_cols``: the column ranges of the
    first Linear that ``x2`` feeds, when an operand of the reference's ``cat`` is identically zero and is left out."""
    if mlp.compute_dtype != torch.float32:
        raise NotImplementedError("graph_weather_amd: bf16 matrix products exist for widths up to 256; wider models run in fp32")
    lin, norm = mlp._linears(), mlp._norm()
    if norm is not None and abs(norm.eps - 1e-5) > 0:
        raise RuntimeError("graph_weather_amd: LayerNorm eps must be 1e-5")
    if not lin[0].weight.is_cuda:
        raise RuntimeError("graph_weather_amd: module parameters must be on a HIP device (no CPU path exists)")

    def run(x_, res_):
        w0 = lin[0].weight if w1_cols is None else torch.cat([lin[0].weight[:, lo:hi] for lo, hi in w1_cols], dim=1)
        h = _Linear.apply(x_, w0, lin[0].bias, True)
        for m in lin[1:-1]:
            h = _Linear.apply(h, m.weight, m.bias, True)
        y = _Linear.apply(h, lin[-1].weight, lin[-1].bias, False)
        if norm is not None:
            return _LayerNorm.apply(y, norm.weight, norm.bias, res_)
        return y if res_ is None else _Add.apply(y, res_)

    if mlp.use_checkpointing and torch.is_grad_enabled():  # graph_net_block.py:73-74
        from torch.utils.checkpoint import checkpoint

        return checkpoint(run, x2, residual, use_reentrant=False)
    return run(x2, residual)


def block(blk, plan: GraphPlan, batch: int, x_src: torch.Tensor, src_rows_pb: int, x_dst: Optional[torch.Tensor], dst_rows_pb: int,
          e: torch.Tensor, e_rows_pb: int):
    """One message-passing block (MetaLayer, graph_net_block.py:221-228) on the shared destination-sorted plan.
    ``x_src`` [batch * n_src, Dn] (``src_rows_pb`` = n_src) or one shared [n_src, Dn] (0); ``x_dst`` likewise, or None for
    destination rows that are all zero; ``e`` [batch * E, De] or shared [E, De] (``e_rows_pb`` 0).  Returns (x', e') for the
    destination rows of every batch element."""
    E, nd = plan.num_edges, plan.n_dst
    emlp, nmlp = blk.edge_model.edge_mlp, blk.node_model.node_mlp
    dn, de = int(x_src.shape[1]), int(e.shape[1])
    xs = _Gather.apply(x_src, plan.src, batch, src_rows_pb, E, plan.src_sorted())
    parts, cols = [xs], [(0, dn)]
    if x_dst is not None:
        parts.append(_Gather.apply(x_dst, plan.dst, batch, dst_rows_pb, E, (None, plan.dst_ptr())))
        cols.append((dn, 2 * dn))
    e_b = e if e_rows_pb > 0 else _broadcast(e, batch, plan)
    parts.append(e_b)
    cols.append((2 * dn, 2 * dn + de))
    e_new = mlp_rows(emlp, torch.cat(parts, dim=1), residual=e_b, w1_cols=None if x_dst is not None else cols)   # :131-137
    agg = _SegmentSum.apply(e_new, plan, batch)                                                                   # :188
    if x_dst is not None:
        xd = x_dst if dst_rows_pb > 0 else _broadcast(x_dst, batch, plan)
        x_new = mlp_rows(nmlp, torch.cat([xd, agg], dim=1), residual=xd)                                          # :189-191
    else:
        x_new = mlp_rows(nmlp, agg, w1_cols=[(dn, dn + de)])
    return x_new, e_new


def run_blocks(gp, x: torch.Tensor, plan: GraphPlan, e: torch.Tensor, e_shared: bool, batch: int, want_edges: bool):
    """``GraphProcessor`` (graph_net_block.py:279-301) on node rows [batch * n, Dn] and edge rows in sorted order."""
    n, E = plan.n_dst, plan.num_edges
    nb = len(gp.blocks)
    seg = 0
    if torch.is_grad_enabled():
        if gp.checkpoint_segments == -1:
            seg = nb
        elif gp.checkpoint_segments > 0:
            seg = int(gp.checkpoint_segments)
        elif gp.use_checkpointing:
            seg = 1

    def span(lo, hi, x_, e_, shared):
        for i in range(lo, hi):
            x_, e_ = block(gp.blocks[i], plan, batch, x_, n, x_, n, e_, 0 if shared else E)
            shared = False
        return x_, e_

    if seg <= 0 or nb == 0:
        x, e = span(0, nb, x, e, e_shared)
        return x, (e if want_edges else None)
    from torch.utils.checkpoint import checkpoint

    shared = e_shared
    for lo in range(0, nb, seg):
        hi = min(nb, lo + seg)
        x, e = checkpoint(lambda x_, e_, lo=lo, hi=hi, shared=shared: span(lo, hi, x_, e_, shared), x, e, use_reentrant=False)
        shared = False
    return x, (e if want_edges else None)


def encoder_is_wide(enc) -> bool:
    b = enc.graph_processor.blocks[0]
    return is_wide(enc.node_encoder, enc.edge_encoder, enc.latent_edge_encoder, b.edge_model.edge_mlp, b.node_model.node_mlp)


def processor_is_wide(gp) -> bool:
    return any(is_wide(b.edge_model.edge_mlp, b.node_model.node_mlp) for b in gp.blocks)


def decoder_is_wide(dec) -> bool:
    b = dec.graph_processor.blocks[0]
    return is_wide(dec.edge_encoder, dec.node_decoder, b.edge_model.edge_mlp, b.node_model.node_mlp)


def encode(enc, features: torch.Tensor) -> torch.Tensor:
    """encoder.py:199-223 -> mesh node rows [(B * M), D] (batch-major, the mesh order of the native path)."""
    if features.dim() != 3 or features.shape[1] != enc.num_latlons:
        raise RuntimeError("features must be [B, %d, input_dim]" % enc.num_latlons)
    B, G, F = (int(s) for s in features.shape)
    enc_plan, _ = enc._plans(features.device)
    xg = mlp_rows(enc.node_encoder, features.contiguous().reshape(B * G, F))
    xm = mlp_rows(enc.node_encoder, enc.h3_nodes)            # batch independent (encoder.py:199-205 repeats it per sample)
    e = mlp_rows(enc.edge_encoder, enc_plan.edge_attr)
    x, _ = block(enc.graph_processor.blocks[0], enc_plan, B, xg, G, xm, 0, e, 0)
    return x


def latent_edges(enc, plan: GraphPlan) -> torch.Tensor:
    """latent_edge_encoder(attr) once, in destination-sorted order (encoder.py:235-241 repeats it B times)."""
    return mlp_rows(enc.latent_edge_encoder, plan.edge_attr)


def decode(dec, processor_features: torch.Tensor, batch_size: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """assimilator_decoder.py:173-200 (+ decoder.py:93 with ``residual`` [B * G, >= output_dim])."""
    B, G, M = batch_size, dec.num_latlons, dec.num_h3
    if processor_features.shape[0] != B * M:
        raise RuntimeError("processor_features must have batch*num_h3 rows")
    plan = dec._plan(processor_features.device)
    e = mlp_rows(dec.edge_encoder, plan.edge_attr)
    # grid rows are zeros (assimilator_decoder.py:84,190-192): no x_dst operand, node input [0 | agg], residual 0
    xg, _ = block(dec.graph_processor.blocks[0], plan, B, processor_features.contiguous(), M, None, 0, e, 0)
    res = None
    if residual is not None:
        if residual.dim() != 2 or residual.shape[0] != B * G or residual.shape[1] < dec.output_dim:
            raise RuntimeError("graph_weather_amd: the residual (start features) must have batch*num_latlons rows of at least "
                               "output_dim = %d features, got %s" % (dec.output_dim, tuple(residual.shape)))
        res = residual[:, :dec.output_dim]
    y = mlp_rows(dec.node_decoder, xg, residual=res)
    return y.reshape(B, G, dec.output_dim)


def forward(encoder, processor, decoder, features: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    """forecast.py:226-228 for a model any of whose parts is wider than the fused kernels."""
    B = int(features.shape[0])
    _, lat_plan = encoder._plans(features.device)
    x = encode(encoder, features)
    x, _ = run_blocks(processor.graph_processor, x, lat_plan, latent_edges(encoder, lat_plan), True, B, False)
    return decode(decoder, x, B, residual=residual)
