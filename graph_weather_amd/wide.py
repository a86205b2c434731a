"""Message passing for models WIDER than 256 features.

The fused HIP kernels (``csrc/gw_kernels.hip``, ``gw_edge.hip``, ``gw_edge16.hip``) are built around 256-float rows; the
reference's own training script, however, constructs the forecaster with node / edge / hidden widths of 1024
(``train/run.py:493-497``).  Such models run here: the same modules, the same ``state_dict`` keys, every ``nn.Linear`` one
fp32-MFMA GEMM with bias and ReLU in its epilogue (``gw_linear_forward``), LayerNorm (+ residual) and the ``scatter_sum``
(graph_net_block.py:188) one kernel each at any width (``csrc/gw_wide.hip``), forward and backward.  Nothing is fused across
layers: this is the coverage path, not the tuned one.

What is kept from the native design: one shared destination-sorted graph plan for all batch elements (no replicated graph),
batch-independent embeddings computed once, the LAYER-1 SPLIT - the ``cat`` of graph_net_block.py:133 / :189 is never formed:
``cat[x_s, x_d, e] . W1^T = (x_s . Ws^T)[src] + (x_d . Wd^T)[dst] + e . We^T``, node products made once per node and gathered in
the epilogue of the edge-level GEMM (``gw_linear_gather_forward``), the decoder's zero grid rows dropped - and segment sums
that walk a CSR in one fixed order (bitwise reproducible, no atomics).  PyTorch moves data (slices, views) and links the
autograd nodes; no torch arithmetic op touches an activation.
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


def linear_gather_forward(x: Optional[torch.Tensor], w: Optional[torch.Tensor], bias: Optional[torch.Tensor], relu: bool,
                          adds: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], int]], rows: int, rows_per_batch: int) -> torch.Tensor:
    """act(x @ w.T + bias + sum_i table_i[b * rows_pb_i + idx_i[k]]) for row m = (b, k) = divmod(m, rows_per_batch); ``adds`` =
    [(table, idx or None, rows_pb)], at most three.  ``x`` / ``w`` None: no product (bias + gathered rows only)."""
    import ctypes

    n = int(adds[0][0].shape[1]) if w is None else int(w.shape[0])
    k = 0
    if w is not None:
        x, w = _rows(x, "x"), _rows(w, "weight")
        k = int(x.shape[1])
        if int(w.shape[1]) != k or int(x.shape[0]) != rows:
            raise RuntimeError("graph_weather_amd: Linear operand shapes do not match")
    na = len(adds)
    tabs = [_rows(t, "addend") for t, _, _ in adds]
    dev = tabs[0].device if na else x.device
    out = torch.empty((rows, n), dtype=torch.float32, device=dev)
    tp = (ctypes.c_void_p * max(na, 1))(*[t.data_ptr() for t in tabs])
    ip = (ctypes.c_void_p * max(na, 1))(*[None if i is None else i.data_ptr() for _, i, _ in adds])
    lp = (ctypes.c_int32 * max(na, 1))(*[_ld(t) for t in tabs])
    rp = (ctypes.c_int32 * max(na, 1))(*[int(r) for _, _, r in adds])
    with on_device_of(out):
        _lib.check(_L().gw_linear_gather_forward(rows, max(1, rows_per_batch), k, n, None if w is None else x.data_ptr(),
                                                 0 if w is None else _ld(x), None if w is None else w.data_ptr(),
                                                 0 if w is None else _ld(w), None if bias is None else bias.contiguous().data_ptr(), na,
                                                 tp, ip, lp, rp, 1 if relu else 0, out.data_ptr(), n, _st(out)),
                   "gw_linear_gather_forward")
    return out


def layernorm_forward(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, res: Optional[torch.Tensor],
                      res_period: int = 0) -> torch.Tensor:
    """``res_period`` > 0: the residual rows are shared by the batch (row m adds res[m % res_period])."""
    y = _rows(y, "y")
    rows, width = int(y.shape[0]), int(y.shape[1])
    if res is not None:
        res = _rows(res, "residual")
    out = torch.empty((rows, width), dtype=torch.float32, device=y.device)
    with on_device_of(out):
        _lib.check(_L().gw_layernorm_forward(rows, width, y.data_ptr(), _ld(y), gamma.contiguous().data_ptr(), beta.contiguous().data_ptr(),
                                             None if res is None else res.data_ptr(), 0 if res is None else _ld(res), int(res_period),
                                             out.data_ptr(), width, _st(out)), "gw_layernorm_forward")
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


def _relu_mask(dh: torch.Tensor, h: torch.Tensor, db: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dz = dh * (h > 0); ``db`` (zeroed by the caller) += column sums of dz."""
    dh = _rows(dh, "dh")
    h = None if h is None else _rows(h, "h")
    rows, width = int(dh.shape[0]), int(dh.shape[1])
    dz = torch.empty((rows, width), dtype=torch.float32, device=dh.device)
    with on_device_of(dz):
        _lib.check(_L().gw_relu_backward(rows, width, dh.data_ptr(), _ld(dh), None if h is None else h.data_ptr(),
                                         0 if h is None else _ld(h), dz.data_ptr(), width, None if db is None else db.data_ptr(), _st(dz)),
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
        ctx.relu, ctx.has_b = relu, b is not None
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
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dw = torch.zeros_like(w)
            db = torch.zeros((int(w.shape[0]),), dtype=torch.float32, device=w.device) if ctx.has_b else None
            with on_device_of(dw):
                gemm_tn_acc(dz, _rows(x, "x"), dw, colsum=db)
        return dx, dw, db, None


class _LinearGather(Function):
    """Linear (+ ReLU) with gathered row tables added before the activation (``linear_gather_forward``): the layer-1 split
    cat[x_s, x_d, e] . W1^T = (x_s . Ws^T)[src] + (x_d . Wd^T)[dst] + e . We^T of graph_net_block.py:131-134 / :189.
    ``meta[i]`` = (idx, rows_pb, (perm, ptr)): how table i is read and the CSR that groups the reading rows by table row (for its
    gradient, a segment sum of dz).  Backward of the product part as in ``_Linear``."""

    @staticmethod
    def forward(ctx, x, w, b, relu: bool, rows: int, rows_per_batch: int, meta, *tables):
        adds = [(t, m[0], m[1]) for t, m in zip(tables, meta)]
        out = linear_gather_forward(x, w, b, relu, adds, rows, rows_per_batch)
        ctx.relu, ctx.rows, ctx.rpb, ctx.meta = relu, rows, rows_per_batch, meta
        ctx.has_b = b is not None
        ctx.save_for_backward(x, w, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .autograd import gemm_tn_acc

        x, w, out = ctx.saved_tensors
        n = int(dout.shape[1])
        db = torch.zeros((n,), dtype=torch.float32, device=dout.device) if ctx.has_b else None
        if w is None:  # no product: the bias gradient rides on the mask kernel
            dz = _relu_mask(dout, out if ctx.relu else None, db) if (ctx.relu or db is not None) else _rows(dout, "dout")
        else:
            dz = _relu_mask(dout, out) if ctx.relu else _rows(dout, "dout")
        dx = dw = None
        if w is not None:
            if ctx.needs_input_grad[0]:
                dx = linear_forward(dz, w.detach().t().contiguous(), None, False)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dw = torch.zeros_like(w)
                with on_device_of(dw):
                    gemm_tn_acc(dz, _rows(x, "x"), dw, colsum=db)
        batch = ctx.rows // ctx.rpb
        dts = []
        for i, (idx, rows_pb, (perm, ptr)) in enumerate(ctx.meta):
            if not ctx.needs_input_grad[7 + i]:
                dts.append(None)
                continue
            if idx is None and rows_pb > 0:  # read row by row: the gradient is dz itself
                dts.append(dz)
            else:
                dts.append(segment_sum_rows(dz, ctx.rpb, batch, batch if rows_pb > 0 else 1, int(ptr.numel()) - 1, ptr, perm))
        return (dx, dw, db, None, None, None, None, *dts)


class _LayerNorm(Function):
    """LayerNorm (+ residual).  ``res_period`` > 0: residual rows shared by the batch - their gradient is summed over it."""

    @staticmethod
    def forward(ctx, y, gamma, beta, res, res_period: int = 0):
        out = layernorm_forward(y, gamma, beta, res, res_period)
        ctx.has_res, ctx.res_period = res is not None, int(res_period)
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
        dres = None
        if ctx.has_res and ctx.needs_input_grad[3]:
            if ctx.res_period > 0:
                p = ctx.res_period
                ident = torch.arange(p + 1, dtype=torch.int32, device=dn.device)
                dres = segment_sum_rows(dn, p, int(dn.shape[0]) // p, 1, p, ident, None)
            else:
                dres = dout
        return dy, dgamma, dbeta, dres, None


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


# ---------------------------------------------------------------------------------------------------------------------
# MLP, block, encoder / processor / decoder
# ---------------------------------------------------------------------------------------------------------------------
def is_wide(*mlps) -> bool:
    """True if any of the MLPs has a width the fused kernels do not take."""
    for m in mlps:
        if m.hidden_dim > 256 or m.out_dim > 256 or any(hi - lo > 256 for lo, hi in m._splits):
            return True
    return False


def _check_mlp(mlp) -> None:
    if mlp.compute_dtype != torch.float32:
        raise NotImplementedError("graph_weather_amd: bf16 matrix products exist for widths up to 256; wider models run in fp32")
    norm = mlp._norm()
    if norm is not None and abs(norm.eps - 1e-5) > 0:
        raise RuntimeError("graph_weather_amd: LayerNorm eps must be 1e-5")
    if not mlp._linears()[0].weight.is_cuda:
        raise RuntimeError("graph_weather_amd: module parameters must be on a HIP device (no CPU path exists)")


def _mlp_tail(mlp, h1: torch.Tensor, residual: Optional[torch.Tensor], res_period: int = 0) -> torch.Tensor:
    """Everything of an MLP behind its first Linear + ReLU (``h1``): hidden layers, last Linear, [LayerNorm], + residual
    (``res_period`` > 0: residual rows shared by the batch)."""
    lin, norm = mlp._linears(), mlp._norm()
    h = h1
    for m in lin[1:-1]:
        h = _Linear.apply(h, m.weight, m.bias, True)
    y = _Linear.apply(h, lin[-1].weight, lin[-1].bias, False)
    if norm is not None:
        return _LayerNorm.apply(y, norm.weight, norm.bias, residual, res_period)
    if residual is None:
        return y
    if res_period > 0:
        ident = torch.arange(res_period + 1, dtype=torch.int32, device=y.device)
        residual = _Gather.apply(residual, None, int(y.shape[0]) // res_period, 0, res_period, (None, ident))
    return _Add.apply(y, residual)


def mlp_rows(mlp, x2: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """graph_net_block.py:45-61,63-77 on rows: [rows, in] -> [rows, out_dim] (+ residual)."""
    _check_mlp(mlp)
    l0 = mlp._linears()[0]

    def run(x_, res_):
        return _mlp_tail(mlp, _Linear.apply(x_, l0.weight, l0.bias, True), res_)

    if mlp.use_checkpointing and torch.is_grad_enabled():  # graph_net_block.py:73-74
        from torch.utils.checkpoint import checkpoint

        return checkpoint(run, x2, residual, use_reentrant=False)
    return run(x2, residual)


def _cached(owner, name: str, params, fn, of: Optional[torch.Tensor] = None):
    """A batch-independent tensor of ``owner`` (an embedding, a layer-1 product of one): kept per parameter version while autograd
    is off; under autograd it is part of the graph and recomputed every step (the reference recomputes it on every forward).
    ``of``: an input tensor the value also depends on - part of the key, and held by the entry so that its address cannot be
    reused by another tensor while the entry lives."""
    params = list(params)
    if torch.is_grad_enabled() and (any(p.requires_grad for p in params) or (of is not None and of.requires_grad)):
        return fn()
    from .layers import KeyedCache, _ver, _version_key

    cache = owner.__dict__.get("_wide_cache")
    if cache is None:
        cache = owner.__dict__["_wide_cache"] = KeyedCache()
    key = _version_key(params) + (() if of is None else ((of.data_ptr(), _ver(of), tuple(of.shape)),))
    return cache.get(name, key, fn, hold=of)


def block(blk, plan: GraphPlan, batch: int, x_src: torch.Tensor, src_rows_pb: int, x_dst: Optional[torch.Tensor], dst_rows_pb: int,
          e: torch.Tensor, e_rows_pb: int, static=None):
    """One message-passing block (MetaLayer, graph_net_block.py:221-228) on the shared destination-sorted plan.
    ``x_src`` [batch * n_src, Dn] (``src_rows_pb`` = n_src) or one shared [n_src, Dn] (0); ``x_dst`` likewise, or None for
    destination rows that are all zero; ``e`` [batch * E, De] or shared [E, De] (``e_rows_pb`` 0).  Returns (x', e') for the
    destination rows of every batch element.

    The ``cat`` of :133 / :189 is never formed: layer 1 is split as in the fused kernels - cat[x_s, x_d, e] . W1^T =
    (x_s . Ws^T)[src] + (x_d . Wd^T)[dst] + e . We^T - so the node products are made once per node (an edge-level product costs
    ~7x as much) and gathered in the epilogue of the edge-level product (``_LinearGather``); with batch-shared edge features
    that product is per edge of ONE sample too and layer 1 has no edge-level matrix work at all.  ``static(name, fn)``: the
    caller's cache for products of batch-shared operands (they only change with the weights)."""
    if static is None:
        static = lambda name, fn: fn()  # noqa: E731
    E, nd = plan.num_edges, plan.n_dst
    emlp, nmlp = blk.edge_model.edge_mlp, blk.node_model.node_mlp
    _check_mlp(emlp)
    _check_mlp(nmlp)
    dn, de = int(x_src.shape[1]), int(e.shape[1])
    l0 = emlp._linears()[0]
    w0, b0 = l0.weight, l0.bias
    if int(w0.shape[1]) != 2 * dn + de:
        raise RuntimeError("graph_weather_amd: edge MLP expects %d input features, got 2 x %d + %d" % (int(w0.shape[1]), dn, de))
    tables = [_Linear.apply(x_src, w0[:, :dn], None, False)]
    metas = [(plan.src, src_rows_pb, plan.src_sorted())]
    if x_dst is not None:
        pd = lambda: _Linear.apply(x_dst, w0[:, dn:2 * dn], None, False)  # noqa: E731
        tables.append(static("pd", pd) if dst_rows_pb == 0 else pd())
        metas.append((plan.dst, dst_rows_pb, (None, plan.dst_ptr())))
    we = w0[:, 2 * dn:]
    if e_rows_pb == 0:
        tables.append(static("pe", lambda: _Linear.apply(e, we, None, False)))
        metas.append((None, 0, (None, _ident_ptr(plan, E))))
        h1 = _LinearGather.apply(None, None, b0, True, batch * E, E, tuple(metas), *tables)
    else:
        h1 = _LinearGather.apply(e, we, b0, True, batch * E, E, tuple(metas), *tables)
    e_new = _mlp_tail(emlp, h1, e, E if e_rows_pb == 0 else 0)                                     # :131-137
    agg = _SegmentSum.apply(e_new, plan, batch)                                                   # :188
    n0 = nmlp._linears()[0]
    wn, bn = n0.weight, n0.bias
    if int(wn.shape[1]) != dn + de:
        raise RuntimeError("graph_weather_amd: node MLP expects %d input features, got %d + %d" % (int(wn.shape[1]), dn, de))
    if x_dst is not None:                                                                         # :189-191
        pxf = lambda: _Linear.apply(x_dst, wn[:, :dn], None, False)  # noqa: E731
        px = static("px", pxf) if dst_rows_pb == 0 else pxf()
        h1n = _LinearGather.apply(agg, wn[:, dn:], bn, True, batch * nd, nd, ((None, dst_rows_pb, (None, _ident_ptr(plan, nd))),), px)
        x_new = _mlp_tail(nmlp, h1n, x_dst, nd if dst_rows_pb == 0 else 0)
    else:
        x_new = _mlp_tail(nmlp, _Linear.apply(agg, wn[:, dn:], bn, True), None)
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
            blk = gp.blocks[i]
            static = None
            if shared:  # We . e of batch-shared edge features: per weight version (and per tensor e) in inference
                static = lambda name, fn, blk=blk, e_in=e_, i=i: _cached(gp, "%s%d" % (name, i), blk.parameters(), fn, of=e_in)  # noqa: E731
            x_, e_ = block(blk, plan, batch, x_, n, x_, n, e_, 0 if shared else E, static=static)
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
    ps = list(enc.parameters())
    xg = mlp_rows(enc.node_encoder, features.contiguous().reshape(B * G, F))
    # batch independent (encoder.py:199-205 repeats them per sample): mesh-node and edge embeddings, their layer-1 products
    xm = _cached(enc, "xm", ps, lambda: mlp_rows(enc.node_encoder, enc.h3_nodes))
    e = _cached(enc, "e_enc", ps, lambda: mlp_rows(enc.edge_encoder, enc_plan.edge_attr))
    x, _ = block(enc.graph_processor.blocks[0], enc_plan, B, xg, G, xm, 0, e, 0,
                 static=lambda name, fn: _cached(enc, "enc_" + name, ps, fn))
    return x


def latent_edges(enc, plan: GraphPlan) -> torch.Tensor:
    """latent_edge_encoder(attr) once, in destination-sorted order (encoder.py:235-241 repeats it B times)."""
    return _cached(enc, "e_lat", enc.latent_edge_encoder.parameters(), lambda: mlp_rows(enc.latent_edge_encoder, plan.edge_attr))


def decode(dec, processor_features: torch.Tensor, batch_size: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """assimilator_decoder.py:173-200 (+ decoder.py:93 with ``residual`` [B * G, >= output_dim])."""
    B, G, M = batch_size, dec.num_latlons, dec.num_h3
    if processor_features.shape[0] != B * M:
        raise RuntimeError("processor_features must have batch*num_h3 rows")
    plan = dec._plan(processor_features.device)
    ps = list(dec.edge_encoder.parameters()) + list(dec.graph_processor.parameters())
    e = _cached(dec, "e_dec", ps, lambda: mlp_rows(dec.edge_encoder, plan.edge_attr))
    # grid rows are zeros (assimilator_decoder.py:84,190-192): no x_dst operand, node input [0 | agg], residual 0
    xg, _ = block(dec.graph_processor.blocks[0], plan, B, processor_features.contiguous(), M, None, 0, e, 0,
                  static=lambda name, fn: _cached(dec, "dec_" + name, ps, fn))
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
