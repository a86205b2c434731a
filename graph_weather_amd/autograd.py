"""Backward pass of the hot path: ``torch.autograd.Function`` wrappers around the C ABI.

The reference trains through autograd over ATen / torch_scatter ops (``train/run.py:509-521``: ``loss.backward();
optimizer.step()``).  Here each fused forward op (``gw_mlp_forward``, ``gw_project_forward``, ``gw_edge_update_forward``,
``gw_node_update_forward``, ``gw_normalized_mse_forward``) is one autograd node whose backward is composed from the
HIP kernels of ``csrc/gw_train.hip`` (weight-gradient GEMMs in fp32 or on split operands, gather and segment-sum duals) and the
register-resident input-gradient chain of an MLP (``gw_mlp_ln_chain_backward``: LayerNorm backward, the masked products of the
Linear / ReLU chain, the layer-0 input gradients, the first Linear's bias gradient and a joining gradient in ONE launch; fp32 and
bf16x3).  PyTorch only links the nodes and sums gradients of tensors that are used more than once - where both uses meet in one
node (an edge row as operand and residual of its block; an edge embedding as product and residual: ``ProjectFunction``
``passthrough``) the sum happens inside that launch; no torch arithmetic op touches an activation.  Forward calls made under
autograd also write the activations the reference's autograd would have saved (relu outputs, pre-LayerNorm rows) - SURVEY.md
appendix G.  MLPs off the kernel shapes (narrow heads, wide models) run the same steps as separate launches.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .ops import Operand, SavedActivations


def _st(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _L():
    return _lib.lib()


# ---------------------------------------------------------------------------------------------------------------------
# thin wrappers over the generic kernels
# ---------------------------------------------------------------------------------------------------------------------
def gemm_nn(a: torch.Tensor, b: torch.Tensor, n: int, b_col0: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m, :n] = a[m, :k] @ b[:k, b_col0:b_col0+n]   (k = a.shape[1])."""
    m, k = int(a.shape[0]), int(a.shape[1])
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _lib.check(_L().gw_gemm_f32(_lib.GEMM_NN, m, n, k, a.data_ptr(), int(a.stride(0)), b.data_ptr() + 4 * b_col0,
                                int(b.stride(0)), out.data_ptr(), int(out.stride(0)), None, _st(a)), "gw_gemm_f32 NN")
    return out


def _x3(mlp) -> bool:
    """The MLP's matrix products run on split operands (bf16x3): its weight-gradient GEMMs do too."""
    return mlp is not None and getattr(mlp, "compute_dtype", None) == ops.BF16X3


def gemm_tn_acc(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, c_col0: int = 0, colsum: Optional[torch.Tensor] = None,
                x3: bool = False) -> None:
    """c[:ma, c_col0:c_col0+nb] += a^T @ b   (a [rows, ma], b [rows, nb]);  colsum[:ma] += column sums of a.  ``x3``: products on
    split operands (include/gw_amd.h: GW_GEMM_TN_BF16X3; any shape)."""
    rows, ma, nb = int(a.shape[0]), int(a.shape[1]), int(b.shape[1])
    _lib.check(_L().gw_gemm_f32(_lib.GEMM_TN_BF16X3 if x3 else _lib.GEMM_TN, ma, nb, rows, a.data_ptr(), int(a.stride(0)), b.data_ptr(), int(b.stride(0)),
                                c.data_ptr() + 4 * c_col0, int(c.stride(0)), None if colsum is None else colsum.data_ptr(), _st(a)),
               "gw_gemm_f32 TN")


def relu_backward(dh: torch.Tensor, h: Optional[torch.Tensor], db: Optional[torch.Tensor]) -> torch.Tensor:
    """in place: dh *= (h > 0); db += column sums."""
    rows, width = int(dh.shape[0]), int(dh.shape[1])
    _lib.check(_L().gw_relu_backward(rows, width, dh.data_ptr(), int(dh.stride(0)), None if h is None else h.data_ptr(),
                                     0 if h is None else int(h.stride(0)), dh.data_ptr() if h is not None else None,
                                     int(dh.stride(0)), None if db is None else db.data_ptr(), _st(dh)), "gw_relu_backward")
    return dh


def layernorm_backward(dn: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                       width: int = 0) -> torch.Tensor:
    """``y``: saved pre-norm rows [rows, >= width] (heads are saved zero padded to 80 columns).  ``width`` = features the
    LayerNorm spans (default: ``gamma.numel()``); for zero-padded narrow models gamma / dn / the result are 256 wide and
    only the first ``width`` columns carry values (the rest of dy is zero)."""
    cols = int(gamma.numel())
    width = cols if width <= 0 else width
    alloc = torch.empty if width == cols else torch.zeros
    dy = alloc((int(y.shape[0]), cols), dtype=torch.float32, device=y.device)
    _lib.check(_L().gw_layernorm_backward(int(y.shape[0]), width, dn.data_ptr(), int(dn.stride(0)), y.data_ptr(), int(y.stride(0)),
                                          gamma.data_ptr(), dy.data_ptr(), int(dy.stride(0)), dgamma.data_ptr(), dbeta.data_ptr(),
                                          _st(y)), "gw_layernorm_backward")
    return dy


def gather_rows(table: torch.Tensor, rows_pb: int, idx: Optional[torch.Tensor], batch: int, n_idx: int,
                add: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty((batch * n_idx, 256), dtype=torch.float32, device=table.device)
    _lib.check(_L().gw_gather_rows(batch, n_idx, table.data_ptr(), rows_pb, None if idx is None else idx.data_ptr(),
                                   None if add is None else add.data_ptr(), out.data_ptr(), _st(table)), "gw_gather_rows")
    return out


def segment_sum_rows(rows: torch.Tensor, rows_pb_in: int, batch: int, batch_out: int, n_seg: int, ptr: torch.Tensor,
                     perm: Optional[torch.Tensor]) -> torch.Tensor:
    out = torch.empty((batch_out * n_seg, 256), dtype=torch.float32, device=rows.device)
    _lib.check(_L().gw_segment_sum_rows(batch, batch_out, n_seg, rows.data_ptr(), rows_pb_in, None if perm is None else perm.data_ptr(),
                                        ptr.data_ptr(), out.data_ptr(), 0, _st(rows)), "gw_segment_sum_rows")
    return out


def _packed_transposed(mlp, layer: int, W: torch.Tensor, lo: int, hi: int):
    """Packed stream of W[:, lo:hi]^T of (kernel-shaped) Linear ``layer`` (cached per weight version) for the fast
    input-gradient product d @ W[:, lo:hi] through the forward's single-layer kernel; only for 256 x 256 blocks, else None.
    On a miss every such block of the MLP is packed by one ``gw_pack_many`` launch, straight from the weights (the items
    address the transposed block by strides: no transposed copy).  The stream has the MLP's matrix-product dtype: fp32, or
    the split bf16 pairs of the bf16x3 mode (the same three-MFMA products in the backward as in the forward)."""
    if W.shape[0] != 256 or hi - lo != 256:
        return None
    cache = mlp.__dict__.setdefault("_packed_t", {})
    wd = ops.gw_dtype_of(mlp.compute_dtype)
    ver = (mlp.native_key(), wd)  # versions of the parameters W was derived from (W itself may be a fresh zero-padded copy)
    if cache.get("ver") != ver:
        cache.clear()
        cache["ver"] = ver
        with torch.no_grad():
            ps = [p.detach() for p in mlp.native_params()]
        n_lin = (len(ps) - (2 if mlp._norm() is not None else 0)) // 2
        blocks = []
        for l in range(n_lin):
            Wl = ps[2 * l]
            if Wl.shape[0] != 256:
                continue
            if Wl.dtype != torch.float32 or not Wl.is_contiguous():
                Wl = Wl.contiguous().float()
            for a, b in (mlp.native_splits() if l == 0 else ((0, int(Wl.shape[1])),)):
                if b - a == 256:
                    blocks.append((l, a, b, Wl))
        if blocks:
            if wd == _lib.DTYPE_BF16X3:
                n, tdt = int(_L().gw_packed_bytes_bf16x3(256, 0, 256)) // 2, torch.int16
            else:
                n, tdt = int(_L().gw_packed_floats(256, 0, 256)), torch.float32
            buf = torch.empty(len(blocks) * n, dtype=tdt, device=W.device)
            mats = []
            for i, (l, a, b, Wl) in enumerate(blocks):
                out = buf[i * n:(i + 1) * n]
                # "Linear" that maps gradients back: output feature f = input column a + f, input feature k = row k of W
                mats.append((Wl.data_ptr() + 4 * a, 1, int(Wl.shape[1]), 256, 256, out.data_ptr()))
                cache[(l, a, b)] = out
            ops.pack_many(wd, mats, [], _st(W))
    return cache.get((layer, lo, hi))


def input_grad(mlp, layer: int, d: torch.Tensor, W: torch.Tensor, lo: int, hi: int,
               relu_of: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(d @ W[:, lo:hi]) [* (relu_of > 0)]  ([rows, hi-lo]): fused single-layer kernel for 256 x 256 blocks (the ReLU
    backward rides in its epilogue), generic GEMM + mask kernel otherwise."""
    pt = _packed_transposed(mlp, layer, W, lo, hi) if (d.shape[1] == 256 and d.stride(0) % 4 == 0) else None
    if pt is not None and (relu_of is None or (relu_of.shape[1] == 256 and relu_of.stride(0) == 256)):
        rows = int(d.shape[0])
        return ops.project_forward([pt], Operand(d, rows, 256), rows, rows, relu_mask=relu_of)[0]  # (dtype from the stream)
    out = gemm_nn(d, W, hi - lo, b_col0=lo)
    if relu_of is not None:
        relu_backward(out, relu_of, None)
    return out


FAN_ADD_INPUT = "input gradient"  # fan_add entry: the launch's own input gradient row joins that product (as gathered)


def chain_backward(d: torch.Tensor, chain, fan, ln=None, colsum=None, fan_add=None, gather=None):
    """gw_mlp_chain_backward[_bf16x3]: ``chain`` = [(packed W^T, relu output, out)], ``fan`` = [(packed W^T block, out)]; all rows
    x 256.  The packed streams carry the dtype (fp32, or int16 words of the split stream).  Extras of gw_mlp_ln_chain_backward:
    ``ln`` = (pre-norm rows, gamma, dgamma, dbeta, dy): ``d`` is the gradient at the OUTPUT of the MLP's LayerNorm and the launch
    walks back through the norm first; ``colsum`` [256] += column sums of the last chain gradient; ``fan_add`` = one tensor, None
    or FAN_ADD_INPUT per fan item, added to that product before it is stored; ``gather`` = (idx int32 [n], table rows per batch
    element, add rows or None), with ``ln`` only: row b * n + k of the launch reads ``d[b * table_rows + idx[k]]`` (+ its row of
    add) - ``d`` is then the table and the gathered rows are never written."""
    import ctypes as C

    def arr(ptrs):
        a = (C.c_void_p * max(len(ptrs), 1))()
        for i, v in enumerate(ptrs):
            a[i] = v
        return a

    streams = [c[0] for c in chain] + [f[0] for f in fan]
    if len({t.dtype for t in streams}) > 1:
        raise RuntimeError("graph_weather_amd: chain_backward: packed streams of different dtypes")
    x3 = bool(streams) and streams[0].dtype != torch.float32  # (no product at all: the C entry refuses)
    n_rows = int(chain[0][2].shape[0]) if chain else (int(fan[0][1].shape[0]) if fan else 0)
    chain_items = (len(chain), arr([c[0].data_ptr() for c in chain]), arr([c[1].data_ptr() for c in chain]), arr([c[2].data_ptr() for c in chain]))
    fan_items = (len(fan), arr([f[0].data_ptr() for f in fan]), arr([f[1].data_ptr() for f in fan]))
    fan_add = list(fan_add or [])
    adds = [t for t in fan_add if isinstance(t, torch.Tensor)]
    mask = sum(1 << i for i, t in enumerate(fan_add) if t is FAN_ADD_INPUT)
    if ln is not None or colsum is not None or adds or mask or gather is not None:
        y, gamma, dgamma, dbeta, dy = ln if ln is not None else (None,) * 5
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        if (adds or mask) and (len(fan_add) != len(fan) or len({int(t.stride(0)) for t in adds}) > 1):
            raise RuntimeError("graph_weather_amd: chain_backward: fan_add needs one entry per fan item and one leading dimension")
        idx, tab_rows, gadd = gather if gather is not None else (None, 0, None)
        _lib.check(_L().gw_mlp_ln_chain_backward(_lib.DTYPE_BF16X3 if x3 else _lib.DTYPE_F32, n_rows, d.data_ptr(), int(d.stride(0)),
                                                 ptr(idx), 0 if idx is None else int(idx.numel()), int(tab_rows), ptr(gadd),
                                                 0 if gadd is None else int(gadd.stride(0)),
                                                 ptr(y), ptr(gamma), ptr(dgamma), ptr(dbeta), ptr(dy), *chain_items, ptr(colsum), *fan_items,
                                                 arr([t.data_ptr() if isinstance(t, torch.Tensor) else 0 for t in fan_add]) if adds else None,
                                                 int(adds[0].stride(0)) if adds else 0, mask, _st(d)),
                   "gw_mlp_ln_chain_backward")
        return
    fn = _L().gw_mlp_chain_backward_bf16x3 if x3 else _L().gw_mlp_chain_backward
    _lib.check(fn(int(d.shape[0]), d.data_ptr(), int(d.stride(0)), *chain_items, *fan_items, _st(d)), "gw_mlp_chain_backward")


# ---------------------------------------------------------------------------------------------------------------------
# backward of the Linear/ReLU chain shared by all fused ops
# ---------------------------------------------------------------------------------------------------------------------
class GatheredRows:
    """An incoming gradient that is a gather: row b * n + k = ``table[b * rows_pb + idx[k]]`` (+ row of ``add``) - the gradient
    of the edge rows of a block, from the gradient of the aggregate (index_select backward of scatter_sum, graph_net_block.py:188)
    and of e' where the block exposes it.  ``rows()`` materialises it (gw_gather_rows); the fused chain launch reads it in place."""

    def __init__(self, table: torch.Tensor, rows_pb: int, idx: torch.Tensor, batch: int, add: Optional[torch.Tensor]):
        self.table, self.rows_pb, self.idx, self.batch, self.add = table, rows_pb, idx, batch, add
        self.n = int(idx.numel())
        self._rows: Optional[torch.Tensor] = None

    def rows(self) -> torch.Tensor:
        if self._rows is None:
            self._rows = gather_rows(self.table, self.rows_pb, self.idx, self.batch, self.n, self.add)
        return self._rows


def _mlp_chain_backward(dout, saved: SavedActivations, weights: Sequence[torch.Tensor], has_norm: bool,
                        gamma: Optional[torch.Tensor], grads: List[Optional[torch.Tensor]], mlp=None, ln_width: int = 0,
                        fan: Sequence[Tuple[int, int]] = (), fan_out: Optional[dict] = None, bias0_by_caller: bool = False,
                        bias0_in_chain: bool = False, fan_add: Optional[dict] = None):
    """Backward through [LayerNorm] <- Linear_L <- ReLU <- ... <- Linear_1 <- ReLU, down to the output of Linear_0.

    ``weights`` = [W0, b0, W1, b1, ..., WL, bL, (gamma, beta)] (state_dict order of the reference ``MLP.model``);
    ``grads`` has the same length and is filled in place for W1..WL, every bias and gamma / beta.  W0's gradient
    depends on how the caller feeds layer 0 (concatenated operands, gathers, pre-multiplied tables) and is left to it.
    Returns (dz0 [rows, hidden] = gradient at the output of Linear_0, already masked by its ReLU, device).
    ``fan``: column blocks (lo, hi) of W0 whose input gradients dz0 @ W0[:, lo:hi] the caller wants; ``fan_out[(lo, hi)]``
    receives them (from the fused chain launch when the MLP has the kernel shapes, from single products otherwise).
    ``bias0_by_caller``: Linear_0's bias gradient (column sums of dz0) is NOT launched here: the caller's first weight-gradient
    GEMM on dz0 takes it along (``gemm_tn_acc(dz0, ..., colsum=grads[1])``; ``grads[1]`` is handed over zeroed) - one pass over
    dz0 less per MLP.  ``bias0_in_chain``: the caller has no such GEMM (every operand pre-multiplied): the fused chain launch
    sums the columns itself and ``fan_out["bias0_done"]`` is set; MLPs off the kernel shapes leave it to the caller's own pass.
    ``fan_add[(lo, hi)]``: rows the caller would add to that input gradient (the same tensor's gradient from another use: a
    block's residual; FAN_ADD_INPUT: ``dout`` itself); the fused launch adds them before the store and lists the block in
    ``fan_out["added"]``.  ``dout`` may be ``GatheredRows``: the fused launch with the LayerNorm prologue gathers it in place
    (``dout.rows()`` is then never made: the caller checks ``dout._rows``); every other path materialises it."""
    n_lin = (len(weights) - (2 if has_norm else 0)) // 2
    if n_lin < 2:
        raise RuntimeError("MLP needs at least one hidden layer")
    # every parameter gradient of this MLP is accumulated into (GEMMs with atomics, column sums): ONE zero fill for all of
    # them - views of a single buffer - instead of one launch per parameter; grads[0] (W0) is handed to the caller zeroed
    gathered = dout if isinstance(dout, GatheredRows) else None
    dev = gathered.table.device if gathered is not None else dout.device
    zbuf = torch.zeros(sum(int(w.numel()) for w in weights), dtype=torch.float32, device=dev)
    zs, off = [], 0
    for w in weights:
        zs.append(zbuf[off:off + int(w.numel())].view(w.shape))
        off += int(w.numel())
    grads[0] = zs[0]
    if gathered is not None:
        n_rows, width = gathered.batch * gathered.n, int(gathered.table.shape[1])
    else:
        dout = dout.contiguous()
        n_rows, width = int(dout.shape[0]), int(dout.shape[1])
    # Kernel-shaped MLPs (256 wide, at most two Linear layers above layer 0): the whole chain of masked input-gradient products
    # and the requested layer-0 blocks in ONE launch (gw_mlp_chain_backward); the weight-gradient GEMMs read what it stored.
    # The LayerNorm backward in front of the chain is that launch's prologue (gw_mlp_ln_chain_backward).
    pts = None
    if (mlp is not None and 2 <= n_lin <= 3 and width == 256 and n_rows > 0
            and all(saved.hidden[l].shape[1] == 256 and saved.hidden[l].stride(0) == 256 for l in range(n_lin - 1))):
        pts = [_packed_transposed(mlp, l, weights[2 * l], 0, int(weights[2 * l].shape[1])) for l in range(n_lin - 1, 0, -1)]
        # (fp32 streams: bwd_chain_kernel; split streams of the bf16x3 mode: bwd_chainx3_kernel - the same three-MFMA products)
        if any(p is None for p in pts):
            pts = None
    ln_fused = (pts is not None and has_norm and int(gamma.numel()) == 256 and ln_width in (0, 256)
                and saved.pre_norm.shape[1] == 256 and saved.pre_norm.stride(0) == 256)
    in_place = (gathered is not None and ln_fused and gathered.table.is_contiguous() and gathered.idx.dtype == torch.int32
                and (gathered.add is None or (gathered.add.stride(1) == 1 and gathered.add.stride(0) % 4 == 0)))
    if gathered is not None and not in_place:
        dout = gathered.rows()
    if has_norm:
        grads[-2] = zs[-2]
        grads[-1] = zs[-1]
        if ln_fused:
            d = torch.empty((n_rows, 256), dtype=torch.float32, device=dev)  # written by the chain launch
        else:
            d = layernorm_backward(dout, saved.pre_norm, gamma, grads[-2], grads[-1], ln_width)
    else:
        d = dout
    fused = None
    if pts is not None:
        fblk = [(tuple(blk), _packed_transposed(mlp, 0, weights[0], blk[0], blk[1])) for blk in (fan if fan_out is not None else ())]
        fblk = [(blk, ft) for blk, ft in fblk if ft is not None][:3]  # (other blocks: single products below)
        outs = [torch.empty((n_rows, 256), dtype=torch.float32, device=d.device) for _ in pts]
        fouts = [torch.empty((n_rows, 256), dtype=torch.float32, device=d.device) for _ in fblk]
        adds = [fan_add.get(blk) if fan_add else None for blk, _ in fblk]
        if not ln_fused:  # (the launch's input is then the LayerNorm backward's output, not dout: dout joins as rows)
            adds = [dout if t is FAN_ADD_INPUT else t for t in adds]
        in_chain = bias0_in_chain and fan_out is not None
        chain_backward(gathered.table if in_place else (dout if ln_fused else d),
                       [(pts[i], saved.hidden[n_lin - 2 - i], outs[i]) for i in range(len(pts))],
                       [(ft, t) for (_, ft), t in zip(fblk, fouts)],
                       ln=(saved.pre_norm, gamma, grads[-2], grads[-1], d) if ln_fused else None,
                       colsum=zs[1] if in_chain else None, fan_add=adds,
                       gather=(gathered.idx, gathered.rows_pb, gathered.add) if in_place else None)
        fused = outs
        for (blk, _), t in zip(fblk, fouts):
            fan_out[blk] = t
        if fan_out is not None:
            fan_out["added"] = [blk for (blk, _), t in zip(fblk, adds) if t is not None]
            if in_chain:
                fan_out["bias0_done"] = True
    if fused is not None:
        ds = [d] + fused  # ds[i]: gradient at the output of Linear_{n_lin-1-i}
        for i, l in enumerate(range(n_lin - 1, 0, -1)):
            gW, gb = zs[2 * l], zs[2 * l + 1]
            gemm_tn_acc(ds[i], saved.hidden[l - 1], gW, colsum=gb, x3=_x3(mlp))
            grads[2 * l], grads[2 * l + 1] = gW, gb
        d = ds[-1]
        grads[1] = zs[1]
        if not bias0_by_caller:
            relu_backward(d, None, grads[1])
        if fan_out is not None:
            for lo, hi in fan:
                if (lo, hi) not in fan_out:
                    fan_out[(lo, hi)] = input_grad(mlp, 0, d, weights[0], lo, hi)
        return d, d.device
    # Every Linear_l (l >= 1) gets its weight gradient d_l^T h_{l-1} and, from the same GEMM, its bias gradient (column
    # sums of d_l); the input gradient d_l W_l is masked by the ReLU that produced h_{l-1} in the product's epilogue.
    for l in range(n_lin - 1, 0, -1):
        W = weights[2 * l]
        h_prev = saved.hidden[l - 1]  # relu output feeding Linear_l, [rows, in_l]
        gW, gb = zs[2 * l], zs[2 * l + 1]
        gemm_tn_acc(d, h_prev, gW, colsum=gb, x3=_x3(mlp))
        grads[2 * l], grads[2 * l + 1] = gW, gb
        d = (input_grad(mlp, l, d, W, 0, int(W.shape[1]), relu_of=h_prev) if mlp is not None
             else relu_backward(gemm_nn(d, W, int(W.shape[1])), h_prev, None))
    # Linear_0's bias gradient: column sums of dz0 (its weight gradient is the caller's: it depends on the operands)
    grads[1] = zs[1]
    if not bias0_by_caller:
        relu_backward(d, None, grads[1])
    if fan_out is not None and mlp is not None:
        for lo, hi in fan:
            fan_out[(lo, hi)] = input_grad(mlp, 0, d, weights[0], lo, hi)
    return d, d.device


class MLPRowsFunction(torch.autograd.Function):
    """``MLP.forward`` on rows (graph_net_block.py:63-77) [+ a residual: ``out + start_features``, decoder.py:93]."""

    @staticmethod
    def forward(ctx, mlp, x2, res, res_rows_pb, res_k, n_rows, rows_per_batch, *params):
        pm = mlp.packed()
        save = SavedActivations(pm, n_rows, x2.device)
        residual_op = None if res is None else Operand(res, res_rows_pb, res_k)
        y = ops.mlp_forward(pm, Operand(x2, rows_per_batch, mlp.native_k()), n_rows, rows_per_batch, residual=residual_op, save=save)
        ctx.mlp, ctx.save, ctx.has_norm = mlp, save, pm.gamma is not None
        ctx.has_res = res is not None
        if res is not None and res.requires_grad and (res_rows_pb <= 0 or int(res.shape[0]) != n_rows):
            raise NotImplementedError("graph_weather_amd: gradient of a residual table shared by the batch is not implemented")
        ctx.save_for_backward(x2, res if res is not None else x2.new_zeros(0), *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, res, *params = ctx.saved_tensors
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        W0 = params[0]
        fan = [(0, int(W0.shape[1]))] if ctx.needs_input_grad[1] else []
        fo: dict = {}
        dz0, _ = _mlp_chain_backward(dy, ctx.save, params, ctx.has_norm, params[-2] if ctx.has_norm else None, grads, ctx.mlp,
                                     ctx.mlp.out_dim, fan=fan, fan_out=fo, bias0_by_caller=True)
        gW0 = grads[0]  # zeroed by _mlp_chain_backward
        gemm_tn_acc(dz0, x2, gW0, colsum=grads[1], x3=_x3(ctx.mlp))
        grads[0] = gW0
        dx = fo[fan[0]] if fan else None
        dres = None
        if ctx.has_res and ctx.needs_input_grad[2]:
            # y = MLP(x) + res[:, :n_out]: the identity term of d(out)/d(features) (multi-step rollouts feed the output back in)
            n_out = int(dy.shape[1])
            if res.shape[1] == n_out:
                dres = dy
            else:
                dres = torch.zeros_like(res)
                dres[:, :n_out] = dy
        return (None, dx, dres, None, None, None, None, *grads)


def mlp_rows(mlp, x2: torch.Tensor, n_rows: int, rows_per_batch: int, residual_op: Optional[Operand] = None) -> torch.Tensor:
    params = mlp.native_params()
    if residual_op is None:
        return MLPRowsFunction.apply(mlp, x2, None, 0, 0, n_rows, rows_per_batch, *params)
    return MLPRowsFunction.apply(mlp, x2, residual_op.tensor, residual_op.rows_per_batch, residual_op.k, n_rows, rows_per_batch,
                                 *params)


# ---------------------------------------------------------------------------------------------------------------------
# recomputation instead of saved activations (the reference's torch.utils.checkpoint calls: graph_net_block.py:73-74,
# 294-297; graphcast/model.py:212-285)
# ---------------------------------------------------------------------------------------------------------------------
class RecomputeFunction(torch.autograd.Function):
    """A segment of the model as ONE autograd node that keeps only its inputs: the forward runs the segment with grad mode
    off - i.e. on the inference kernels, which save no activations - and the backward re-runs it with grad mode on (the
    training kernels, with their activation saves, which live only until this node's backward returns) and backpropagates
    through that replay.  Gradients of the segment's parameters are returned through autograd like any other."""

    @staticmethod
    def forward(ctx, fn, n_in, params, *tensors):
        ctx.fn, ctx.n_in, ctx.params = fn, n_in, params
        ctx.save_for_backward(*tensors[:n_in])
        with torch.no_grad():
            outs = fn(*tensors[:n_in])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        n_in, params = ctx.n_in, ctx.params
        ins = [t.detach().requires_grad_(bool(need)) for t, need in zip(ctx.saved_tensors, ctx.needs_input_grad[3:3 + n_in])]
        with torch.enable_grad():
            outs = ctx.fn(*ins)
        pairs = [(o, d) for o, d in zip(outs, douts) if d is not None and o.requires_grad]
        wanted = [t for t in ins if t.requires_grad] + [p for p in params if p.requires_grad]
        got = iter(torch.autograd.grad([o for o, _ in pairs], wanted, [d for _, d in pairs], allow_unused=True)) if pairs and wanted \
            else iter(())
        g_in = [next(got, None) if t.requires_grad else None for t in ins]
        g_par = [next(got, None) if p.requires_grad else None for p in params]
        return (None, None, None, *g_in, *g_par)


def recompute(fn, inputs: Sequence[torch.Tensor], module) -> Tuple[torch.Tensor, ...]:
    """``fn(*inputs) -> tuple of tensors`` as a recomputed segment (see RecomputeFunction).  ``module`` names the parameters
    the segment uses (an ``nn.Module`` or a list of them).  Without anything to differentiate the segment just runs."""
    mods = module if isinstance(module, (list, tuple)) else [module]
    params = [p for m in mods for p in m.parameters()]
    if not torch.is_grad_enabled() or not (any(p.requires_grad for p in params) or any(t.requires_grad for t in inputs)):
        return tuple(fn(*inputs))
    return RecomputeFunction.apply(fn, len(inputs), params, *inputs, *params)


def _input_grad_joined(mlp, d: torch.Tensor, W: torch.Tensor, lo: int, hi: int, addend: Optional[torch.Tensor]) -> torch.Tensor:
    """d @ W[:, lo:hi] + addend: another gradient of the same tensor joins the product before it is stored (the chain launch's
    ``fan_add``; the addend may be the buffer of an earlier product - every lane reads its values before it writes them)."""
    if addend is None:
        return input_grad(mlp, 0, d, W, lo, hi)
    pt = _packed_transposed(mlp, 0, W, lo, hi) if (d.shape[1] == 256 and d.stride(0) % 4 == 0 and d.shape[0] > 0) else None
    if (pt is not None and addend.shape == d.shape and addend.stride(1) == 1 and addend.stride(0) % 4 == 0 and addend.stride(0) >= 256
            and addend.dtype == torch.float32):
        out = torch.empty((int(d.shape[0]), 256), dtype=torch.float32, device=d.device)
        chain_backward(d, [], [(pt, out)], fan_add=[addend])
        return out
    return input_grad(mlp, 0, d, W, lo, hi).add_(addend)


class ProjectFunction(torch.autograd.Function):
    """out_s = x @ W[:, lo_s:hi_s]^T for slices of a layer-1 weight (the layer-1 split of graph_net_block.py:131-134).
    ``passthrough``: x itself is one more output - for a caller that also uses x directly (the residual ``e' = MLP(..) + e`` of an
    encoder / decoder block, whose edge embedding e enters the edge MLP as the product We.e): the gradients of both uses arrive in
    ONE backward call and leave it as one tensor, instead of autograd adding two 464 MB tables (1 degree) in a pass of its own."""

    @staticmethod
    def forward(ctx, mlp, slice_ids, passthrough, x, n_rows, rows_per_batch, W):
        pm = mlp.packed()
        outs = ops.project_forward([pm.w1[s] for s in slice_ids], Operand(x, rows_per_batch, 256), n_rows, rows_per_batch)
        ctx.mlp, ctx.slice_ids, ctx.passthrough = mlp, slice_ids, passthrough
        ctx.save_for_backward(x, W)
        return (*outs, x) if passthrough else tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        x, W = ctx.saved_tensors
        gW = torch.zeros_like(W)
        dx = douts[len(ctx.slice_ids)] if ctx.passthrough else None  # gradient of the direct use of x
        if dx is not None and not ctx.needs_input_grad[3]:
            dx = None
        for s, d in zip(ctx.slice_ids, douts):
            if d is None:
                continue
            lo, hi = ctx.mlp.native_splits()[s]
            d = d.contiguous()
            gemm_tn_acc(d, x, gW, c_col0=lo, x3=_x3(ctx.mlp))  # dW[:, lo:hi] += d^T x
            if ctx.needs_input_grad[3]:
                dx = _input_grad_joined(ctx.mlp, d, W, lo, hi, dx)
        return None, None, None, dx, None, None, gW


def project(mlp, slice_ids: Sequence[int], x: torch.Tensor, n_rows: int, rows_per_batch: int, passthrough: bool = False) -> Tuple[torch.Tensor, ...]:
    """-> the products; with ``passthrough`` followed by x as an output of the same autograd node (see ProjectFunction)."""
    return ProjectFunction.apply(mlp, tuple(slice_ids), bool(passthrough), x, n_rows, rows_per_batch, mlp.native_params()[0])


# ---------------------------------------------------------------------------------------------------------------------
# message-passing ops
# ---------------------------------------------------------------------------------------------------------------------
class OperandSpec:
    """How an operand of a fused op is fed: mode in {"zero", "raw", "proj"}; rows_pb 0 = table shared by the batch."""

    def __init__(self, mode: str, rows_pb: int = 0):
        self.mode, self.rows_pb = mode, rows_pb


def _scatter_rows(rows: torch.Tensor, kind: int, plan, batch: int, table_rows_pb: int, n_table_rows: int) -> torch.Tensor:
    """Gradient of a row table that was read through the plan's index of ``kind`` (0: src, 1: dst, 2: edge id):
    sums the per-edge rows [batch * E, 256] into [batch or 1, table rows, 256]."""
    E = plan.num_edges
    batch_out = batch if table_rows_pb > 0 else 1
    if kind == 1:
        return segment_sum_rows(rows, E, batch, batch_out, n_table_rows, plan.dst_ptr(), None)
    if kind == 0:
        perm, ptr = plan.src_sorted()
        return segment_sum_rows(rows, E, batch, batch_out, n_table_rows, ptr, perm)
    if batch_out == batch:
        return rows
    return segment_sum_rows(rows, E, batch, 1, E, plan.identity_ptr(), None)


class EdgeUpdateFunction(torch.autograd.Function):
    """``EdgeProcessor.forward`` + ``scatter_sum`` (graph_net_block.py:131-137, :188) on a destination-sorted shared graph."""

    @staticmethod
    def forward(ctx, mlp, plan, batch, specs, want_edges, x_src, x_dst, e_in, e_res, e_res_rows_pb, *params):
        pm = mlp.packed()
        E, n_dst = plan.num_edges, plan.n_dst
        dev = e_res.device
        tensors = (x_src, x_dst, e_in)
        opnds = []
        for t, sp in zip(tensors, specs):
            opnds.append(ops.ZERO if sp.mode == "zero" else Operand(t, sp.rows_pb, 256, projected=(sp.mode == "proj")))
        agg = torch.zeros((batch * n_dst, 256), dtype=torch.float32, device=dev)
        e_out = torch.empty((batch * E, 256), dtype=torch.float32, device=dev) if want_edges else None
        save = SavedActivations(pm, batch * E, dev)
        ops.edge_update_forward(pm, batch, plan.src, plan.dst, opnds[0], opnds[1], opnds[2], Operand(e_res, e_res_rows_pb, 256),
                                n_dst, agg, e_out, save=save)
        ctx.mlp, ctx.plan, ctx.batch, ctx.specs, ctx.save, ctx.e_res_rows_pb = mlp, plan, batch, specs, save, e_res_rows_pb
        ctx.want_edges = want_edges
        ctx.same_e = e_in is e_res  # (one tensor in two argument slots: its two gradients may be returned as one)
        ctx.save_for_backward(x_src, x_dst, e_in, *params)
        if want_edges:
            return agg, e_out
        return agg, torch.empty(0, device=dev)

    @staticmethod
    def backward(ctx, dagg, de_out):
        x_src, x_dst, e_in, *params = ctx.saved_tensors
        plan, B, specs, mlp = ctx.plan, ctx.batch, ctx.specs, ctx.mlp
        E = plan.num_edges
        # gradient at e' = LN(..) + e_res: from the aggregate (gather by destination) and, if exposed, from e_out
        add = de_out.contiguous() if (ctx.want_edges and de_out is not None and de_out.numel()) else None
        dagg = dagg.contiguous()
        # (a gather: the fused chain launch reads it in place and the [B * E, 256] table exists only if something else needs it)
        dn = GatheredRows(dagg, plan.n_dst, plan.dst, B, add)
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        has_norm = mlp._norm() is not None
        fan = [tuple(mlp.native_splits()[i]) for i, sp in enumerate(specs) if sp.mode == "raw" and ctx.needs_input_grad[5 + i]]
        fo: dict = {}
        no_gemm_on_dz0 = not any(sp.mode == "raw" for sp in specs)
        # e' = MLP(..., e) + e with e a per-sample table: the gradient of e from the residual (dn) joins the one from the operand
        same_e = (ctx.same_e and specs[2].mode == "raw" and specs[2].rows_pb > 0 and ctx.e_res_rows_pb > 0
                  and ctx.needs_input_grad[7] and ctx.needs_input_grad[8])
        e_blk = tuple(mlp.native_splits()[2])
        dz0, _ = _mlp_chain_backward(dn, ctx.save, params, has_norm, params[-2] if has_norm else None, grads, mlp, mlp.out_dim,
                                     fan=fan, fan_out=fo, bias0_by_caller=True, bias0_in_chain=no_gemm_on_dz0,
                                     fan_add={e_blk: FAN_ADD_INPUT} if same_e else None)
        e_joined = same_e and e_blk in fo.get("added", ())
        W0 = params[0]
        gW0 = grads[0]  # zeroed by _mlp_chain_backward
        # bias gradient of Linear_0: rides on the first weight-gradient GEMM over dz0 (None once taken, or summed by the chain launch)
        gb0 = None if fo.get("bias0_done") else grads[1]
        tensors = (x_src, x_dst, e_in)
        n_rows_tab = (plan.n_src, plan.n_dst, E)
        dts: List[Optional[torch.Tensor]] = [None, None, None]
        for i, (t, sp) in enumerate(zip(tensors, specs)):
            if sp.mode == "zero":
                continue
            lo, hi = mlp.native_splits()[i]
            if sp.mode == "proj":
                if ctx.needs_input_grad[5 + i]:
                    dts[i] = _scatter_rows(dz0, i, plan, B, sp.rows_pb, n_rows_tab[i])
            else:  # raw rows multiplied by W0[:, lo:hi]
                idx = (plan.src, plan.dst, None)[i]
                g = t if (idx is None and sp.rows_pb > 0) else gather_rows(t, sp.rows_pb, idx, B, E)
                gemm_tn_acc(dz0, g, gW0, c_col0=lo, colsum=gb0, x3=_x3(mlp))
                gb0 = None
                if ctx.needs_input_grad[5 + i]:
                    dg = fo[(lo, hi)]
                    dts[i] = _scatter_rows(dg, i, plan, B, sp.rows_pb, n_rows_tab[i])
        if gb0 is not None:  # every operand pre-multiplied: no GEMM over dz0 here
            relu_backward(dz0, None, gb0)
        grads[0] = gW0
        de_res = None
        if ctx.needs_input_grad[8] and not e_joined:  # (joined: dts[2] already carries dn - e_in and e_res are one tensor)
            if ctx.e_res_rows_pb > 0:
                de_res = dn.rows()
            elif dn._rows is None and add is None:
                # a residual table shared by the batch, and nothing has made the per-sample rows: sum_b dagg[b, dst[e]] is the
                # gather of the batch sum of the (small) aggregate gradient - not a pass over a [B * E, 256] table
                ident = getattr(plan, "_ident_dst", None)
                if ident is None or ident.device != dagg.device:
                    ident = plan._ident_dst = torch.arange(plan.n_dst + 1, dtype=torch.int32, device=dagg.device)
                de_res = gather_rows(segment_sum_rows(dagg, plan.n_dst, B, 1, plan.n_dst, ident, None), plan.n_dst, plan.dst, 1, E)
            else:
                de_res = segment_sum_rows(dn.rows(), E, B, 1, E, plan.identity_ptr(), None)
        return (None, None, None, None, None, dts[0], dts[1], dts[2], de_res, None, *grads)


def edge_update(mlp, plan, batch: int, specs, want_edges: bool, x_src, x_dst, e_in, e_res, e_res_rows_pb: int):
    params = mlp.native_params()
    dummy = e_res.new_zeros(0)
    ts = [t if t is not None else dummy for t in (x_src, x_dst, e_in)]
    agg, e_out = EdgeUpdateFunction.apply(mlp, plan, batch, tuple(specs), want_edges, ts[0], ts[1], ts[2], e_res, e_res_rows_pb, *params)
    return agg, (e_out if want_edges else None)


class NodeUpdateFunction(torch.autograd.Function):
    """``NodeProcessor.forward`` after the aggregation (graph_net_block.py:189-191)."""

    @staticmethod
    def forward(ctx, mlp, n_rows, rows_per_batch, x_spec, res_rows_pb, x, x_res, agg, *params):
        pm = mlp.packed()
        xo = ops.ZERO if x_spec.mode == "zero" else Operand(x, x_spec.rows_pb, 256, projected=(x_spec.mode == "proj"))
        ro = ops.ZERO if x_res.numel() == 0 else Operand(x_res, res_rows_pb, 256)
        save = SavedActivations(pm, n_rows, agg.device)
        out = ops.node_update_forward(pm, n_rows, rows_per_batch, xo, ro, Operand(agg, rows_per_batch, 256), save=save)
        ctx.mlp, ctx.n_rows, ctx.rows_per_batch, ctx.x_spec, ctx.res_rows_pb, ctx.save = mlp, n_rows, rows_per_batch, x_spec, res_rows_pb, save
        ctx.has_res = x_res.numel() != 0
        ctx.save_for_backward(x, agg, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, agg, *params = ctx.saved_tensors
        mlp, n, rpb = ctx.mlp, ctx.n_rows, ctx.rows_per_batch
        batch = n // rpb
        dout = dout.contiguous()
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        has_norm = ctx.mlp._norm() is not None
        (xlo, xhi), (alo, ahi) = mlp.native_splits()
        sp = ctx.x_spec
        fan = ([(alo, ahi)] if ctx.needs_input_grad[7] else []) + ([(xlo, xhi)] if (sp.mode == "raw" and ctx.needs_input_grad[5]) else [])
        fo: dict = {}
        dz0, _ = _mlp_chain_backward(dout, ctx.save, params, has_norm, params[-2] if has_norm else None, grads, ctx.mlp, ctx.mlp.out_dim,
                                     fan=fan, fan_out=fo, bias0_by_caller=True)
        W0 = params[0]
        gW0 = grads[0]  # zeroed by _mlp_chain_backward
        gemm_tn_acc(dz0, agg, gW0, c_col0=alo, colsum=grads[1], x3=_x3(ctx.mlp))
        dagg = fo[(alo, ahi)] if ctx.needs_input_grad[7] else None
        dx = None

        def over_batch(rows):  # gradient of a table shared by the batch: sum the per-sample rows
            ident = torch.arange(rpb + 1, dtype=torch.int32, device=rows.device)
            return segment_sum_rows(rows, rpb, batch, 1, rpb, ident, None)

        if sp.mode == "raw":
            xg = x if sp.rows_pb > 0 else gather_rows(x, 0, None, batch, rpb)
            gemm_tn_acc(dz0, xg, gW0, c_col0=xlo, x3=_x3(ctx.mlp))
            if ctx.needs_input_grad[5]:
                dx = fo[(xlo, xhi)]
                if sp.rows_pb == 0:
                    dx = over_batch(dx)
        elif sp.mode == "proj" and ctx.needs_input_grad[5]:
            dx = dz0 if sp.rows_pb > 0 else over_batch(dz0)
        grads[0] = gW0
        dres = None
        if ctx.has_res and ctx.needs_input_grad[6]:
            dres = dout if ctx.res_rows_pb > 0 else over_batch(dout)
        return (None, None, None, None, None, dx, dres, dagg, *grads)


def node_update(mlp, n_rows: int, rows_per_batch: int, x_spec: OperandSpec, x, x_res, res_rows_pb: int, agg):
    params = mlp.native_params()
    dummy = agg.new_zeros(0)
    return NodeUpdateFunction.apply(mlp, n_rows, rows_per_batch, x_spec, res_rows_pb, x if x is not None else dummy,
                                    x_res if x_res is not None else dummy, agg, *params)


class NormalizedMSEFunction(torch.autograd.Function):
    """``NormalizedMSELoss.forward`` (losses.py:66-94)."""

    @staticmethod
    def forward(ctx, pred, target, lat_weights, inv_var):
        ctx.save_for_backward(pred, target, lat_weights, inv_var if inv_var is not None else pred.new_zeros(0))
        ctx.has_var = inv_var is not None
        return ops.normalized_mse_forward(pred, target, lat_weights, inv_var)

    @staticmethod
    def backward(ctx, dloss):
        pred, target, w, iv = ctx.saved_tensors
        b, c = int(pred.shape[0]), int(pred.shape[-1])
        nodes = pred.numel() // (b * c)
        dpred = torch.empty_like(pred)
        dl = dloss.reshape(1).contiguous().float()
        _lib.check(_L().gw_normalized_mse_backward(pred.data_ptr(), target.data_ptr(), iv.data_ptr() if ctx.has_var else None,
                                                   1 if (ctx.has_var and iv.numel() == pred.numel()) else 0, w.data_ptr(), int(w.numel()), b, nodes, c, dl.data_ptr(), dpred.data_ptr(),
                                                   _st(pred)), "gw_normalized_mse_backward")
        return dpred, None, None, None
