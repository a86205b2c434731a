"""Tensor-level wrappers over the C ABI (``include/gw_amd.h``).  PyTorch is used only to own device memory
and to name the current HIP stream; every arithmetic operation of the hot path happens inside libgw_amd.so."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GwActivationSave, GwMlpWeights, GwOperand


class KernelTimer:
    """Optional HIP-event bracket around tagged C-ABI calls (used by bench.py for the live roofline figure).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self, tags):
        self.tags = set(tags)
        self.events = {t: [] for t in self.tags}
        self.units = {t: [] for t in self.tags}  # batch elements each timed launch processed (per-sample streams: 1)

    def start(self, tag):
        if tag not in self.tags:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def stop(self, tag, ev, units: int = 1):
        if ev is not None:
            ev[1].record()
            self.events[tag].append(ev)
            self.units[tag].append(int(units))

    def mean_units(self, tag) -> float:
        u = self.units[tag]
        return sum(u) / max(1, len(u))

    def mean_ms(self, tag) -> float:
        evs = self.events[tag]
        return sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))


TIMER: Optional[KernelTimer] = None


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device_of(t: torch.Tensor):
    """Context that makes ``t``'s GPU the current HIP device for the C-ABI call inside it: kernels are launched on the
    stream of ``t.device`` and a launch on a stream of a non-current device fails.  Free when it already is current."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


def _require(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"graph_weather_amd: {name} must live on a HIP device (no CPU path exists)")
    if t.dtype != dtype:
        raise RuntimeError(f"graph_weather_amd: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"graph_weather_amd: {name} must be contiguous")


@dataclass
class Operand:
    """A row table feeding an MLP input slice.  ``tensor`` is [rows, ld]; ``rows_per_batch`` = 0 means the table
    is shared by all batch elements; ``index`` (int32) maps a column to a row within its batch element."""

    tensor: Optional[torch.Tensor]
    rows_per_batch: int
    k: int
    index: Optional[torch.Tensor] = None
    projected: bool = False  # rows are X . W1_slice^T (see project_forward): gather-added, no MFMA pass
    tiles: bool = False  # ``tensor`` is a byte buffer of bf16 edge tiles (include/gw_amd.h: GW_LAYOUT_EDGE_TILES_BF16)

    def c(self) -> GwOperand:
        if self.k == 0 or self.tensor is None:
            return GwOperand(None, None, 0, 0, 0, 0, 0)
        if self.tiles:
            _require(self.tensor, "edge tiles", torch.uint8)
            # rows_per_batch: n_edges = one tile set per batch element, 0 = one set shared by the batch
            return GwOperand(self.tensor.data_ptr(), None, int(self.rows_per_batch), 256, 256, 0, _lib.LAYOUT_EDGE_TILES_BF16)
        if self.tensor.dtype == torch.bfloat16:  # an aggregate as bf16 rows in K order (include/gw_amd.h: GW_LAYOUT_ROWS_BF16K)
            if self.projected or self.index is not None or self.k != 256:
                raise RuntimeError("graph_weather_amd: bf16 rows (K order) are a format of raw 256-wide aggregates only")
            _require(self.tensor, "operand", torch.bfloat16)
            return GwOperand(self.tensor.data_ptr(), None, int(self.rows_per_batch), int(self.tensor.stride(0)), 256, 0,
                             _lib.LAYOUT_ROWS_BF16K)
        if self.tensor.dtype == torch.float16:  # layer-1 node products as fp16 rows (include/gw_amd.h: GW_LAYOUT_ROWS_F16)
            if not self.projected or self.index is not None:
                raise RuntimeError("graph_weather_amd: fp16 rows are a format of projected (layer-1 product) operands only")
            _require(self.tensor, "operand", torch.float16)
            return GwOperand(self.tensor.data_ptr(), None, int(self.rows_per_batch), int(self.tensor.stride(0)), int(self.k), 1,
                             _lib.LAYOUT_ROWS_F16)
        _require(self.tensor, "operand")
        if self.index is not None:
            _require(self.index, "operand index", torch.int32)
        return GwOperand(self.tensor.data_ptr(), None if self.index is None else self.index.data_ptr(),
                         int(self.rows_per_batch), int(self.tensor.stride(0)), int(self.k), 1 if self.projected else 0, 0)


ZERO = Operand(None, 0, 0)


class SavedActivations:
    """Buffers a forward call fills for its backward (include/gw_amd.h: gw_activation_save): the relu outputs of
    every Linear but the last, and the pre-LayerNorm output."""

    def __init__(self, pm: "PackedMLP", n_rows: int, device):
        self.hidden = torch.empty((pm.n_mid + 1, n_rows, pm.hidden), dtype=torch.float32, device=device)
        # heads (n_out <= 80) are saved as zero-padded 80-column rows (the 5-tile kernel variant stores whole tiles)
        width = pm.n_out if pm.n_out == 256 else 80
        self.pre_norm = torch.empty((n_rows, width), dtype=torch.float32, device=device) if pm.gamma is not None else None

    def c(self) -> GwActivationSave:
        return GwActivationSave(self.hidden.data_ptr(), int(self.hidden.stride(0)), int(self.hidden.stride(1)),
                                None if self.pre_norm is None else self.pre_norm.data_ptr())


def pack_many(weight_dtype: int, mats, vecs, stream: int) -> None:
    """gw_pack_many over any number of items.  mats: (w_ptr, stride_f, stride_k, n_out, kseg, out_ptr[, rows]); vecs: (v_ptr, n,
    out_ptr[, n_out]) - rows / n_out: size of the packed item when it is larger than the source (zero rows of an output head)."""
    L = _lib.lib()
    M = _lib.PACK_MAX_ITEMS
    mi = vi = 0
    while mi < len(mats) or vi < len(vecs):
        mc, vc = mats[mi:mi + M], vecs[vi:vi + M]
        ma = (_lib.GwPackItem * max(len(mc), 1))()
        for k, it in enumerate(mc):
            w, sf, sk, n_out, kseg, out = it[:6]
            ma[k] = _lib.GwPackItem(w, sf, sk, n_out, kseg, it[6] if len(it) > 6 else 0, 0, out)
        va = (_lib.GwPadItem * max(len(vc), 1))()
        for k, it in enumerate(vc):
            va[k] = _lib.GwPadItem(it[0], it[1], it[3] if len(it) > 3 else 0, it[2])
        _lib.check(L.gw_pack_many(weight_dtype, len(mc), ma, len(vc), va, stream), "gw_pack_many")
        mi += len(mc)
        vi += len(vc)


BF16X3 = "bf16x3"
"""Compute-dtype name of the split-operand mode (include/gw_amd.h: GW_DTYPE_BF16X3): every matrix product as three bf16 MFMAs on
hi / lo operand pairs with fp32 accumulation - the reference's fp32 results to ~1e-5 at 5.3 x the fp32 matrix rate.  torch has no
dtype for it, hence a name; ``set_compute_dtype(model, "bf16x3")``."""

# torch dtype of a packed weight stream per GW_DTYPE_* (the split stream is typed int16: 2-byte elements no arithmetic is done on,
# and distinct from the bf16 stream, so the dtype of a packed slice names its format)
_STREAM_TORCH_DTYPE = {_lib.DTYPE_F32: torch.float32, _lib.DTYPE_BF16: torch.bfloat16, _lib.DTYPE_BF16X3: torch.int16}
_STREAM_GW_DTYPE = {v: k for k, v in _STREAM_TORCH_DTYPE.items()}


def gw_dtype_of(compute_dtype) -> int:
    if compute_dtype == torch.float32:
        return _lib.DTYPE_F32
    if compute_dtype == torch.bfloat16:
        return _lib.DTYPE_BF16
    if isinstance(compute_dtype, str) and compute_dtype.lower() in (BF16X3, "bf16x3", "split"):
        return _lib.DTYPE_BF16X3
    raise RuntimeError("graph_weather_amd: compute dtype must be torch.float32, torch.bfloat16 or \"bf16x3\"")


class PackedMLP:
    """Device-resident packed form of one reference ``MLP`` (graph_net_block.py:45-61)."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                 ln: Optional[Tuple[torch.Tensor, torch.Tensor]], splits: Sequence[Tuple[int, int]],
                 compute_dtype=torch.float32, ln_width: int = 0):
        L = _lib.lib()
        self.weight_dtype = gw_dtype_of(compute_dtype)
        n_lin = len(weights)
        if n_lin < 2:
            raise RuntimeError("graph_weather_amd: MLP needs at least one hidden layer")
        dev = weights[0].device
        self.hidden = int(weights[0].shape[0])
        self.n_out = int(weights[-1].shape[0])
        self.n_mid = n_lin - 2
        self.in_dim = int(weights[0].shape[1])
        self.ln_width = int(ln_width) if (ln is not None and 0 < ln_width < self.n_out) else 0  # 0: all n_out features
        for w in weights[1:-1]:
            if tuple(w.shape) != (self.hidden, self.hidden):
                raise RuntimeError("graph_weather_amd: hidden layers must be square")
        st = torch.cuda.current_stream(dev).cuda_stream
        guard = on_device_of(weights[0])
        guard.__enter__()  # every packing launch below runs with the weights' GPU current (released at the end of __init__)

        # Every slice and vector of the MLP goes through ONE gw_pack_many launch (27 MLPs x ~10 packing launches per weight
        # version otherwise: the bulk of a forward right after an optimizer step).  Items address the parameters in place.
        wd = self.weight_dtype
        mats, vecs, keep = [], [], []

        def src(t):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.contiguous().float()
            keep.append(t)
            return t

        def out_for(n_out, kseg):
            if wd == _lib.DTYPE_BF16:
                return torch.empty(L.gw_packed_bytes_bf16(n_out, 0, kseg) // 2, dtype=torch.bfloat16, device=dev)
            if wd == _lib.DTYPE_BF16X3:
                return torch.empty(L.gw_packed_bytes_bf16x3(n_out, 0, kseg) // 2, dtype=torch.int16, device=dev)
            return torch.empty(L.gw_packed_floats(n_out, 0, kseg), dtype=torch.float32, device=dev)

        def pack(w, k_lo, k_hi, out=None, rows=0):
            w = src(w)
            n_out, k_total = int(w.shape[0]), int(w.shape[1])
            if out is None:
                out = out_for(max(n_out, rows), k_hi - k_lo)
            mats.append((w.data_ptr() + 4 * k_lo, k_total, 1, n_out, k_hi - k_lo, out.data_ptr(), rows))
            return out

        def pad(v, out=None, n_out=0):
            v = src(v)
            if out is None:
                out = torch.empty(L.gw_padded_n(max(int(v.shape[0]), n_out)), dtype=torch.float32, device=dev)
            vecs.append((v.data_ptr(), int(v.shape[0]), out.data_ptr(), n_out))
            return out

        self.w1 = [pack(weights[0], lo, hi) for lo, hi in splits]
        self.b1 = pad(biases[0])
        if self.n_mid:
            one = out_for(self.hidden, self.hidden)
            nb = int(L.gw_padded_n(self.hidden))
            self.w_mid = torch.empty(self.n_mid * one.numel(), dtype=one.dtype, device=dev)
            self.b_mid = torch.empty(self.n_mid * nb, dtype=torch.float32, device=dev)
            for l, (w, b) in enumerate(zip(weights[1:-1], biases[1:-1])):
                pack(w, 0, self.hidden, out=self.w_mid[l * one.numel():(l + 1) * one.numel()])
                pad(b, out=self.b_mid[l * nb:(l + 1) * nb])
        else:
            self.w_mid = self.b_mid = None
        # output heads (n_out < 80) run on a fixed 5-tile (80 row) kernel variant: the last Linear, its bias and a LayerNorm on
        # the head are packed as 80 rows (rows n_out..79 zero; n_out itself still masks the stores)
        head = 80 if self.n_out < 80 else 0
        self.k_in, self.out_rows = int(weights[0].shape[1]), head  # (what fixed-size consumers check: include/gw_amd.h v15)
        self.w_out = pack(weights[-1], 0, self.hidden, rows=head)
        self.b_out = pad(biases[-1], n_out=head)
        self.gamma = pad(ln[0], n_out=head) if ln is not None else None
        self.beta = pad(ln[1], n_out=head) if ln is not None else None
        pack_many(self.weight_dtype, mats, vecs, st)
        del keep
        guard.__exit__(None, None, None)

    def c(self, active: Sequence[bool] = (True, True, True)) -> GwMlpWeights:
        w = GwMlpWeights()
        for i in range(3):
            w.w1[i] = self.w1[i].data_ptr() if i < len(self.w1) and active[i] else None
        w.b1 = self.b1.data_ptr()
        w.w_mid = self.w_mid.data_ptr() if self.w_mid is not None else None
        w.b_mid = self.b_mid.data_ptr() if self.b_mid is not None else None
        w.w_out = self.w_out.data_ptr()
        w.b_out = self.b_out.data_ptr()
        w.ln_gamma = self.gamma.data_ptr() if self.gamma is not None else None
        w.ln_beta = self.beta.data_ptr() if self.beta is not None else None
        w.hidden, w.n_mid, w.n_out = self.hidden, self.n_mid, self.n_out
        w.weight_dtype = self.weight_dtype
        w.ln_width = self.ln_width
        w.k_in, w.out_rows = self.k_in, self.out_rows
        return w


def mlp_forward(pm: PackedMLP, x: Operand, n_rows: int, rows_per_batch: int, residual: Optional[Operand] = None,
                out: Optional[torch.Tensor] = None, save: Optional[SavedActivations] = None) -> torch.Tensor:
    """graph_net_block.py:63-77 on ``n_rows`` rows (+ optional residual add)."""
    dev = x.tensor.device
    if out is None:
        out = torch.empty((n_rows, pm.n_out), dtype=torch.float32, device=dev)
    _require(out, "out")
    xc = x.c()
    wc = pm.c()
    rc = residual.c() if residual is not None else None
    if rc is not None:
        rc.k = pm.n_out
    with on_device_of(out):
        _lib.check(_lib.lib().gw_mlp_forward(n_rows, max(1, rows_per_batch), xc, wc, rc, out.data_ptr(), int(out.stride(0)),
                                             None if save is None else save.c(), _stream(out)), "gw_mlp_forward")
    return out


def mlp_post_forward(pm: PackedMLP, x: Operand, n_rows: int, rows_per_batch: int, post_w: Sequence[torch.Tensor],
                     post_half: bool = False, want_out: bool = False, out: Optional[torch.Tensor] = None,
                     post_out: Optional[Sequence[torch.Tensor]] = None):
    """graph_net_block.py:63-77 on ``n_rows`` rows, then the products of the output rows with packed [256, 256] slices in the same
    launch (include/gw_amd.h: gw_mlp_post_forward).  Returns (out or None, [products]); ``out`` / ``post_out``: caller's buffers."""
    import ctypes

    dev = x.tensor.device
    if out is None and want_out:
        out = torch.empty((n_rows, 256), dtype=torch.float32, device=dev)
    n_post = len(post_w)
    outs = (list(post_out) if post_out is not None else
            [torch.empty((n_rows, 256), dtype=torch.float16 if post_half else torch.float32, device=dev) for _ in range(n_post)])
    for o_ in outs:
        _require(o_, "post product rows", torch.float16 if post_half else torch.float32)
    wp = (ctypes.c_void_p * n_post)(*[w_.data_ptr() for w_ in post_w])
    op = (ctypes.c_void_p * n_post)(*[o_.data_ptr() for o_ in outs])
    with on_device_of(outs[0]):
        _lib.check(_lib.lib().gw_mlp_post_forward(n_rows, max(1, rows_per_batch), x.c(), pm.c(), None if out is None else out.data_ptr(),
                                                  256, n_post, wp, op, _lib.LAYOUT_ROWS_F16 if post_half else _lib.LAYOUT_ROWS_F32,
                                                  _stream(outs[0])), "gw_mlp_post_forward")
    return out, outs


def project_forward(w_slices: Sequence[torch.Tensor], x: Operand, n_rows: int, rows_per_batch: int,
                    weight_dtype: Optional[int] = None, relu_mask: Optional[torch.Tensor] = None,
                    zero_rows: Optional[torch.Tensor] = None, out_half: bool = False) -> List[torch.Tensor]:
    """out_s = x . W_s^T for up to four packed [256, 256] layer-1 slices in one launch (layer-1 split of
    graph_net_block.py:131-134 / :189: products over node tables are shared by all incident edges)."""
    import ctypes

    dev = x.tensor.device
    n = len(w_slices)
    outs = [torch.empty((n_rows, 256), dtype=torch.float16 if out_half else torch.float32, device=dev) for _ in range(n)]
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in w_slices])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    if weight_dtype is None:
        weight_dtype = _STREAM_GW_DTYPE[w_slices[0].dtype]
    with on_device_of(outs[0]):
        _lib.check(_lib.lib().gw_project_forward(n_rows, max(1, rows_per_batch), x.c(), n, wp, op, 256,
                                                 _lib.LAYOUT_ROWS_F16 if out_half else _lib.LAYOUT_ROWS_F32, weight_dtype,
                                                 None if relu_mask is None else relu_mask.data_ptr(),
                                                 None if zero_rows is None else zero_rows.data_ptr(), _stream(outs[0])),
                   "gw_project_forward")
    return outs


def edge_tiles_bytes(batch: int, n_edges: int) -> int:
    return int(_lib.lib().gw_edge_tiles_bytes(batch, n_edges))


def edge_rows_to_tiles(rows: torch.Tensor, batch: int, n_edges: int, rows_per_batch: int) -> torch.Tensor:
    """fp32 edge rows [batch * n_edges (or n_edges when shared), >= 256] -> bf16 edge tiles (byte buffer)."""
    _require(rows, "edge rows")
    tiles = torch.empty(edge_tiles_bytes(batch, n_edges), dtype=torch.uint8, device=rows.device)
    with on_device_of(rows):
        _lib.check(_lib.lib().gw_edge_rows_to_tiles(batch, n_edges, rows.data_ptr(), rows_per_batch, int(rows.stride(0)),
                                                    tiles.data_ptr(), _stream(rows)), "gw_edge_rows_to_tiles")
    return tiles


def edge_update_forward(pm: PackedMLP, batch: int, src: torch.Tensor, dst: torch.Tensor, x_src: Operand, x_dst: Operand,
                        e_in: Operand, e_res: Operand, n_dst: int, agg: torch.Tensor, e_out: Optional[torch.Tensor],
                        tag: Optional[str] = None, save: Optional[SavedActivations] = None, deterministic: bool = False,
                        segment_tiles: bool = False, segment_split: bool = False) -> None:
    """graph_net_block.py:131-137 (EdgeProcessor) fused with the scatter_sum of :188.  ``agg`` must be zeroed.
    ``e_out``: None, fp32 rows [batch * n_edges, 256], or a uint8 buffer of ``edge_tiles_bytes`` (bf16 edge tiles).
    ``deterministic``: bitwise reproducible segment sums (carry records + a fix-up launch instead of atomics).
    ``segment_tiles``: ``src`` / ``dst`` are the padded arrays of segment-aligned tiles (include/gw_amd.h:
    GW_EDGE_SEGMENT_TILES; ``GraphPlan.seg_tiles()``): ``agg`` rows are written with plain stores and need no zero fill when every
    destination has an edge; a bfloat16 ``agg`` is written as bf16 rows in K order (GW_EDGE_AGG_BF16K).  ``segment_split``: the
    padded list splits runs longer than a tile over several tiles (``SegTiles.split``; zero-filled fp32 ``agg``)."""
    _require(src, "src", torch.int32)
    _require(dst, "dst", torch.int32)
    agg_bf16 = agg.dtype == torch.bfloat16
    if agg_bf16 and not segment_tiles:
        raise RuntimeError("graph_weather_amd: a bf16 aggregate comes with segment-aligned tiles")
    _require(agg, "agg", torch.bfloat16 if agg_bf16 else torch.float32)
    e_out_layout = _lib.LAYOUT_ROWS_F32
    if e_out is not None:
        if e_out.dtype == torch.uint8:
            _require(e_out, "e_out tiles", torch.uint8)
            e_out_layout = _lib.LAYOUT_EDGE_TILES_BF16
        else:
            _require(e_out, "e_out")
    n_edges = int(src.shape[0])
    wc = pm.c((x_src.k > 0 and not x_src.projected, x_dst.k > 0 and not x_dst.projected, e_in.k > 0 and not e_in.projected))
    xs, xd, ei = x_src.c(), x_dst.c(), e_in.c()
    flags = _lib.EDGE_DETERMINISTIC if deterministic else 0
    if segment_tiles:
        flags |= _lib.EDGE_SEGMENT_TILES | (_lib.EDGE_AGG_BF16K if agg_bf16 else 0) | (_lib.EDGE_SEGMENT_SPLIT if segment_split else 0)
    ws, ws_bytes = None, 0
    if save is None:  # scratch for the kernel the library would like to use (the library never allocates)
        ws_bytes = int(_lib.lib().gw_edge_update_workspace_bytes(batch, n_edges, xs, xd, ei, wc, flags))
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=agg.device)
    ev = TIMER.start(tag) if TIMER is not None else None
    with on_device_of(agg):
        _lib.check(_lib.lib().gw_edge_update_forward(batch, n_edges, src.data_ptr(), dst.data_ptr(), xs, xd, ei,
                                                     e_res.c(), wc, None if e_out is None else e_out.data_ptr(), e_out_layout,
                                                     agg.data_ptr(), n_dst, None if save is None else save.c(),
                                                     None if ws is None else ws.data_ptr(), ws_bytes, flags, _stream(agg)),
                   "gw_edge_update_forward")
    if ev is not None:
        TIMER.stop(tag, ev, batch)


def node_update_forward(pm: PackedMLP, n_rows: int, rows_per_batch: int, x: Operand, x_res: Operand, agg: Operand,
                        out: Optional[torch.Tensor] = None, save: Optional[SavedActivations] = None,
                        post_w: Optional[Sequence[torch.Tensor]] = None, zero_rows: Optional[torch.Tensor] = None,
                        post_half: bool = False):
    """graph_net_block.py:189-191 (NodeProcessor after aggregation).  With ``post_w`` (packed [256, 256] slices of the NEXT
    block's layer-1 weight) the products of the new rows with them are computed in the same launch: returns
    (x_new, [products]); ``zero_rows`` [n_rows, 256] is zero-filled on the side (the next block's aggregate)."""
    import ctypes

    dev = agg.tensor.device
    if out is None:
        out = torch.empty((n_rows, pm.n_out), dtype=torch.float32, device=dev)
    _require(out, "out")
    wc = pm.c((x.k > 0 and not x.projected, True, False))
    n_post = 0 if post_w is None else len(post_w)
    outs, wp, op = [], None, None
    if n_post:
        outs = [torch.empty((n_rows, 256), dtype=torch.float16 if post_half else torch.float32, device=dev) for _ in range(n_post)]
        wp = (ctypes.c_void_p * n_post)(*[w_.data_ptr() for w_ in post_w])
        op = (ctypes.c_void_p * n_post)(*[o_.data_ptr() for o_ in outs])
    with on_device_of(out):
        _lib.check(_lib.lib().gw_node_update_forward(n_rows, rows_per_batch, x.c(), x_res.c(), agg.c(), wc, out.data_ptr(),
                                                     int(out.stride(0)), None if save is None else save.c(), n_post, wp, op,
                                                     _lib.LAYOUT_ROWS_F16 if (post_half and n_post) else _lib.LAYOUT_ROWS_F32,
                                                     None if zero_rows is None else zero_rows.data_ptr(), _stream(out)),
                   "gw_node_update_forward")
    return (out, outs) if post_w is not None else out


def node_update_head_forward(pm: PackedMLP, head: PackedMLP, n_rows: int, rows_per_batch: int, x: Operand, agg: Operand,
                             residual: Optional[Operand], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """graph_net_block.py:189-191 followed by the output head (+ residual) in one launch: include/gw_amd.h:
    gw_node_update_head_forward.  Returns [n_rows, head.n_out]."""
    if out is None:
        out = torch.empty((n_rows, head.n_out), dtype=torch.float32, device=agg.tensor.device)
    _require(out, "out")
    wc = pm.c((x.k > 0 and not x.projected, True, False))
    rc = residual.c() if residual is not None else None
    if rc is not None:
        rc.k = head.n_out
    with on_device_of(out):
        _lib.check(_lib.lib().gw_node_update_head_forward(n_rows, rows_per_batch, x.c(), agg.c(), wc, head.c(), rc, out.data_ptr(),
                                                          int(out.stride(0)), _stream(out)), "gw_node_update_head_forward")
    return out


def normalized_mse_forward(pred: torch.Tensor, target: torch.Tensor, lat_weights: torch.Tensor,
                           inv_var: Optional[torch.Tensor]) -> torch.Tensor:
    """losses.py:66-94."""
    _require(pred, "pred")
    _require(target, "target")
    _require(lat_weights, "lat_weights")
    if pred.shape != target.shape or pred.dim() < 3:
        raise RuntimeError("graph_weather_amd: pred/target must both be [B, nodes..., C]")
    b = int(pred.shape[0])
    c = int(pred.shape[-1])
    nodes = pred.numel() // (b * c)
    full = 0
    if inv_var is not None:
        _require(inv_var, "inv_var")
        if inv_var.numel() == pred.numel():
            full = 1
        elif inv_var.numel() != c:
            raise RuntimeError("graph_weather_amd: 1 / feature_variance must be per channel [C] or shaped like pred")
    loss = torch.zeros((), dtype=torch.float32, device=pred.device)
    with on_device_of(pred):
        _lib.check(_lib.lib().gw_normalized_mse_forward(pred.data_ptr(), target.data_ptr(),
                                                        None if inv_var is None else inv_var.data_ptr(), full, lat_weights.data_ptr(),
                                                        int(lat_weights.numel()), b, nodes, c, loss.data_ptr(), _stream(pred)),
                   "gw_normalized_mse_forward")
    return loss
