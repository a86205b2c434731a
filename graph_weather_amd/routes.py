"""Route selection: which kernel form a launch of the message-passing path takes, decided in ONE place.

``layers.py`` dispatches the reference's modules (graph_net_block.py:115-228, encoder.py:197-242, processor.py:83-128,
assimilator_decoder.py:173-200) onto several kernel forms: fp32 rows, split-operand rows (bf16x3), the frozen bf16 mode with
its resident-weight kernels on edge tiles, segment-aligned tiles, the differentiable path.  Which one applies used to be
re-derived by boolean expressions at every call site (two advisor bugs of earlier rounds lived in exactly those); here each
decision is a pure function of a few plain values, returns a NAMED route, and is tested on the CPU (tests/test_routes.py).
The C side keeps the same tables where it has to re-check eligibility (csrc/gw_edge16.hip: ``edge16_eligible``;
csrc/gw_noders.hip: ``node_rs_groups``, exported as ``gw_node_update_row_split_groups`` so the two can be compared)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

BF16X3 = "bf16x3"  # split-operand products (ops.BF16X3; repeated here so this module imports nothing of the package)

# ---- one edge MLP -------------------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class MlpForm:
    """What the routes read of one MLP: matrix-product dtype and the packed shape facts the resident bf16 kernels require."""

    dtype: object   # torch.float32 | torch.bfloat16 | "bf16x3"
    n_mid: int      # middle (hidden -> hidden) layers
    ln_width: int   # features the LayerNorm spans when fewer than the tile width (0 = all of them)
    has_norm: bool

    @staticmethod
    def of(mlp, packed=None) -> "MlpForm":
        pm = mlp.packed() if packed is None else packed
        return MlpForm(mlp.compute_dtype, int(pm.n_mid), int(pm.ln_width), pm.gamma is not None)


EDGE_AUTOGRAD = "autograd"        # differentiable path (autograd.py): fp32 / bf16x3 kernels that also write activation saves
EDGE_ROWS_FP32 = "rows_fp32"      # csrc/gw_edge.hip (edge_kernel) / chain_kernel: fp32 rows in, fp32 rows out
EDGE_ROWS_X3 = "rows_bf16x3"      # csrc/gw_split.hip: the same tables, split-operand products
EDGE_TILES_BF16 = "tiles_bf16"    # frozen budget mode: resident-weight kernels, per-sample edge features as bf16 edge tiles
EDGE_ROWS_BF16 = "rows_bf16"      # frozen budget mode, shapes the resident kernels do not take: streaming bf16 kernel on rows


def resident_bf16(m: MlpForm, n_edges: int) -> bool:
    """The resident-weight bf16 kernels (csrc/gw_edge16*.hip) take this edge MLP: one middle layer, LayerNorm over all 256
    features, a non-empty edge list.  (Mirror of ``edge16_eligible`` on the C side, which also checks the operands.)"""
    return m.dtype == torch.bfloat16 and m.n_mid == 1 and m.ln_width == 0 and m.has_norm and n_edges > 0


def edge_route(m: MlpForm, n_edges: int, train: bool) -> str:
    if train:
        return EDGE_AUTOGRAD
    if m.dtype == torch.float32:
        return EDGE_ROWS_FP32
    if m.dtype == BF16X3:
        return EDGE_ROWS_X3
    return EDGE_TILES_BF16 if resident_bf16(m, n_edges) else EDGE_ROWS_BF16


def edge_out_kind(route: str, need_edges: bool, next_route: Optional[str]):
    """How a block hands e' to its consumer: "tiles" only between two blocks that both read tiles (set_compute_dtype on a
    sub-module can leave neighbours in different modes), fp32 rows (True) otherwise, nothing (False) when nobody reads it."""
    if not need_edges:
        return False
    return "tiles" if (route == EDGE_TILES_BF16 and next_route == EDGE_TILES_BF16) else True


def post_products_half(next_route: Optional[str]) -> bool:
    """The node update writes the NEXT block's layer-1 node products as fp16 rows when that block gathers them in the bf16
    layer-1 kernel (once per incident edge: their bytes dominate what it reads)."""
    return next_route == EDGE_TILES_BF16


# ---- one bipartite block (encoder / decoder) -------------------------------------------------------------------------------

BLOCK_ROWS = "rows"      # edge update with residual, aggregate as fp32 rows
BLOCK_TEAM = "team"      # bf16: team-pipelined edge kernel, no residual (sums of e enter the node update as a cached table)
BLOCK_SPLIT = "split"    # bf16x3: every layer-1 operand projected, no residual, fp32 rows


def block_route(edge: MlpForm, node_dtype, n_edges: int, wide: bool, autograd: bool, deterministic: bool) -> str:
    if wide or autograd or n_edges <= 0:
        return BLOCK_ROWS
    if edge.dtype == BF16X3 and node_dtype == BF16X3:
        return BLOCK_SPLIT
    if edge.dtype == torch.bfloat16 and node_dtype == torch.bfloat16 and not deterministic and resident_bf16(edge, n_edges):
        return BLOCK_TEAM
    return BLOCK_ROWS


# ---- the processor stack ------------------------------------------------------------------------------------------------


def stack_on_segment_tiles(blocks: Sequence[Tuple[MlpForm, object, bool]], n_edges: int, max_slots: Optional[int]) -> bool:
    """The whole stack runs on the plan's segment-aligned tiles (csrc/gw_edge16p.hip): bf16 inference with the resident kernels
    in EVERY block (edge form, node dtype, deterministic flag per block), atomics mode, at most 16 destinations per tile
    (``max_slots`` of ``GraphPlan.seg_tiles()``; None = the plan has no such tiling)."""
    if len(blocks) == 0 or n_edges <= 0 or max_slots is None or max_slots > 16:
        return False
    return all(resident_bf16(e, n_edges) and nd == torch.bfloat16 and not det for e, nd, det in blocks)


def segment_route_allowed(train: bool, want_edges: bool, lo: int, shared: bool) -> bool:
    """... and this call may take it: inference, e' of the last block not requested, starting at block 0, and block 0's edge
    features BATCH-SHARED (the running aggregate starts from their cached segment sums; per-sample features handed in by a
    caller - Processor.forward without efficient batching - take the tile route without segment alignment: ADVICE r4)."""
    return (not train) and (not want_edges) and lo == 0 and shared


def mesh_streams(streams_attr: int, edge_dtypes: Sequence[object], batch: int) -> int:
    """HIP streams the fused inference forward runs the mesh stack on: the ``streams`` attribute if set, else per-sample chains
    on 2 streams for the 64-column-workgroup modes (fp32, bf16x3) at batch >= 2 (their mesh-sized launches leave workgroup
    slots idle in the last round; the bf16 kernels are persistent and occupy every CU by themselves), else 1."""
    if streams_attr > 0:
        return max(1, min(int(streams_attr), batch))
    tiled = all(d in (torch.float32, BF16X3) for d in edge_dtypes)
    return 2 if (tiled and batch >= 2) else 1


# ---- node update form (csrc/gw_noders.hip) ---------------------------------------------------------------------------------

NODE_COLS64 = "cols64"  # chain_kernel / chainx3_kernel / chain16_kernel: one wave per 16-column group, 64 columns per workgroup


def node_update_row_split_groups(n_rows: int) -> int:
    """Column groups per workgroup of the row-split node update, 0 = the 64-column kernels: mesh-sized launches only - at most
    one round of 48-column workgroups on the 256 CUs (768 column groups of 16).  Mirror of ``node_rs_groups``
    (``gw_node_update_row_split_groups`` in the C ABI)."""
    groups = (int(n_rows) + 15) // 16
    if groups <= 0 or groups > 3 * 256:
        return 0
    return max(1, (groups + 255) // 256)


def node_update_form(dtype, n_rows: int, agg_fp32_rows: bool, x_mode: str, n_mid: int, ln_width: int, saves: bool) -> str:
    """"row_split_cgN" or NODE_COLS64.  ``x_mode``: "raw" | "proj" (fp32 product rows) | "zero" | "other" (16-bit table formats
    of the frozen bf16 mode).  (Mirror of ``node_rs_eligible``.)"""
    cg = node_update_row_split_groups(n_rows)
    ok = (dtype in (torch.float32, BF16X3) and cg > 0 and agg_fp32_rows and x_mode in ("raw", "proj", "zero") and n_mid == 1
          and ln_width == 0 and not saves)
    return "row_split_cg%d" % cg if ok else NODE_COLS64
