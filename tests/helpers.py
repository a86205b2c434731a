from typing import Dict, Tuple

import numpy as np
import torch

from graph_weather_amd.utils import deterministic_fill_


def _mlp_shapes(prefix: str, i: int, h: int, o: int, norm: bool) -> Dict[str, Tuple[int, ...]]:
    s = {f"{prefix}.model.0.weight": (h, i), f"{prefix}.model.0.bias": (h,), f"{prefix}.model.2.weight": (h, h),
         f"{prefix}.model.2.bias": (h,), f"{prefix}.model.4.weight": (o, h), f"{prefix}.model.4.bias": (o,)}
    if norm:
        s.update({f"{prefix}.model.5.weight": (o,), f"{prefix}.model.5.bias": (o,)})
    return s


def _block_shapes(prefix: str, d: int = 256):
    s = {}
    s.update(_mlp_shapes(prefix + ".edge_model.edge_mlp", 3 * d, d, d, True))
    s.update(_mlp_shapes(prefix + ".node_model.node_mlp", 2 * d, d, d, True))
    return s


def forecaster_param_shapes(num_mesh: int, feat: int = 102, out: int = 78, d: int = 256, blocks: int = 9, hd: int = 128):
    """state_dict layout of the reference GraphWeatherForecaster at default dims."""
    s = {"encoder.h3_nodes": (num_mesh, feat)}
    s.update(_mlp_shapes("encoder.node_encoder", feat, d, d, True))
    s.update(_mlp_shapes("encoder.edge_encoder", 2, d, d, True))
    s.update(_mlp_shapes("encoder.latent_edge_encoder", 2, d, d, True))
    s.update(_block_shapes("encoder.graph_processor.blocks.0", d))
    for b in range(blocks):
        s.update(_block_shapes(f"processor.graph_processor.blocks.{b}", d))
    s.update(_mlp_shapes("decoder.edge_encoder", 2, d, d, True))
    s.update(_block_shapes("decoder.graph_processor.blocks.0", d))
    s.update(_mlp_shapes("decoder.node_decoder", d, hd, out, False))
    return s


def make_params(shapes, seed: int = 0) -> Dict[str, torch.Tensor]:
    p = {k: torch.empty(s) for k, s in shapes.items()}
    deterministic_fill_(p, seed)
    return p


def k_of(s: int, q: int) -> int:
    """K walk order of the kernels: step s, lane quarter q -> input feature index."""
    return 16 * (s >> 2) + 4 * q + (s & 3)


def pack_linear_ref(w: np.ndarray, k_lo: int, k_hi: int) -> np.ndarray:
    """numpy statement of gw_pack_linear's layout: out[s][b4][lane][i]."""
    n_out = w.shape[0]
    kseg = k_hi - k_lo
    nsteps = 4 if kseg <= 16 else (28 if kseg <= 112 else ((kseg + 15) // 16) * 4)  # padded to the layer-1 kernel variant's steps
    nt = (n_out + 15) // 16
    nt4 = (nt + 3) // 4
    out = np.zeros((nsteps, nt4, 64, 4), dtype=np.float32)
    for s in range(nsteps):
        for lane in range(64):
            kk = k_of(s, lane >> 4)
            if kk >= kseg:
                continue
            for b4 in range(nt4):
                for i in range(4):
                    f = 16 * (4 * b4 + i) + (lane & 15)
                    if f < n_out:
                        out[s, b4, lane, i] = w[f, k_lo + kk]
    return out


def mfma_16x16x4_emulate(a_lane: np.ndarray, b_lane: np.ndarray, acc: np.ndarray) -> np.ndarray:
    """v_mfma_f32_16x16x4_f32 with the documented operand layouts (cdna_hip_programming.md section 3):
    A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15], D col=lane&15, row=4*(lane>>4)+r.
    a_lane, b_lane: [64]; acc: [4, 64] (register r, lane)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for lane in range(64):
        A[lane & 15, lane >> 4] = a_lane[lane]
        B[lane >> 4, lane & 15] = b_lane[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        for r in range(4):
            out[r, lane] += D[4 * (lane >> 4) + r, lane & 15]
    return out


def k16_of(s: int, q: int, i: int) -> int:
    """K walk order of the bf16 kernels: K-step s (32 wide), lane quarter q, element i of the 8-vector."""
    return 32 * s + 16 * (i >> 2) + 4 * q + (i & 3)


def pack_linear_bf16_ref(w: np.ndarray, k_lo: int, k_hi: int) -> np.ndarray:
    """numpy statement of gw_pack_linear_bf16's layout: out[s][tile][lane][i] (float32 values before rounding)."""
    n_out = w.shape[0]
    kseg = k_hi - k_lo
    nsteps = 1 if kseg <= 32 else (4 if kseg <= 128 else (kseg + 31) // 32)  # padded to the layer-1 kernel variant's steps
    ntp = (((n_out + 15) // 16) + 3) // 4 * 4
    out = np.zeros((nsteps, ntp, 64, 8), dtype=np.float32)
    for s in range(nsteps):
        for lane in range(64):
            for i in range(8):
                kk = k16_of(s, lane >> 4, i)
                if kk >= kseg:
                    continue
                for tile in range(ntp):
                    f = 16 * tile + (lane & 15)
                    if f < n_out:
                        out[s, tile, lane, i] = w[f, k_lo + kk]
    return out


def edge_tile_feature_index() -> np.ndarray:
    """[8 (K-step s), 4 (q), 8 (i)] -> feature 32 s + 16 (i >> 2) + 4 q + (i & 3): the bf16 MFMA B-operand order of an edge tile
    (include/gw_amd.h: GW_LAYOUT_EDGE_TILES_BF16)."""
    s, q, i = np.meshgrid(np.arange(8), np.arange(4), np.arange(8), indexing="ij")
    return 32 * s + 16 * (i >> 2) + 4 * q + (i & 3)


def edge_tiles_from_rows(rows: torch.Tensor) -> torch.Tensor:
    """numpy/torch statement of gw_edge_rows_to_tiles: rows [B, E, 256] fp32 -> bf16 tensor [B, ceil(E/64), 4, 8, 64, 8]
    (tile, group, K-step, lane = 16 q + column, element i); padding edges are zero."""
    B, E, _ = rows.shape
    neb = (E + 63) // 64
    padded = torch.zeros((B, neb * 64, 256), dtype=torch.float32)
    padded[:, :E] = rows
    f = torch.from_numpy(edge_tile_feature_index())  # [8, 4, 8]
    x = padded.reshape(B, neb, 4, 16, 256)  # [B, tile, group, column j, feature]
    # out[b, tile, g, s, q, j, i] = x[b, tile, g, j, f[s, q, i]]
    out = x[:, :, :, :, f]  # [B, neb, 4, 16(j), 8(s), 4(q), 8(i)]
    out = out.permute(0, 1, 2, 4, 5, 3, 6).contiguous()  # [B, neb, 4, s, q, j, i]
    return out.reshape(B, neb, 4, 8, 64, 8).to(torch.bfloat16)


def edge_rows_from_tiles(tiles: torch.Tensor, E: int) -> torch.Tensor:
    """Inverse of edge_tiles_from_rows (values as float32): [B, neb, 4, 8, 64, 8] -> [B, E, 256]."""
    B, neb = tiles.shape[0], tiles.shape[1]
    t = tiles.float().reshape(B, neb, 4, 8, 4, 16, 8)  # [B, neb, g, s, q, j, i]
    f = torch.from_numpy(edge_tile_feature_index()).reshape(-1)  # order (s, q, i)
    t = t.permute(0, 1, 2, 5, 3, 4, 6).reshape(B, neb, 4, 16, 256)  # [.., j, (s, q, i)]
    rows = torch.empty_like(t)
    rows[..., f] = t
    return rows.reshape(B, neb * 64, 256)[:, :E]


def bf16k_feature_of_position() -> np.ndarray:
    """GW_LAYOUT_ROWS_BF16K (include/gw_amd.h): position 32 s + 8 q + i of a row holds feature k16_of(s, q, i)."""
    f = np.empty(256, dtype=np.int64)
    for s in range(8):
        for q in range(4):
            for i in range(8):
                f[32 * s + 8 * q + i] = k16_of(s, q, i)
    assert sorted(f.tolist()) == list(range(256))
    return f


def rows_to_bf16k(rows: torch.Tensor) -> torch.Tensor:
    """fp32 rows [n, 256] -> bf16 rows in K order (round to nearest even)."""
    return rows[:, torch.from_numpy(bf16k_feature_of_position())].to(torch.bfloat16).contiguous()


def rows_from_bf16k(rows_k: torch.Tensor) -> torch.Tensor:
    """bf16 rows in K order -> fp32 rows [n, 256] in feature order."""
    out = torch.empty(rows_k.shape, dtype=torch.float32)
    out[:, torch.from_numpy(bf16k_feature_of_position())] = rows_k.float().cpu()
    return out
