"""Small host-side helpers shared by tests, bench and the golden-vector generator."""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch


def _fill_array(key: str, shape, seed: int) -> np.ndarray:
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    n = rs.standard_normal(tuple(shape))
    if key.endswith("h3_nodes"):
        v = 0.5 * n
    elif len(shape) == 2:
        v = n / np.sqrt(shape[1])
    elif key.endswith("weight"):  # LayerNorm gain
        v = 1.0 + 0.1 * n
    else:  # biases (Linear and LayerNorm)
        v = 0.1 * n
    return v.astype(np.float32)


@torch.no_grad()
def deterministic_fill_(module_or_state: "torch.nn.Module | Dict[str, torch.Tensor]", seed: int = 0):
    """Fill every floating-point parameter from a per-key seeded stream.

    The value of a tensor depends only on (its state_dict key, its shape, seed) - not on module
    construction order or the torch RNG - so the reference model, the oracle and the HIP model get
    identical weights without shipping a 31 MB checkpoint (np.random.RandomState streams are frozen).
    """
    state = module_or_state.state_dict() if isinstance(module_or_state, torch.nn.Module) else module_or_state
    for key, t in state.items():
        if t.is_floating_point():
            t.copy_(torch.from_numpy(_fill_array(key, t.shape, seed)).to(t.device, t.dtype))
    return module_or_state


def seeded_features(batch: int, num_nodes: int, feat: int = 102, seed: int = 42) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((batch, num_nodes, feat)).astype(np.float32))


def regular_lat_lons(step: float):
    """Reference grid convention (README.md:48-51, tests/models/layers/test_efficient_batching.py:12-20)."""
    lats = np.arange(-90.0, 90.0, step)
    lons = np.arange(0.0, 360.0, step)
    return [(float(lat), float(lon)) for lat in lats for lon in lons]


def validate_lat_lons(lat_lons) -> None:
    """``graph_weather/utils.py:6-13``: non-empty sequence of (lat, lon) with latitudes in [-90, 90]."""
    if lat_lons is None or len(lat_lons) == 0:
        raise ValueError("lat_lons must not be empty.")
    for index, (lat, _lon) in enumerate(lat_lons):
        if not (-90.0 <= lat <= 90.0):
            raise ValueError(f"Coordinate {index}: latitude {lat} is outside [-90, 90].")
