"""The C-ABI shared library loads without a GPU and exports every symbol include/gw_amd.h declares."""
import ctypes
import os
import re

import pytest
import torch

from graph_weather_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gw_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gw_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_loads_and_exports_header_symbols():
    path = _lib.build_library()
    assert os.path.exists(path)
    cdll = ctypes.CDLL(path)
    declared = _declared_functions()
    assert set(declared) == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(cdll, name), name
    L = _lib.lib()
    assert L.gw_version() == _lib.ABI_VERSION == 19
    assert L.gw_packed_floats(256, 0, 256) == 256 * 256
    assert L.gw_packed_floats(78, 0, 128) == 32 * 2 * 256
    assert L.gw_packed_floats(256, 0, 102) == 28 * 4 * 256
    assert L.gw_packed_floats(256, 0, 78) == 28 * 4 * 256   # padded to the 28-step layer-1 kernel (k <= 112), not 20 steps
    assert L.gw_packed_floats(256, 0, 3) == 4 * 4 * 256
    assert L.gw_padded_n(78) == 96


def test_argument_validation_without_gpu():
    L = _lib.lib()
    assert L.gw_pack_linear(None, 256, 256, 0, 256, None, None) == -1
    assert b"bad arguments" in L.gw_last_error()
    assert L.gw_edge_update_forward(0, 10, None, None, None, None, None, None, None, None, 0, None, 1, None, None, 0, 0, None) == -1
    assert L.gw_edge_update_workspace_bytes(2, 100, None, None, None, None, 0) == 0
    assert L.gw_project_forward(10, 10, None, 1, None, None, 256, 0, 0, None, None, None) == -1
    assert L.gw_pack_linear_bf16(None, 256, 256, 0, 256, None, None) == -1
    assert L.gw_packed_bytes_bf16(256, 0, 256) == 8 * 16 * 1024  # 8 K-steps x 16 row tiles x 1 KiB
    assert L.gw_packed_bytes_bf16(78, 0, 128) == 4 * 8 * 1024    # 5 tiles padded to 8
    assert L.gw_packed_bytes_bf16(256, 0, 102) == 4 * 16 * 1024
    assert L.gw_packed_bytes_bf16(256, 0, 78) == 4 * 16 * 1024   # padded to the 4-step layer-1 kernel (k <= 128), not 3 steps


def test_product_has_no_cpu_path():
    import graph_weather_amd as gw
    from graph_weather_amd.utils import regular_lat_lons

    model = gw.GraphWeatherForecaster(regular_lat_lons(30.0))
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(1, 72, 102))
    with pytest.raises(RuntimeError):
        gw.MLP(8, 256, 256)(torch.zeros(4, 8))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "graph_weather_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_edge_update_workspace_query_is_host_logic():
    """gw_edge_update_workspace_bytes: 32 KiB per 64-edge tile and batch element when the register-resident bf16 kernel
    applies (bf16 weights, one middle layer, every non-zero operand pre-projected), 0 otherwise - no GPU involved."""
    from graph_weather_amd._lib import DTYPE_BF16, DTYPE_F32, GwMlpWeights, GwOperand

    L = _lib.lib()
    from graph_weather_amd._lib import LAYOUT_EDGE_TILES_BF16

    proj = GwOperand(1, None, 10, 256, 256, 1, 0)   # ptr only has to be non-null for the query
    raw = GwOperand(1, None, 10, 256, 256, 0, 0)
    tiles = GwOperand(1, None, 0, 256, 256, 0, LAYOUT_EDGE_TILES_BF16)
    zero = GwOperand(None, None, 0, 0, 0, 0, 0)
    w = GwMlpWeights()
    w.hidden, w.n_mid, w.n_out, w.weight_dtype, w.ln_width = 256, 1, 256, DTYPE_BF16, 0
    w.ln_gamma, w.ln_beta = 1, 1  # (LayerNorm present: the resident bf16 kernels need it; pointers only have to be non-null here)
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, zero, proj, w, 0) == 0  # all projected: gathered inside the one launch
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, proj, w, 0) == 0
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, raw, w, 0) == 0   # raw fp32 rows: the streaming kernel
    w.w1[2] = 1
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, tiles, w, 0) == 3 * 3 * 32768  # raw edge operand as bf16 tiles: layer-1 kernel
    assert L.gw_edge_update_workspace_bytes(3, 130, tiles, proj, proj, w, 0) == 0  # node operands are never tiles
    assert L.gw_edge_tiles_bytes(3, 130) == 3 * 3 * 32768
    # deterministic segment sums: carry records of 528 floats per 64-column tile on top (bf16: tiles per batch element)
    w.w1[2] = None
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, proj, w, 1) == 3 * 3 * 528 * 4
    w.w1[2] = 1
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, tiles, w, 1) == 3 * 3 * (32768 + 528 * 4)
    w.w1[2] = None
    w.weight_dtype = DTYPE_F32
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, proj, raw, w, 1) == 7 * 528 * 4   # fp32: ceil(390 / 64) tiles over the flat columns
    assert L.gw_edge_update_workspace_bytes(3, 130, raw, proj, raw, w, 1) == 0   # two raw operands: the general kernel has no deterministic mode
    w.weight_dtype = DTYPE_BF16
    w.weight_dtype = DTYPE_F32
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, zero, proj, w, 0) == 0   # fp32: the streaming kernels, no scratch
    w.weight_dtype, w.n_mid = DTYPE_BF16, 2
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, zero, proj, w, 0) == 0
    w.n_mid, w.ln_width = 1, 128
    assert L.gw_edge_update_workspace_bytes(3, 130, proj, zero, proj, w, 0) == 0   # zero-padded narrow models: general kernel
