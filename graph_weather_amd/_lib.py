"""ctypes binding of ``libgw_amd.so`` (C ABI declared in ``include/gw_amd.h``).

There is deliberately no fallback: if the shared library is missing or a call fails, a
``RuntimeError`` is raised - the product path never computes on the CPU or through torch ops.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libgw_amd.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-shared", "-fPIC"]

ABI_VERSION = 19
DTYPE_F32, DTYPE_BF16, DTYPE_BF16X3 = 0, 1, 2
LAYOUT_ROWS_F32, LAYOUT_EDGE_TILES_BF16, LAYOUT_ROWS_F16, LAYOUT_ROWS_BF16K = 0, 1, 2, 3
EDGE_DETERMINISTIC, EDGE_SEGMENT_TILES, EDGE_AGG_BF16K, EDGE_SEGMENT_SPLIT = 1, 2, 4, 8

EXPORTS = [
    "gw_version", "gw_last_error", "gw_debug_timestamps", "gw_packed_floats", "gw_pack_linear", "gw_packed_bytes_bf16",
    "gw_pack_linear_bf16", "gw_packed_bytes_bf16x3", "gw_pack_linear_bf16x3", "gw_padded_n", "gw_pad_vector", "gw_pack_many", "gw_mlp_chain_backward", "gw_mlp_chain_backward_bf16x3", "gw_mlp_ln_chain_backward",
    "gw_mlp_forward", "gw_mlp_post_forward", "gw_project_forward", "gw_edge_update_forward", "gw_edge_update_workspace_bytes", "gw_edge_tiles_bytes",
    "gw_edge_rows_to_tiles", "gw_node_update_forward", "gw_node_update_row_split_groups", "gw_node_update_head_forward",
    "gw_normalized_mse_forward", "gw_gemm_f32", "gw_relu_backward", "gw_layernorm_backward", "gw_gather_rows",
    "gw_segment_sum_rows", "gw_normalized_mse_backward", "gw_adamw_step", "gw_nudging_forward", "gw_nudging_backward",
    "gw_linear_forward", "gw_linear_gather_forward", "gw_layernorm_forward", "gw_add_rows", "gw_gather_rows_wide", "gw_segment_sum_rows_wide",
]

GEMM_NN, GEMM_TN, GEMM_TN_BF16X3 = 0, 1, 2


class GwOperand(Structure):
    _fields_ = [("ptr", c_void_p), ("index", c_void_p), ("rows_per_batch", c_int32), ("ld", c_int32), ("k", c_int32),
                ("projected", c_int32), ("layout", c_int32)]


class GwMlpWeights(Structure):
    _fields_ = [("w1", c_void_p * 3), ("b1", c_void_p), ("w_mid", c_void_p), ("b_mid", c_void_p), ("w_out", c_void_p),
                ("b_out", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("hidden", c_int32),
                ("n_mid", c_int32), ("n_out", c_int32), ("weight_dtype", c_int32), ("ln_width", c_int32), ("k_in", c_int32),
                ("out_rows", c_int32)]


PACK_MAX_ITEMS = 16


class GwPackItem(Structure):  # include/gw_amd.h: gw_pack_item
    _fields_ = [("w", c_void_p), ("stride_f", c_int64), ("stride_k", c_int64), ("n_out", c_int32), ("kseg", c_int32),
                ("rows", c_int32), ("reserved", c_int32), ("out", c_void_p)]


class GwPadItem(Structure):  # include/gw_amd.h: gw_pad_item
    _fields_ = [("v", c_void_p), ("n", c_int32), ("n_out", c_int32), ("out", c_void_p)]


class GwActivationSave(Structure):
    _fields_ = [("hidden", c_void_p), ("hidden_stride", c_int64), ("hidden_ld", c_int32), ("pre_norm", c_void_p)]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libgw_amd.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "gw_amd.h"))
    # GW_TUNING=1 builds the A/B knobs in (env-selected kernel variants, skip switches that give wrong results - used by
    # scripts/gpu_tune.sh and GW_TUNING=1 scripts/gpu_run.sh only; GW_HIPCC_EXTRA adds flags such as -DGW_LAYER_ABL=1); the shipped library is built without them.  The flag set is recorded beside
    # the library so that switching modes rebuilds.
    flags = HIPCC_FLAGS + (["-DGW_TUNING"] + os.environ.get("GW_HIPCC_EXTRA", "").split() if os.environ.get("GW_TUNING") == "1" else [])
    stamp = LIB_PATH + ".flags"
    want = " ".join(flags)
    try:
        have = open(stamp).read()
    except OSError:
        have = want if "-DGW_TUNING" not in want else ""
    if (not force and os.path.exists(LIB_PATH) and have == want
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + flags + srcs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(want)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "graph_weather_amd: %s is missing - the HIP extension must be built (python -c 'import __graft_entry__ as g; "
            "g.build()'); there is no CPU/torch fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.gw_version.restype = c_int
    L.gw_last_error.restype = c_char_p
    L.gw_debug_timestamps.restype = c_int
    L.gw_debug_timestamps.argtypes = [c_void_p, c_int, c_int]
    L.gw_packed_floats.restype = c_size_t
    L.gw_packed_floats.argtypes = [c_int, c_int, c_int]
    L.gw_pack_linear.restype = c_int
    L.gw_pack_linear.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.gw_packed_bytes_bf16.restype = c_size_t
    L.gw_packed_bytes_bf16.argtypes = [c_int, c_int, c_int]
    L.gw_pack_linear_bf16.restype = c_int
    L.gw_pack_linear_bf16.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.gw_packed_bytes_bf16x3.restype = c_size_t
    L.gw_packed_bytes_bf16x3.argtypes = [c_int, c_int, c_int]
    L.gw_pack_linear_bf16x3.restype = c_int
    L.gw_pack_linear_bf16x3.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.gw_padded_n.restype = c_int
    L.gw_padded_n.argtypes = [c_int]
    L.gw_pad_vector.restype = c_int
    L.gw_pad_vector.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
    L.gw_mlp_chain_backward.restype = c_int
    L.gw_mlp_chain_backward.argtypes = [c_int64, c_void_p, c_int32, c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                        c_int32, POINTER(c_void_p), POINTER(c_void_p), c_void_p]
    L.gw_mlp_chain_backward_bf16x3.restype = c_int
    L.gw_mlp_chain_backward_bf16x3.argtypes = L.gw_mlp_chain_backward.argtypes
    L.gw_mlp_ln_chain_backward.restype = c_int
    L.gw_mlp_ln_chain_backward.argtypes = [c_int32, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                           c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int32, ctypes.c_uint32, c_void_p]
    L.gw_pack_many.restype = c_int
    L.gw_pack_many.argtypes = [c_int32, c_int32, POINTER(GwPackItem), c_int32, POINTER(GwPadItem), c_void_p]
    L.gw_mlp_forward.restype = c_int
    L.gw_mlp_forward.argtypes = [c_int64, c_int32, POINTER(GwOperand), POINTER(GwMlpWeights), POINTER(GwOperand),
                                 c_void_p, c_int32, POINTER(GwActivationSave), c_void_p]
    L.gw_mlp_post_forward.restype = c_int
    L.gw_mlp_post_forward.argtypes = [c_int64, c_int32, POINTER(GwOperand), POINTER(GwMlpWeights), c_void_p, c_int32, c_int32,
                                      POINTER(c_void_p), POINTER(c_void_p), c_int32, c_void_p]
    L.gw_edge_update_forward.restype = c_int
    L.gw_edge_update_forward.argtypes = [c_int32, c_int32, c_void_p, c_void_p, POINTER(GwOperand), POINTER(GwOperand),
                                         POINTER(GwOperand), POINTER(GwOperand), POINTER(GwMlpWeights), c_void_p, c_int32, c_void_p,
                                         c_int32, POINTER(GwActivationSave), c_void_p, c_size_t, c_int32, c_void_p]
    L.gw_edge_tiles_bytes.restype = c_size_t
    L.gw_edge_tiles_bytes.argtypes = [c_int32, c_int32]
    L.gw_edge_rows_to_tiles.restype = c_int
    L.gw_edge_rows_to_tiles.argtypes = [c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]
    L.gw_edge_update_workspace_bytes.restype = c_size_t
    L.gw_edge_update_workspace_bytes.argtypes = [c_int32, c_int32, POINTER(GwOperand), POINTER(GwOperand), POINTER(GwOperand),
                                                 POINTER(GwMlpWeights), c_int32]
    L.gw_node_update_forward.restype = c_int
    L.gw_node_update_forward.argtypes = [c_int64, c_int32, POINTER(GwOperand), POINTER(GwOperand), POINTER(GwOperand),
                                         POINTER(GwMlpWeights), c_void_p, c_int32, POINTER(GwActivationSave), c_int32,
                                         POINTER(c_void_p), POINTER(c_void_p), c_int32, c_void_p, c_void_p]
    L.gw_node_update_row_split_groups.restype = c_int
    L.gw_node_update_row_split_groups.argtypes = [c_int64]
    L.gw_node_update_head_forward.restype = c_int
    L.gw_node_update_head_forward.argtypes = [c_int64, c_int32, POINTER(GwOperand), POINTER(GwOperand), POINTER(GwMlpWeights),
                                              POINTER(GwMlpWeights), POINTER(GwOperand), c_void_p, c_int32, c_void_p]
    L.gw_project_forward.restype = c_int
    L.gw_project_forward.argtypes = [c_int64, c_int32, POINTER(GwOperand), c_int32, POINTER(c_void_p), POINTER(c_void_p),
                                     c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    L.gw_normalized_mse_forward.restype = c_int
    L.gw_normalized_mse_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_void_p, c_void_p]
    L.gw_gemm_f32.restype = c_int
    L.gw_gemm_f32.argtypes = [c_int32, c_int64, c_int32, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                              c_void_p]
    L.gw_relu_backward.restype = c_int
    L.gw_relu_backward.argtypes = [c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]
    L.gw_layernorm_backward.restype = c_int
    L.gw_layernorm_backward.argtypes = [c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                        c_void_p, c_void_p]
    L.gw_gather_rows.restype = c_int
    L.gw_gather_rows.argtypes = [c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    L.gw_segment_sum_rows.restype = c_int
    L.gw_segment_sum_rows.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]
    L.gw_normalized_mse_backward.restype = c_int
    L.gw_normalized_mse_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                             c_void_p, c_void_p, c_void_p]
    L.gw_adamw_step.restype = c_int
    L.gw_adamw_step.argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                c_int32, c_void_p]
    L.gw_nudging_forward.restype = c_int
    L.gw_nudging_forward.argtypes = [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]
    L.gw_nudging_backward.restype = c_int
    L.gw_nudging_backward.argtypes = [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    L.gw_linear_forward.restype = c_int
    L.gw_linear_forward.argtypes = [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                    c_void_p]
    L.gw_layernorm_forward.restype = c_int
    L.gw_layernorm_forward.argtypes = [c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p,
                                       c_int32, c_void_p]
    L.gw_linear_gather_forward.restype = c_int
    L.gw_linear_gather_forward.argtypes = [c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                           POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), c_int32, c_void_p,
                                           c_int32, c_void_p]
    L.gw_add_rows.restype = c_int
    L.gw_add_rows.argtypes = [c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p]
    L.gw_gather_rows_wide.restype = c_int
    L.gw_gather_rows_wide.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]
    L.gw_segment_sum_rows_wide.restype = c_int
    L.gw_segment_sum_rows_wide.argtypes = [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_void_p]
    if L.gw_version() != ABI_VERSION:
        raise RuntimeError("graph_weather_amd: libgw_amd.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gw_last_error().decode(errors="replace")
        raise RuntimeError("graph_weather_amd: %s failed (%d): %s" % (what, rc, msg))
