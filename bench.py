#!/usr/bin/env python
"""bench.py - forward forecasts/s of GraphWeatherForecaster on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one forward pass of the hot path over one batch of synthetic input that is already resident in HBM.
Default workload at every N = BASELINE.json configs[1] per GPU ("c2"): 1 degree grid (64 800 nodes), 102 -> 78 features,
batch 2, fp32 arithmetic (fp32-in / fp32-accumulate MFMA), random-init weights.  Batch elements are independent, so N GPUs
run N independent shards with no data-path collective (weak scaling); value = all forecasts / max-over-ranks time.

Other configurations of BASELINE.json (--config): c3 = same grid, bf16 matrix products, batch 16; c4 = global batch 64 sharded
over the ranks (8 per GPU at N = 8; strong scaling); c5 = 0.25 degree grid, mesh resolution 3, batch 1.

Prints ONE JSON line with the driver contract fields plus
  "roofline":     the dominant kernel (decoder edge update): EXECUTED matrix FLOPs per launch / HIP-event duration against the
                  MFMA peak of the arithmetic dtype (frac <= 1); the figure on the reference's algorithmic FLOPs (which the
                  layer-1 split legally does not execute) is kept under algorithmic_*; step_frac = the whole forward;
  "cold_ms_per_step": a forward right after the weights changed (every per-weight-version cache misses: edge / mesh embeddings
                  and their layer-1 products recomputed, weights re-packed) - what the reference pays on every forward;
  "extra":        c3 and c5 measured in the same run (N = 1 only), each with value, ms_per_step and its dominant kernel;
  "cpu_baseline": the CPU oracle (port of the reference forward) on this host: batch 2, 1 warm-up + 3 timed forwards, mean
                  and min, replicated-graph semantics (what the reference executes) with the shared-graph variant beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 acc
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparsity figure)
PEAK_HBM_TBS = 8.0
D = 256
EDGE_MLP_FLOPS = 2 * (768 * 256 + 256 * 256 + 256 * 256)  # per edge, SURVEY.md 8(d): mlp(768,256,256) - algorithmic
LAYER = 2 * D * D  # one 256 x 256 layer on one column

CONFIGS = {
    "c5x3": dict(grid=0.25, resolution=3, batch=1, precision="bf16x3"),  # c5 with split-operand products
    "c2": dict(grid=1.0, resolution=2, batch=2, precision="fp32"),
    "c2x3": dict(grid=1.0, resolution=2, batch=2, precision="bf16x3"),   # c2 with split-operand products (inside the 1e-3 bar)
    "c3": dict(grid=1.0, resolution=2, batch=16, precision="bf16"),
    "c3x3": dict(grid=1.0, resolution=2, batch=16, precision="bf16x3"),  # c3's batch with split-operand products
    "c4": dict(grid=1.0, resolution=2, batch=64, precision="fp32"),  # global batch, sharded over the ranks
    "c5": dict(grid=0.25, resolution=3, batch=1, precision="fp32"),
    "c4g": dict(grid=1.0, resolution=2, batch=8, precision="fp32"),      # what ONE GPU carries of c4 at N = 8
    "c4gx3": dict(grid=1.0, resolution=2, batch=8, precision="bf16x3"),
}


def executed_flops_per_forecast(graphs) -> dict:
    """Matrix FLOPs the kernels execute per forecast (default dims) after the layer-1 split, by stage.  Per column: one
    256x256 layer = 131 072 FLOP.  Batch-independent products (edge / mesh embeddings and their layer-1 products) are cached
    per weight version in eval and are not in this count (they are what cold_ms_per_step adds)."""
    G, M = graphs.num_grid, graphs.num_mesh
    e_enc, e_lat, e_dec = graphs.enc_plan.num_edges, graphs.lat_plan.num_edges, graphs.dec_plan.num_edges
    enc = G * 2 * (112 * D + D * D + D * D) + e_enc * 3 * LAYER + M * 3 * LAYER  # node encoder (K padded to 112), edge (raw x_src), node (raw agg)
    proc = 0
    for b in range(9):
        proc += M * 2 * LAYER  # P_s, P_d
        proc += e_lat * (2 if b == 0 else 3) * LAYER  # block 0: e is batch shared (product cached)
        proc += M * 4 * LAYER  # node update: x raw + agg raw + 2 layers
    dec = M * LAYER + e_dec * 2 * LAYER + G * 3 * LAYER + G * 2 * (D * 128 + 128 * 128 + 128 * 80)
    return {"encoder": enc, "processor": proc, "decoder": dec, "total": enc + proc + dec}


def algorithmic_flops_per_forecast(graphs) -> float:
    """model-definition math on live rows."""
    G, M = graphs.num_grid, graphs.num_mesh
    e_enc, e_lat, e_dec = graphs.enc_plan.num_edges, graphs.lat_plan.num_edges, graphs.dec_plan.num_edges

    def mlp(i, h, o, rows):
        return 2 * rows * (i * h + h * h + h * o)

    enc = mlp(102, D, D, G) + mlp(768, D, D, e_enc) + mlp(512, D, D, M)
    proc = 9 * (mlp(768, D, D, e_lat) + mlp(512, D, D, M))
    dec = mlp(768, D, D, e_dec) + mlp(512, D, D, G) + 2 * G * (D * 128 + 128 * 128 + 128 * 78)
    return enc + proc + dec


def one_socket_cpus():
    """Logical CPUs of physical package 0 (sysfs topology), or None when the topology cannot be read."""
    import glob

    cpus = []
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
        try:
            if int(open(os.path.join(d, "topology", "physical_package_id")).read()) == 0:
                cpus.append(int(os.path.basename(d)[3:]))
        except (OSError, ValueError):
            return None
    return sorted(cpus) or None


def cpu_baseline(lat_lons, state, graphs):
    """SURVEY.md 8(d) protocol: the oracle forward (port of the reference; fp32, eval, no_grad) on this host's cores, batch 2,
    1 warm-up + 3 timed forwards, mean and min; replicated-graph semantics (what the reference executes by default) and the
    shared-graph variant (encoder.py:168-196 ...) beside it, to separate the reference's replicated-graph waste from the
    hardware ratio.  Threads: 32 (measured round 1: faster than all 256 logical CPUs for these scatter / GEMM sizes), and -
    north_star's wording, "single-socket" - the replicated variant again pinned to the CPUs of socket 0 with one thread per
    physical core of that socket (at most 64)."""
    from graph_weather_amd.utils import seeded_features
    from oracle import reference_math as om

    feats = seeded_features(2, len(lat_lons), 102, seed=42)
    g = graphs.as_oracle_dict()
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    threads = max(1, min(32, ncpu))

    def timed(shared, n=3):
        om.forecaster_forward(state, g, feats, shared=shared)  # warm-up
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            om.forecaster_forward(state, g, feats, shared=shared)
            ts.append(time.perf_counter() - t0)
        return ts

    res = {}
    socket = None
    with torch.no_grad():
        torch.set_num_threads(threads)
        for name, shared in (("replicated", False), ("shared", True)):
            res[name] = timed(shared)
        # one socket: affinity = the logical CPUs of package 0, threads = its physical cores (SMT siblings left idle)
        cpus = one_socket_cpus()
        if cpus and hasattr(os, "sched_setaffinity") and len(cpus) < ncpu:
            old = os.sched_getaffinity(0)
            try:
                os.sched_setaffinity(0, set(cpus) & old or old)
                st = max(1, min(64, len(set(cpus) & old) // 2 or 1))
                torch.set_num_threads(st)
                ts = timed(False)
                socket = {"threads": st, "logical_cpus_of_socket": len(cpus), "seconds": ts, "value": 2.0 / (sum(ts) / len(ts)),
                          "value_best": 2.0 / min(ts)}
            finally:
                os.sched_setaffinity(0, old)
    torch.set_num_threads(default_threads)
    rep, sh = res["replicated"], res["shared"]
    mean = sum(rep) / len(rep)
    sample = ("1 degree, batch 2, fp32 torch CPU oracle (port of the reference forward): 1 warm-up + 3 timed forwards per "
              "variant; replicated graph (reference default) " + ", ".join(f"{t:.2f}s" for t in rep)
              + "; shared graph " + ", ".join(f"{t:.2f}s" for t in sh) + f"; {threads} threads of {ncpu} logical CPUs")
    if socket is not None:
        sample += ("; replicated graph pinned to socket 0 (" + str(socket["threads"]) + " threads on its "
                   + str(socket["logical_cpus_of_socket"]) + " logical CPUs): " + ", ".join(f"{t:.2f}s" for t in socket["seconds"]))
    return {"value": 2.0 / mean, "unit": "forecasts/s", "cores": threads, "kind": "port",
            "value_best": 2.0 / min(rep), "shared_graph_value": 2.0 / (sum(sh) / len(sh)), "shared_graph_value_best": 2.0 / min(sh),
            "single_socket": socket, "sample": sample}


def pmc_traffic(cfg="c2"):
    """HBM bytes per launch of the dominant kernel from the NEWEST committed rocprofv3 --pmc summary of the workload
    (profiles/rNN_pmc_<cfg>.json, re-collected every round by scripts/gpu_final.sh; profiles/pmc_decoder_edge.json = round 2's c2):
    separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction on FETCH_SIZE applied by scripts/gpu_pmc.sh.  Returns
    (bytes, summary dict, file name); None if not collected.  These counters are NOT collected in the bench run itself: the
    figure is read from the tracked profile of the same workload (a PMC pass serialises kernels and cannot share a run with
    the timed region)."""
    import glob

    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % cfg)), reverse=True)
    if cfg == "c2":
        names.append(os.path.join(ROOT, "profiles", "pmc_decoder_edge.json"))
    for p in names:
        try:
            d = json.load(open(p))
            if d.get("hbm_read_bytes") is None or d.get("hbm_write_bytes") is None:
                continue
            return d["hbm_read_bytes"] + d["hbm_write_bytes"], d, os.path.basename(p)
        except (OSError, ValueError):
            continue
    return None, None, None


def pmc_kernel_traffic(cfg, name_prefixes, grid_threads):
    """HBM bytes per launch of ONE kernel instantiation at ONE grid size from the newest committed per-kernel PMC summary of the
    workload (profiles/rNN_pmc_<cfg>_all_kernels.json, scripts/gpu_pmc_cfg.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes).
    Returns (bytes, file name, mfma_busy_frac) or (None, None, None)."""
    import glob

    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s_all_kernels.json" % cfg)), reverse=True):
        try:
            d = json.load(open(p))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            name, _, grid = k.rpartition(" grid=")
            try:
                g = int(float(grid))
            except ValueError:
                continue
            if g == grid_threads and name.startswith(tuple(name_prefixes)) and v.get("hbm_read_bytes") is not None \
                    and v.get("hbm_write_bytes") is not None:
                return v["hbm_read_bytes"] + v["hbm_write_bytes"], os.path.basename(p), v.get("mfma_busy_frac")
    return None, None, None


def pmc_c3_traffic():
    """Counter traffic (bytes per launch) of the bf16 kernels at C3 from the tracked per-kernel PMC summary: the newest of
    profiles/r04_pmc_c3.json / r03_pmc_c3.json / r02_pmc_c3_edge16_v2.json.  Returns ({"processor_block": bytes, "decoder": bytes}, file)."""
    for name in ("r05_pmc_c3.json", "r04_pmc_c3.json", "r03_pmc_c3.json", "r02_pmc_c3_edge16_v2.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue

        def rw(entry):
            r = next((v for k, v in entry.items() if k.startswith("hbm_read_bytes")), None)
            w = next((v for k, v in entry.items() if k.startswith("hbm_write_bytes")), None)
            return None if r is None or w is None else r + w

        # one processor block = the layer-1 kernel + the resident-weight kernel that writes e' (the last block's variant, which
        # drops e', is a separate instantiation and not part of the per-block figure)
        proc = [rw(v) for k, v in d.items() if "blocks1-8" in k and "edge16p_kernel<false>" not in k]
        dec = [rw(v) for k, v in d.items() if "[decoder]" in k]
        if proc and all(x is not None for x in proc):
            return {"processor_block": sum(proc), "decoder": sum(x for x in dec if x is not None) or None}, name
    return None, None


def build_model(cfg, dev):
    import graph_weather_amd as gw
    from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons

    lat_lons = regular_lat_lons(cfg["grid"])
    model = gw.GraphWeatherForecaster(lat_lons, resolution=cfg["resolution"])
    deterministic_fill_(model, seed=0)
    return model, lat_lons


DEFAULT_PATH_NOTE = ("model(features) in eval() under torch.no_grad(), as a caller writes it: from the third call of a shape on "
                     "graph_weather_amd replays the forward from one HIP graph (graphed.AutoGraph; same kernels, same arguments, "
                     "output cloned); kernel durations are HIP-event timed in a separate eager pass of the same process")


def time_forward(model, feats, steps, warmup, barrier, kernel_timer=True):
    """W untimed steps, then exactly K steps between barrier + synchronise.  ``kernel_timer``: HIP events around the tagged edge
    launches (an active timer keeps every call eager: events cannot be recorded into a HIP graph)."""
    from graph_weather_amd import ops

    with torch.no_grad():
        for _ in range(warmup):
            y = model(feats)
        ops.TIMER = ops.KernelTimer(["decoder_edge", "processor_edge", "encoder_edge"]) if kernel_timer else None
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(feats)
        barrier()
        elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    assert torch.isfinite(y).all()
    return elapsed, timer


def cold_step_ms(model, feats, n=3):
    """Forward right after every parameter's version changed: all per-weight-version caches miss (layers.py: packed weights,
    edge / mesh embeddings, their layer-1 products)."""
    from graph_weather_amd.optim import _bump_versions

    ts = []
    auto, model.auto_graph = getattr(model, "auto_graph", None), False  # eager: a re-capture per weight version is not the cold step
    with torch.no_grad():
        for _ in range(n):
            _bump_versions(model.parameters())  # what an optimizer step does to Tensor._version (no kernel launched)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(feats)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    if auto is not None:
        model.auto_graph = auto
    return 1e3 * sum(ts) / len(ts)


def eager_ms(model, feats, steps=20, repeats=3):
    """The same forward with the automatic HIP graph switched off (every launch issued from Python): median ms per step."""
    auto, model.auto_graph = getattr(model, "auto_graph", None), False
    ts = []
    with torch.no_grad():
        model(feats)
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model(feats)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0) / steps)
    if auto is not None:
        model.auto_graph = auto
    ts.sort()
    return {"ms_per_step": ts[len(ts) // 2], "ms_per_step_all": ts, "value": feats.shape[0] / (ts[len(ts) // 2] * 1e-3), "unit": "forecasts/s",
            "note": "model.auto_graph = False: ~50 launches per forward issued from Python (host-bound below ~3.5 ms of GPU work)"}


def kernel_report(graphs, batch, precision, timer, ms_per_step, stanza_timer=None):
    """roofline object: dominant kernel = decoder edge update (one C-ABI call).  Every figure is per TIMED LAUNCH: the batch
    elements a launch processed come from the timer (under per-sample streams a processor launch handles one sample, the
    decoder launch the whole batch).  ``stanza_timer``: a separate one-stream pass for the processor / encoder edge updates
    (when the timed region ran the mesh stack on several streams, a launch's HIP-event time includes co-scheduled work)."""
    # bf16x3: every product is three bf16 MFMAs - the peak for PRODUCT FLOPs is a third of the bf16 MFMA peak
    peak = {"fp32": PEAK_F32_MATRIX_TFLOPS, "bf16": PEAK_BF16_MATRIX_TFLOPS, "bf16x3": PEAK_BF16_MATRIX_TFLOPS / 3.0}[precision]
    e_dec, e_lat = graphs.dec_plan.num_edges, graphs.lat_plan.num_edges
    dec_ms = timer.mean_ms("decoder_edge")
    dec_b = timer.mean_units("decoder_edge")
    t2 = stanza_timer if stanza_timer is not None else timer
    proc_ms = t2.mean_ms("processor_edge")
    proc_b = t2.mean_units("processor_edge")
    executed = 2 * LAYER * e_dec * dec_b  # the two 256x256 layers; layer 1 is a gather-add of cached / per-node products
    algorithmic = EDGE_MLP_FLOPS * e_dec * dec_b
    ex = executed_flops_per_forecast(graphs)
    # algorithmic HBM bytes of one decoder edge launch: per (sample, edge) the cached product row and the residual edge-feature
    # row (2 x 1 KiB), per destination row one 1 KiB sum written; indices 8 B per edge
    alg_bytes = dec_b * e_dec * (2 * 1024 + 8) + dec_b * graphs.num_grid * 1024
    if precision == "bf16x3":
        # split mode: per (sample, edge) the fp32 product row of the source node and the cached per-edge product row (2 x 1 KiB); no
        # residual row (its sums enter the node update as a cached table); per destination one fp32 sum row
        alg_bytes = dec_b * e_dec * (2 * 1024 + 8) + dec_b * graphs.num_grid * 1024
    elif precision != "fp32":
        # bf16 path: per (sample, edge) the fp16 product row of the source node (512 B), the batch-shared cached product row once
        # per edge (1 KiB), no residual row (its sums enter the node update as a cached table); per destination one bf16 sum (512 B)
        alg_bytes = dec_b * e_dec * (512 + 8) + e_dec * 1024 + dec_b * graphs.num_grid * 512
    # gather / scatter stage of one processor block: 2 E D s + 2 M D 4 bytes per sample, s = bytes per stored
    # edge-feature element between blocks (4: fp32 rows; 2: bf16 edge tiles - "halve for bf16 storage")
    e_bytes = 2 if precision == "bf16" else 4
    gs_bytes = proc_b * (2 * e_lat * D * e_bytes + 2 * graphs.num_mesh * D * 4)
    gs_bytes_fp32 = proc_b * (2 * e_lat * D * 4 + 2 * graphs.num_mesh * D * 4)
    gs_bytes_halved = proc_b * (2 * e_lat * D * 2 + 2 * graphs.num_mesh * D * 2)  # SURVEY 8(d) "halve for bf16 storage", node tables too
    ach = executed / (dec_ms * 1e-3) / 1e12
    gs_traffic, gs_traffic_file, gs_busy = (None, None, None)
    if precision == "bf16" and proc_b == 16 and graphs.num_grid == 64800:
        t, gs_traffic_file = pmc_c3_traffic()
        gs_traffic = None if t is None else t["processor_block"]
    elif precision in ("fp32", "bf16x3") and graphs.num_grid == 64800 and proc_b:
        # the processor's edge-update instantiation at the grid of this launch (64 edges x 256 threads per workgroup)
        grid = ((int(round(proc_b)) * e_lat + 63) // 64) * 256
        pref = ("edge_kernel<true, 2>",) if precision == "fp32" else ("chainx3_kernel<8, true, 3, 16, 16, 1,",)
        gs_traffic, gs_traffic_file, gs_busy = pmc_kernel_traffic("c2" if precision == "fp32" else "c2x3", pref, grid)
    return {
        "bound": "mfma", "kernel": "decoder edge update (gw_edge_update_forward on the mesh->grid graph)",
        "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
        "basis": "EXECUTED matrix FLOPs per launch (2 x 256x256 layers per edge and sample) / HIP-event duration"
                 + ("; bf16x3 issues 3 bf16 MFMAs per product: peak = bf16 MFMA peak / 3" if precision == "bf16x3" else ""),
        "launch_ms": dec_ms, "launch_batch": dec_b, "executed_flops_per_launch": executed,
        "algorithmic_flops_per_launch": algorithmic, "algorithmic_tflops": algorithmic / (dec_ms * 1e-3) / 1e12,
        "algorithmic_frac": algorithmic / (dec_ms * 1e-3) / 1e12 / peak,
        "algorithmic_note": "reference edge MLP 768->256->256->256 per edge (SURVEY.md 8d); 60 % of it (layer 1) is removed by the "
                            "layer-1 split (x_dst == 0, per-node and cached per-edge products), so this ratio may exceed 1",
        "algorithmic_bytes_per_launch": alg_bytes, "hbm_tbs": alg_bytes / (dec_ms * 1e-3) / 1e12,
        "step_frac": ex["total"] * batch / (ms_per_step * 1e-3) / 1e12 / peak,
        "step_executed_gflop": ex["total"] * batch / 1e9,
        "other_kernels_ms": {"processor_edge": proc_ms, "encoder_edge": timer.mean_ms("encoder_edge")},
        "gather_scatter": {"kernel": "processor edge update (one block, one C-ABI call)", "launch_batch": proc_b,
                           "timing": "one-stream pass after the timed region" if stanza_timer is not None else "timed region",
                           "storage_bytes_per_edge_element": e_bytes,
                           "algorithmic_bytes": gs_bytes, "launch_ms": proc_ms,
                           "achieved_tbs": gs_bytes / (proc_ms * 1e-3) / 1e12, "peak_tbs": PEAK_HBM_TBS,
                           "frac": gs_bytes / (proc_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                           "frac_fp32_basis": gs_bytes_fp32 / (proc_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                           "frac_fully_halved_basis": None if precision != "bf16" else gs_bytes_halved / (proc_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                           "basis_note": "frac = storage basis (edge features as stored: bf16 tiles in bf16 mode; node tables fp32); "
                                         "frac_fully_halved_basis = SURVEY 8(d) bytes with every table halved; frac_fp32_basis = the "
                                         "reference's fp32 bytes",
                           "traffic": gs_traffic, "traffic_file": gs_traffic_file,
                           "traffic_over_algorithmic": None if gs_traffic is None else gs_traffic / gs_bytes,
                           "mfma_busy_frac_pmc": gs_busy,
                           "traffic_tbs": None if gs_traffic is None else gs_traffic / (proc_ms * 1e-3) / 1e12,
                           "executed_tflops": 3 * LAYER * e_lat * proc_b / (proc_ms * 1e-3) / 1e12},
    }


def run_extra(name, dev, steps, warmup, repeats=1, parity=False):
    """One of the other BASELINE configurations on this GPU (N = 1): value, ms_per_step, dominant kernel.  ``repeats`` > 1: the
    timed region is run that many times and the MEDIAN reported (all values listed) - boxes of the pool differ by 2-5 %.
    ``parity`` (16-bit modes): the same batch through the fp32 kernels of the same model; max |delta| / max |fp32 delta| of the
    decoder delta (out - input) - the fp32 kernels themselves are pinned to the oracle at 1e-6 (tests/test_gpu_round2.py)."""
    import gc

    from graph_weather_amd.utils import seeded_features

    cfg = CONFIGS[name]
    torch.cuda.init()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    model, lat_lons = build_model(cfg, dev)
    graphs = model.encoder.graphs
    build_s = time.perf_counter() - t0
    model = model.to(dev).eval()
    set_precision(model, cfg["precision"])
    feats = seeded_features(cfg["batch"], len(lat_lons), 102, seed=42).to(dev)
    gc.collect()
    runs = [time_forward(model, feats, steps, max(3, warmup) if i == 0 else 1, torch.cuda.synchronize, kernel_timer=False)
            for i in range(max(1, repeats))]
    runs.sort(key=lambda r: r[0])
    elapsed, _ = runs[len(runs) // 2]
    ms = 1e3 * elapsed / steps
    _, timer = time_forward(model, feats, min(steps, 5), 1, torch.cuda.synchronize)  # eager pass: HIP events on the edge launches
    r = kernel_report(graphs, cfg["batch"], cfg["precision"], timer, ms)
    traffic, pmc, pmc_file = pmc_traffic(name) if name in ("c2x3",) else (None, None, None)
    out = {"workload": f"{cfg['grid']:g}deg grid ({len(lat_lons)} nodes), mesh res {cfg['resolution']} ({graphs.num_mesh} nodes), "
                       f"batch {cfg['batch']}, {cfg['precision']}",
           "value": cfg["batch"] * steps / elapsed, "unit": "forecasts/s", "ms_per_step": ms, "steps": steps, "warmup": warmup,
           "repeats": len(runs), "ms_per_step_all": [1e3 * e / steps for e, _ in runs], "path": DEFAULT_PATH_NOTE,
           "graph_build_s": build_s, "dominant_kernel": r["kernel"], "launch_ms": r["launch_ms"], "frac": r["frac"], "peak": r["peak"],
           "step_frac": r["step_frac"], "other_kernels_ms": r["other_kernels_ms"], "gather_scatter_frac": r["gather_scatter"]["frac"],
           "gather_scatter": r["gather_scatter"],
           "algorithmic_gflop_per_forecast": algorithmic_flops_per_forecast(graphs) / 1e9,
           "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2**30}
    if traffic is not None:
        out["traffic"] = {"bytes_per_launch": traffic, "file": pmc_file, "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                          "mfma_busy_frac_pmc": pmc.get("mfma_busy_frac"), "lds_array_frac_pmc": pmc.get("lds_array_frac")}
    if parity and cfg["batch"] <= 2 and cfg["grid"] >= 1.0:
        out["cold_ms_per_step"] = cold_step_ms(model, feats)  # forward right after a weight update (every cache misses)
    if parity and cfg["grid"] >= 1.0:
        out["eager"] = eager_ms(model, feats, steps=steps)
        out["graph"] = graph_replay_ms(model, feats, steps=steps)
    if parity and cfg["batch"] <= 2 and cfg["grid"] >= 1.0 and cfg["precision"] != "fp32":
        out["h2d"] = h2d_step_ms(model, feats)  # the batch handed over from pinned host memory every step (PCIe), this mode
    if parity and cfg["precision"] != "fp32":
        with torch.no_grad():
            y = model(feats)
            set_precision(model, "fp32")
            y32 = model(feats)
        d32 = (y32 - feats[..., :78]).double()
        err = (y.double() - y32.double()).abs().max().item() / max(d32.abs().max().item(), 1e-30)
        out["parity_vs_fp32_kernels"] = {"max_rel_of_delta_scale": err, "bar": 1e-3, "inside_bar": err <= 1e-3,
                                         "note": "same model, same batch, fp32-MFMA kernels (pinned to the CPU oracle at ~1e-6); the "
                                                 "oracle comparisons of this mode are tests/test_gpu_split.py"}
        del y, y32, d32
    del model, feats
    gc.collect()
    torch.cuda.empty_cache()
    return out


def c4_per_gpu(dev):
    """BASELINE configs[3] (global batch 64 over 8 GPUs) as far as one GPU can show it: the per-GPU load - batch 8 at 1 degree -
    on this GPU, fp32 and bf16x3.  The forward has no collective, so the stated prediction for the 8-GPU
    node is 8 x this figure; the driver's SCALE record is the measurement."""
    out = {}
    for name, cfg in (("fp32", "c4g"), ("bf16x3", "c4gx3")):
        r = run_extra(cfg, dev, steps=10, warmup=3, repeats=3)
        out[name] = {k: r[k] for k in ("workload", "value", "unit", "ms_per_step", "ms_per_step_all", "step_frac", "launch_ms", "frac",
                                       "peak_memory_gb")}
        out[name]["predicted_8gpu_forecasts_per_s"] = 8.0 * r["value"]
    out["note"] = ("per-GPU shard of c4 (8 of 64 samples); prediction = 8 x the one-GPU rate: batch elements never interact and the "
                   "forward issues no collective, so the only multi-GPU costs are launch skew between ranks and host contention")
    return out


def graph_replay_ms(model, feats, steps=20, repeats=3):
    """The same forward replayed from ONE HIP graph (graph_weather_amd.ForwardGraph: every launch of every stream captured once;
    the batch is copied into the graph's input buffer in front of each replay): median ms per step of `repeats` x `steps`."""
    import graph_weather_amd as gw

    fg = gw.ForwardGraph(model)
    with torch.no_grad():
        y = fg(feats)
        ref = model(feats)
        err = (y - ref).abs().max().item() / max((ref - feats[..., :78]).abs().max().item(), 1e-30)
        ts = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fg(feats)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0) / steps)
    ts.sort()
    return {"ms_per_step": ts[len(ts) // 2], "ms_per_step_all": ts, "value": feats.shape[0] / (ts[len(ts) // 2] * 1e-3), "unit": "forecasts/s",
            "max_rel_vs_eager": err, "note": "whole forward as one HIP graph (same kernels, same arguments; the caller's input buffer is "
                                             "read in place once it has been handed over three times in a row); an explicit ForwardGraph object returning its own output buffer - `value` is model(features) itself, which replays the same way "
                                             "and clones the output)"}


def set_precision(model, precision: str) -> None:
    model.set_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x3": "bf16x3"}[precision])


def h2d_step_ms(model, feats, steps=20):
    """DESIGN.md PCIe note as a number: the c2 step when the batch starts in pinned HOST memory.  serial = copy, then forward,
    every step on one stream; overlapped = the copy of step i + 1 on a copy stream under the forward of step i (two device
    buffers).  ``value`` of the bench line never includes a host copy (inputs are HBM resident in the timed region)."""
    host = feats.cpu().pin_memory()
    dev = feats.device
    bufs = [torch.empty_like(feats), torch.empty_like(feats)]
    main = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(device=dev)
    res = {}
    with torch.no_grad():
        for _ in range(2):
            bufs[0].copy_(host, non_blocking=True)
            model(bufs[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bufs[0].copy_(host, non_blocking=True)
            model(bufs[0])
        torch.cuda.synchronize()
        res["serial_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        with torch.cuda.stream(copy):
            bufs[0].copy_(host, non_blocking=True)
            ready[0].record(copy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            cur, nxt = i & 1, (i + 1) & 1
            with torch.cuda.stream(copy):
                if i >= 1:
                    copy.wait_event(done[nxt])  # the forward that read this buffer two steps ago has finished
                bufs[nxt].copy_(host, non_blocking=True)
                ready[nxt].record(copy)
            main.wait_event(ready[cur])
            model(bufs[cur])
            done[cur].record(main)
        torch.cuda.synchronize()
        res["overlapped_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps
    res["bytes_per_step"] = host.numel() * 4
    res["note"] = ("input batch copied from pinned host memory every step (PCIe); not part of `value`.  The overlapped figure depends on "
                   "the copy stream getting a hardware queue of its own: HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) queues in "
                   "order of creation, and a later stream shares one with an earlier - possibly a busy mesh chain's "
                   "(profiles/r06_bench_hw_queues.log: 7.8 ms with 8 queues, no gain over the serial loop with 4)")
    return res


def run_wide(dev, steps=3, warmup=3):
    """The reference training script's model widths (train/run.py:493-497: nodes, edges, hidden layers and decoder 1024 wide) on
    the 1 degree grid, batch 1: the layer-by-layer path of graph_weather_amd/wide.py (generic fp32-MFMA kernels, nothing fused).
    FLOPs are the reference's own arithmetic for those widths (no layer-1 split there; the decoder's zero operand is skipped)."""
    import gc

    import graph_weather_amd as gw
    from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

    W = 1024
    torch.cuda.init()
    torch.cuda.reset_peak_memory_stats(dev)
    lat_lons = regular_lat_lons(1.0)
    model = gw.GraphWeatherForecaster(lat_lons, edge_dim=W, hidden_dim_processor_edge=W, node_dim=W, hidden_dim_processor_node=W,
                                      hidden_dim_decoder=W)
    deterministic_fill_(model, seed=0)
    model = model.to(dev).eval()
    g = model.encoder.graphs
    G, M = g.num_grid, g.num_mesh
    e_enc, e_lat, e_dec = g.enc_plan.num_edges, g.lat_plan.num_edges, g.dec_plan.num_edges
    mlp = lambda i, h, o, rows: 2.0 * rows * (i * h + h * h + h * o)  # noqa: E731
    flops = (mlp(102, W, W, G) + mlp(3 * W, W, W, e_enc) + mlp(2 * W, W, W, M) + 9 * (mlp(3 * W, W, W, e_lat) + mlp(2 * W, W, W, M))
             + mlp(2 * W, W, W, e_dec) + mlp(W, W, W, G) + mlp(W, W, 78, G))
    # what the wide path executes per forecast in inference: layer 1 split (node products once per node, gathered per edge),
    # batch-independent embeddings and their products cached per weight version (not in the count, as for the fused path)
    u = 2.0 * W * W
    executed = (mlp(102, W, W, G) + G * u + 2 * e_enc * u + 3 * M * u
                + sum(6 * M * u + (2 if b == 0 else 3) * e_lat * u for b in range(9))
                + M * u + 2 * e_dec * u + 3 * G * u + mlp(W, W, 78, G))
    feats = seeded_features(1, len(lat_lons), 102, seed=42).to(dev)
    elapsed, _ = time_forward(model, feats, steps, warmup, torch.cuda.synchronize, kernel_timer=False)
    ms = 1e3 * elapsed / steps
    rate = lambda f: f / (ms * 1e-3) / 1e12  # noqa: E731
    out = {"workload": "1deg grid, widths 1024 (train/run.py:493-497), batch 1, fp32, wide (layer-by-layer) path",
           "value": steps / elapsed, "unit": "forecasts/s", "ms_per_step": ms, "steps": steps, "warmup": warmup,
           "executed_gflop_per_forecast": executed / 1e9, "executed_tflops": rate(executed),
           "step_frac": rate(executed) / PEAK_F32_MATRIX_TFLOPS,
           "algorithmic_gflop_per_forecast": flops / 1e9, "algorithmic_tflops": rate(flops),
           "algorithmic_frac": rate(flops) / PEAK_F32_MATRIX_TFLOPS,
           "algorithmic_note": "reference arithmetic for these widths (cat[x_s, x_d, e] through the first Linear of every edge); the "
                               "layer-1 split executes 63 % of it, so this ratio may approach or exceed 1",
           "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2**30}
    del model, feats
    gc.collect()
    torch.cuda.empty_cache()
    return out


def train_bench(args, cfg, batch, model, feats, lat_lons, dev, world, rank, backend="nccl", train_factories=None):
    """Training step of the same workload (not the BASELINE metric; reported with its own metric name): forward under
    autograd, NormalizedMSELoss, backward into a flat gradient buffer, bucketed gradient all-reduce across ranks (RCCL), one
    multi-tensor AdamW launch.  ``train_factories`` = (loss factory, optimizer factory) stand-ins for the world-2 gloo test on
    CPU (the HIP loss / AdamW kernels cannot run there); the loop itself - flat buffer, hooks, allreduce, step - is this one."""
    from graph_weather_amd import sharding as sh

    ctx = sh.ShardContext(rank, int(os.environ.get("LOCAL_RANK", "0")), world, backend if world > 1 else None)
    model.train()
    flat = sh.FlatGradients(model.parameters()).attach(ctx)
    if train_factories is None:
        import graph_weather_amd as gw

        crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons, normalize=False)
        opt = gw.AdamW(model.parameters(), lr=1e-4, flat=flat)
    else:
        crit = train_factories[0](lat_lons)
        opt = train_factories[1](model.parameters(), flat)
    target = torch.randn(batch, len(lat_lons), 78, device=dev)

    def step():
        flat.zero_()
        loss = crit(model(feats), target)
        loss.backward()
        flat.allreduce(ctx)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    sh.barrier(ctx, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sh.barrier(ctx, dev)
    elapsed = sh.max_over_ranks(ctx, time.perf_counter() - t0, dev)
    assert torch.isfinite(loss)
    total = sh.sum_over_ranks(ctx, batch, dev)
    if rank == 0:
        print(json.dumps({
            "metric": "training samples/sec (1° grid, 102→78 feat): forward + loss + backward + grad all-reduce + AdamW",
            "value": total * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.config == "c4" else "weak",
            "vs_baseline": None, "dtype": "f32" if cfg["precision"] == "fp32" else cfg["precision"] + " products, fp32 saves / weight gradients / optimizer",
            "data": "synthetic",
            "config": {"workload": f"GraphWeatherForecaster {cfg['grid']:g}deg training step ({args.config}), batch={batch} on rank 0, {cfg['precision']}",
                       "global_batch": int(total), "parallelism": f"data parallel x{world}, bucketed gradient all-reduce (RCCL) on a flat buffer"},
            "collectives_per_step": len(flat.buckets) if world > 1 else 0,
            "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2**30 if dev.type == "cuda" else None}), flush=True)
    sh.shutdown(ctx)


def run_train_extra(dev, steps=5, warmup=2, precision="fp32"):
    """The training step of the c2 workload (1 degree, batch 2) measured in the driver's run: forward under autograd,
    NormalizedMSELoss, backward into the flat gradient buffer, one-launch AdamW (N = 1: no collective).  precision "bf16x3" =
    mixed precision: split-operand products in the forward and in the backward's input-gradient products; activation saves,
    weight-gradient GEMMs, LayerNorm / ReLU backward, master weights and AdamW in fp32."""
    import gc

    import graph_weather_amd as gw
    from graph_weather_amd import sharding as sh
    from graph_weather_amd.utils import seeded_features

    cfg = CONFIGS["c2"]
    torch.cuda.reset_peak_memory_stats(dev)
    model, lat_lons = build_model(cfg, dev)
    model = model.to(dev).train()
    if precision != "fp32":
        set_precision(model, precision)
    crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons, normalize=False)
    flat = sh.FlatGradients(model.parameters())
    opt = gw.AdamW(model.parameters(), lr=1e-4, flat=flat)
    feats = seeded_features(cfg["batch"], len(lat_lons), 102, seed=42).to(dev)
    target = torch.randn(cfg["batch"], len(lat_lons), 78, device=dev)

    def step():
        flat.zero_()
        loss = crit(model(feats), target)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss)
    out = {"workload": f"c2 training step: 1deg grid, batch 2, {precision}: forward + NormalizedMSELoss + backward + AdamW (one GPU)",
           "ms_per_step": 1e3 * elapsed / steps, "value": cfg["batch"] * steps / elapsed, "unit": "samples/s", "steps": steps,
           "warmup": warmup, "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2**30}
    del model, feats, target, flat, opt
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _rank_entry(rank, world, rendezvous, argv, backend, device, model_factory, train_factories):
    # the self-launched ranks meet through a file store (a fresh path of this launch): no TCP port to pick, no window in which
    # another process can take a port between "found free" and "bound by rank 0"
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), GW_BENCH_INIT_FILE=rendezvous)
    main(argv, backend=backend, device=device, model_factory=model_factory, train_factories=train_factories)


def self_launch(args, argv, backend, device, model_factory, train_factories) -> None:
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): build the graphs ONCE here, then fork N ranks
    (RANK / LOCAL_RANK / WORLD_SIZE set as torch.distributed.run would; rendezvous through a file store of this launch) that each run main() - one
    process per GPU over RCCL, rank 0 prints the one JSON line on the inherited stdout.  The fork happens before this process
    has touched the HIP runtime, so every child creates its own context on its own device; the children find the index arrays
    of the three graphs in the forked memory (graphs.build_forecast_graphs keeps the last builds) instead of rebuilding them
    per rank.  Exits non-zero if any rank fails (the others are terminated: a dead peer would leave them in a collective)."""
    import multiprocessing as mp

    from graph_weather_amd.graphs import build_forecast_graphs
    from graph_weather_amd.utils import regular_lat_lons

    cfg = dict(CONFIGS[args.config])
    if args.grid is not None:
        cfg["grid"] = args.grid
    if model_factory is None:
        build_forecast_graphs(regular_lat_lons(cfg["grid"]), cfg["resolution"])  # memoised: the ranks inherit it
    else:
        model_factory(cfg, "cpu")  # a stand-in factory builds (and thereby memoises) whatever graphs it uses
    import tempfile

    rdv_dir = tempfile.mkdtemp(prefix="gw_bench_rdv_")
    rendezvous = os.path.join(rdv_dir, "store")
    sys.stdout.flush()
    ctx = mp.get_context("fork")
    raw = list(sys.argv[1:] if argv is None else argv)
    procs = [ctx.Process(target=_rank_entry, args=(r, args.gpus, rendezvous, raw, backend, device, model_factory, train_factories))
             for r in range(args.gpus)]
    for p in procs:
        p.start()
    failed = None
    alive = set(range(len(procs)))
    while alive and failed is None:
        for r in sorted(alive):
            procs[r].join(timeout=0.2)
            if procs[r].exitcode is not None:
                alive.discard(r)
                if procs[r].exitcode != 0:
                    failed = (r, procs[r].exitcode)
                    break
    if failed is not None:
        for r in alive:
            procs[r].terminate()  # (exact children of this process)
        for p in procs:
            p.join(timeout=10)
    import shutil

    shutil.rmtree(rdv_dir, ignore_errors=True)
    if failed is not None:
        raise SystemExit("bench.py: rank %d of %d exited with code %s" % (failed[0], args.gpus, failed[1]))


def main(argv=None, backend="nccl", device=None, model_factory=None, train_factories=None):
    """``backend`` / ``device`` / ``model_factory`` / ``train_factories`` exist for the world-2 gloo tests (tests/test_sharding.py),
    which drive this very function on CPU with a stand-in model: rank / launch / barrier / max-over-ranks / JSON path (and,
    with --mode train, the flat-buffer / hook / all-reduce / optimizer loop) are then what is tested."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="BASELINE.json configuration: c2 (default, the metric's config: 1deg fp32 batch 2 per GPU), c3 (bf16, batch 16 "
                         "per GPU), c4 (global batch 64 sharded over the ranks), c5 (0.25deg, mesh res 3, batch 1 per GPU); c2x3 / c3x3 = "
                         "c2 / c3's batch with split-operand (bf16x3) products")
    ap.add_argument("--grid", type=float, default=None, help="override the grid spacing in degrees")
    ap.add_argument("--batch", type=int, default=None, help="override the batch per GPU")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3"], default=None, help="override the matrix-product dtype")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward = BASELINE metric (default); train = forward + loss + backward + gradient all-reduce + AdamW")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams of the mesh stack in the forward (0 = automatic: per-sample "
                                                           "chains on 2 streams for fp32 and batch >= 2; 1 = one stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c3 / c5 measurements and the cold-cache step")
    args = ap.parse_args(argv)

    from graph_weather_amd import sharding as sh

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (see self_launch)
        return self_launch(args, argv, backend, device, model_factory, train_factories)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a launcher with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    on_gpu = device is None
    if on_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device(device)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init = {}
        if os.environ.get("GW_BENCH_INIT_FILE"):  # ranks forked by self_launch: file store instead of MASTER_ADDR / MASTER_PORT
            init = dict(init_method="file://" + os.environ["GW_BENCH_INIT_FILE"], world_size=world, rank=rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, **init)
        else:
            dist.init_process_group(backend=backend, **init)

    from graph_weather_amd.utils import seeded_features

    cfg = dict(CONFIGS[args.config])
    if args.grid is not None:
        cfg["grid"] = args.grid
    if args.precision is not None:
        cfg["precision"] = args.precision
    if args.config == "c4":  # global batch sharded over the ranks (8 per GPU at N = 8): strong scaling
        lo, hi = sh.shard_range(cfg["batch"] if args.batch is None else args.batch, world, rank)
        batch = hi - lo
    else:
        batch = cfg["batch"] if args.batch is None else args.batch
    model, lat_lons = (model_factory or build_model)(cfg, dev)
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    graphs = model.encoder.graphs
    model = model.to(dev).eval()
    if cfg["precision"] != "fp32":
        set_precision(model, cfg["precision"])
    if args.streams and hasattr(model, "processor"):
        model.processor.graph_processor.streams = args.streams
    feats = seeded_features(batch, len(lat_lons), 102, seed=42 + rank).to(dev)  # resident in HBM before timing

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    if args.mode == "train":
        return train_bench(args, cfg, batch, model, feats, lat_lons, dev, world, rank, backend=backend, train_factories=train_factories)
    # A full Python garbage collection walks the model's large host-side containers (64 800 lat/lon tuples, grid
    # mappings) and takes ~130 ms - longer than 15 steps; freeze the existing heap so that no cyclic-GC pass over it
    # lands inside the timed region (standard practice for latency benchmarks; the steps themselves create no cycles).
    import gc

    gc.collect()
    gc.freeze()
    # The timed region is model(features) as a caller writes it (from the third call on a HIP-graph replay, DEFAULT_PATH_NOTE);
    # the per-kernel HIP events need eager launches: a separate pass right after it, in this process, on rank 0.
    elapsed, timer = time_forward(model, feats, args.steps, args.warmup, barrier, kernel_timer=False)
    timer_pass_ms = None
    if on_gpu and rank == 0:
        # a second timed region of the same K steps, same bracket, with HIP events on the stream each edge launch runs on
        t_el, timer = time_forward(model, feats, args.steps, 1, torch.cuda.synchronize)
        timer_pass_ms = 1e3 * t_el / args.steps
    # The kernel stanzas of the mesh stack: when the timed region ran it as per-sample chains on several HIP streams, an edge
    # launch's event time includes whatever the other stream ran beside it - time those launches again on ONE stream
    # (5 extra steps outside the timed region) so that bytes / FLOPs / duration belong to the same launch.
    stanza_timer = None
    gp = getattr(getattr(model, "processor", None), "graph_processor", None)
    if on_gpu and rank == 0 and gp is not None and gp.forward_streams(batch) > 1:
        keep = gp.streams
        gp.streams = 1
        _, stanza_timer = time_forward(model, feats, 5, 1, torch.cuda.synchronize)
        gp.streams = keep
    total_batch = batch
    per_rank_ms, dist_world = [1e3 * elapsed / args.steps], 1
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # (each rank's own clock over the barrier-bracketed region, for the record)
        per_rank_ms = [1e3 * float(e.item()) / args.steps for e in every]
        dist_world = dist.get_world_size()
        t = torch.tensor([elapsed, float(batch)], dtype=torch.float64, device=dev)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        elapsed, total_batch = float(t[0].item()), int(round(t[1].item()))

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        prec = cfg["precision"]
        roof = None
        if timer is not None:
            roof = kernel_report(graphs, batch, prec, timer, ms, stanza_timer)
            is_c2 = (cfg["grid"] == 1.0 and batch == 2 and prec == "fp32")
            traffic, pmc, pmc_file = pmc_traffic() if is_c2 else (None, None, None)
            roof["traffic"] = traffic
            roof["traffic_file"] = pmc_file
            roof["traffic_unit"] = ("bytes per launch, read from the newest tracked rocprofv3 PMC summary of the same workload "
                                    "(profiles/" + str(pmc_file) + "; counters are not collected inside the bench run)")
            roof["mfma_busy_frac_pmc"] = None if pmc is None else pmc.get("mfma_busy_frac")
        out = {
            "metric": "forward forecasts/sec (1° grid, 102→78 feat)", "value": total_batch * args.steps / elapsed,
            "unit": "forecasts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if args.config == "c4" else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 (MFMA operands; fp32 accumulate, LayerNorm, residuals, sums)",
                      "bf16x3": "bf16x3 (hi/lo bf16 operand pairs, 3 MFMAs per product; fp32 accumulate, LayerNorm, residuals, sums)"}[prec],
            "data": "synthetic",
            "config": {"workload": f"{args.config}: GraphWeatherForecaster {cfg['grid']:g}deg grid ({len(lat_lons)} nodes), 102->78 feat, "
                                   f"batch={batch} on rank 0, {prec}, mesh res {cfg['resolution']} ({graphs.num_mesh} nodes), random-init weights",
                       "global_batch": total_batch, "parallelism": f"batch-sharded x{world}, no collective in forward"},
            "roofline": roof, "path": DEFAULT_PATH_NOTE if on_gpu else None,
            "kernel_timer_pass_ms_per_step": timer_pass_ms,  # the K-step region the roofline's launch durations were HIP-event timed in
            "algorithmic_gflop_per_forecast": algorithmic_flops_per_forecast(graphs) / 1e9,
            "per_rank_ms_per_step": per_rank_ms, "dist_world_size": dist_world,
        }
        if world == 1 and not args.no_extra and on_gpu:
            out["cold_ms_per_step"] = cold_step_ms(model, feats)
            out["cold_note"] = ("forward right after every parameter changed: packed weights, edge / mesh embeddings and their layer-1 "
                                "products are rebuilt (the reference recomputes the embeddings on every forward)")
        if world == 1 and not args.no_extra and args.config == "c2" and on_gpu:
            out["eager"] = eager_ms(model, feats, steps=args.steps)
            out["graph"] = graph_replay_ms(model, feats, steps=args.steps)
            h2d = h2d_step_ms(model, feats)
            del model, feats
            gc.collect()
            torch.cuda.empty_cache()
            out["extra"] = {"c2_split": run_extra("c2x3", dev, steps=20, warmup=3, repeats=3, parity=True),
                            "c3_split": run_extra("c3x3", dev, steps=20, warmup=3, repeats=3, parity=True),
                            "c3": run_extra("c3", dev, steps=20, warmup=3, repeats=3, parity=True),
                            "c5": run_extra("c5", dev, steps=5, warmup=2),
                            "c5_split": run_extra("c5x3", dev, steps=5, warmup=2, parity=True),
                            "wide1024": run_wide(dev), "train": run_train_extra(dev), "train_split": run_train_extra(dev, precision="bf16x3"),
                            "h2d": h2d, "c4_per_gpu": c4_per_gpu(dev)}
            out["extra"]["c3"]["status"] = ("plain bf16 operands: OUTSIDE the 1e-3 parity bar (parity_vs_fp32_kernels); a frozen budget "
                                            "mode - kernels unchanged since round 4, no new work is routed through it")
        if world == 1 and not args.no_cpu_baseline and cfg["grid"] == 1.0:
            out["cpu_baseline"] = cpu_baseline(lat_lons, cpu_state, graphs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _cli():
    """Entry of `python bench.py ...`.  GW_BENCH_BACKEND / GW_BENCH_DEVICE / GW_BENCH_FACTORY ("module:function") select the
    process-group backend, a non-GPU device and a stand-in model factory: the hooks tests/test_sharding.py uses to run this
    very command line (self-launch included) on CPU over gloo."""
    kw = {}
    if (os.environ.get("GW_BENCH_BACKEND") or os.environ.get("GW_BENCH_FACTORY")) and os.environ.get("GW_BENCH_DEVICE") != "cpu":
        # the stand-in hooks exist for the CPU tests only: on a GPU device the line printed must come from the real model over RCCL
        raise SystemExit("bench.py: GW_BENCH_BACKEND / GW_BENCH_FACTORY are honoured only together with GW_BENCH_DEVICE=cpu")
    if os.environ.get("GW_BENCH_BACKEND"):
        kw["backend"] = os.environ["GW_BENCH_BACKEND"]
    if os.environ.get("GW_BENCH_DEVICE"):
        kw["device"] = os.environ["GW_BENCH_DEVICE"]
    if os.environ.get("GW_BENCH_FACTORY"):
        import importlib

        mod, fn = os.environ["GW_BENCH_FACTORY"].split(":")
        kw["model_factory"] = getattr(importlib.import_module(mod), fn)
    main(**kw)


if __name__ == "__main__":
    _cli()
