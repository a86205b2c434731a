"""Round-5 GPU tests outside the split-operand mode (tests/test_gpu_split.py): findings of the round-4 review."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features  # noqa: E402

DEV = "cuda:0"


def _rel(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


@pytest.mark.parametrize("dtype,budget", [(torch.bfloat16, 3e-2), ("bf16x3", 2e-4)])
@pytest.mark.parametrize("batch,efficient", [(1, False), (2, False), (2, True)])
def test_processor_forward_with_per_sample_edge_features_in_16_bit_modes(dtype, budget, batch, efficient):
    """``Processor.forward(x, edge_index, edge_attr)`` without efficient batching hands PER-SAMPLE edge
    features to block 0 on the latent mesh (<= 16 destinations per tile: the graph on which the bf16 stack would take the
    segment-aligned route).  Block 0's aggregate must be scatter_sum(LN(.) + e) (graph_net_block.py:131-137,188) - the
    segment-tile route starts its running sum from the cached sums of batch-SHARED features only, so this call must not take
    it.  Compared with the fp32 kernels on the same weights and inputs (themselves pinned to the oracle): a missing sum(e) term
    is an O(1) error, far outside either budget."""
    lat_lons = regular_lat_lons(30.0)
    enc = gw.Encoder(lat_lons, input_dim=102)
    proc = gw.Processor()
    deterministic_fill_(enc, seed=6)
    deterministic_fill_(proc, seed=7)
    enc, proc = enc.to(DEV).eval(), proc.to(DEV).eval()
    enc.efficient_batching = efficient
    feats = seeded_features(batch, len(lat_lons), 102, seed=3).to(DEV)
    with torch.no_grad():
        x, ei, ea = enc(feats)  # replicated graph (reference default) or the single shared graph
        kw = dict(batch_size=batch, efficient_batching=True) if efficient else {}
        ref = proc(x, ei, ea, **kw)
        gw.set_compute_dtype(proc, dtype)
        out = proc(x, ei, ea, **kw)
        out2 = proc(x.clone(), ei, ea.clone(), **kw)
    r = _rel(out, ref)
    print(f"[processor.forward {dtype} batch {batch} efficient={efficient}] max-rel vs fp32 kernels {r:.2e}")
    assert r <= budget
    assert _rel(out2, out) <= max(budget / 4, 1e-5)


def test_bf16_forward_on_two_streams_builds_its_shared_tiles_before_the_fork():
    """with ``gp.streams > 1`` in bf16 mode the per-sample chains run on side streams; the padded shared
    products / tile set of block 0 must exist before the fork (GraphProcessor.prepare_shared), or the first forward after a
    weight change races.  The first forward after each weight change equals the steady-state one."""
    model = gw.GraphWeatherForecaster(regular_lat_lons(30.0))
    deterministic_fill_(model, seed=5)
    model = model.to(DEV).eval()
    model.set_compute_dtype(torch.bfloat16)
    gp = model.processor.graph_processor
    gp.streams = 2
    feats = torch.randn(4, 72, 102, device=DEV)
    with torch.no_grad():
        for trial in range(3):
            for p in model.parameters():
                p.add_(0)  # new weight version: every cache misses
            first = model(feats)
            assert gp._e0_seg_cache is not None
            steady = model(feats)
            assert _rel(first, steady) <= 5e-3, f"trial {trial}"
        gp.streams = 1
        one = model(feats)
    assert _rel(steady, one) <= 5e-3
