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


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3", torch.bfloat16])
def test_forward_graph_replays_the_eager_forward_and_follows_weight_updates(dtype):
    """graph_weather_amd.ForwardGraph: the whole inference forward (every launch of both HIP streams) as one HIP graph.  Equal to the
    eager forward (bitwise in deterministic mode); re-captured by itself after a weight update, a compute-dtype switch and a new
    batch size - a stale graph would keep multiplying by the OLD packed weights."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    model.set_compute_dtype(dtype)
    tol = 5e-3 if dtype == torch.bfloat16 else 1e-5
    feats = seeded_features(2, len(lat_lons), 102, seed=42).to(DEV)
    fg = model.graphed()
    with torch.no_grad():
        y = fg(feats).clone()
        assert fg.captures == 1
        ref = model(feats)
        assert _rel(y - feats[..., :78], ref - feats[..., :78]) <= tol
        other = seeded_features(2, len(lat_lons), 102, seed=7).to(DEV)
        y2 = fg(other).clone()  # same shape: replay only
        assert fg.captures == 1
        assert _rel(y2 - other[..., :78], model(other) - other[..., :78]) <= tol
        assert not torch.equal(y2, y)
        for p in model.parameters():  # an optimizer step's worth of change
            p.mul_(1.01)
        y3 = fg(feats).clone()
        assert fg.captures == 2
        ref3 = model(feats)
        assert _rel(y3 - feats[..., :78], ref3 - feats[..., :78]) <= tol
        assert _rel(y3, y) > 1e-4  # (the new weights really are in the graph)
        y4 = fg(feats[:1].contiguous()).clone()  # new batch size
        assert fg.captures == 3 and y4.shape[0] == 1
        assert _rel(y4 - feats[:1, :, :78], ref3[:1] - feats[:1, :, :78]) <= tol
        if dtype != torch.bfloat16:
            model.set_deterministic(True)
            a = fg(feats).clone()
            assert fg.captures == 4
            assert torch.equal(a, model(feats)) and torch.equal(a, fg(feats))
    model.train()
    with pytest.raises(RuntimeError, match="inference"):
        fg(feats)
