"""Models wider than the fused kernels' 256 features (the reference's training script builds 1024-wide ones,
train/run.py:493-497): ``graph_weather_amd/wide.py`` over ``csrc/gw_wide.hip``.  Kernels against fp64 torch on the CPU at
ragged shapes, autograd nodes against torch autograd, modules and the forecaster against the oracle (forward 1e-5 on the
delta scale like the native path; gradients with the bars of tests/test_gpu_backward.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd import wide  # noqa: E402
from graph_weather_amd.graphs import plan_from_coo  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons  # noqa: E402
from oracle import reference_math as om  # noqa: E402

from .test_gpu_backward import _check_param_grads, _rel  # noqa: E402
from .test_gpu_parity import _close  # noqa: E402

DEV = "cuda:0"


def _t(rs, *shape, scale=1.0):
    return torch.from_numpy((scale * rs.standard_normal(shape)).astype(np.float32))


@pytest.mark.parametrize("rows,k,n,relu,bias", [(1000, 102, 300, True, True), (777, 3072, 1024, True, True), (1, 2, 512, False, True),
                                                (130, 1024, 78, False, False), (129, 257, 129, True, True), (4096, 512, 512, False, True)])
def test_linear_forward_against_float64(rows, k, n, relu, bias):
    rs = np.random.RandomState(rows + k)
    x, w = _t(rs, rows, k), _t(rs, n, k, scale=1.0 / np.sqrt(k))
    b = _t(rs, n) if bias else None
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    if relu:
        ref = torch.relu(ref)
    got = wide.linear_forward(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), relu).cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, f"linear {rows}x{k}->{n}: {err:.2e}"


def test_linear_forward_with_row_strides():
    """Operands that are column slices of wider tensors (row stride > width): the decoder's residual, weight column ranges."""
    rs = np.random.RandomState(3)
    xw, ww = _t(rs, 300, 700), _t(rs, 260, 900, scale=0.05)
    x, w = xw[:, 100:612], ww[:, 4:516]
    ref = x.double() @ w.double().t()
    got = wide.linear_forward(xw.to(DEV)[:, 100:612], ww.to(DEV)[:, 4:516], None, False).cpu().double()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-6


@pytest.mark.parametrize("rows,width,res", [(333, 300, True), (1000, 1024, True), (5, 4096, False), (64, 257, False), (77, 78, True)])
def test_layernorm_forward_and_backward_against_torch(rows, width, res):
    rs = np.random.RandomState(width)
    y, g, b = _t(rs, rows, width), 1 + _t(rs, width, scale=0.1), _t(rs, width, scale=0.1)
    r = _t(rs, rows, width) if res else None
    dout = _t(rs, rows, width)
    yr, gr, br = (t.double().requires_grad_(True) for t in (y, g, b))
    ref = torch.nn.functional.layer_norm(yr, (width,), gr, br, 1e-5) + (0 if r is None else r.double())
    ref.backward(dout.double())
    yd, gd, bd = (t.to(DEV).requires_grad_(True) for t in (y, g, b))
    rd = None if r is None else r.to(DEV).requires_grad_(True)
    out = wide._LayerNorm.apply(yd, gd, bd, rd, 0)
    out.backward(dout.to(DEV))
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5
    assert _rel(yd.grad, yr.grad) < 1e-4 and _rel(gd.grad, gr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4
    if rd is not None:
        assert torch.equal(rd.grad.cpu(), dout)


def test_layernorm_with_a_residual_shared_by_the_batch():
    rs = np.random.RandomState(12)
    B, E, W = 3, 50, 300
    y, g, b, r, dout = _t(rs, B * E, W), 1 + _t(rs, W, scale=0.1), _t(rs, W, scale=0.1), _t(rs, E, W), _t(rs, B * E, W)
    yr, rr = y.double().requires_grad_(True), r.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(yr, (W,), g.double(), b.double(), 1e-5) + rr.repeat(B, 1)
    ref.backward(dout.double())
    yd, rd = y.to(DEV).requires_grad_(True), r.to(DEV).requires_grad_(True)
    out = wide._LayerNorm.apply(yd, g.to(DEV), b.to(DEV), rd, E)
    out.backward(dout.to(DEV))
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5
    assert _rel(yd.grad, yr.grad) < 1e-4 and _rel(rd.grad, rr.grad) < 1e-5


@pytest.mark.parametrize("with_product", [True, False])
def test_linear_with_gathered_addends_forward_and_backward(with_product):
    """relu(e . We^T + b + Ps[src] + Pd[dst] (+ Pe[k])): per-sample, batch-shared and identity-indexed tables; k = 0 form."""
    rs = np.random.RandomState(21)
    B, n_src, n_dst, E, K, N = 3, 40, 25, 333, 200, 300
    plan = plan_from_coo(rs.randint(0, n_src, size=E), np.where(rs.rand(E) < 0.3, 7, rs.randint(0, n_dst, size=E)), n_src, n_dst).to(DEV)
    s, d = plan.src.long().cpu(), plan.dst.long().cpu()
    x, w, b = _t(rs, B * E, K), _t(rs, N, K, scale=1 / np.sqrt(K)), _t(rs, N)
    ps, pd, pe, dout = _t(rs, B * n_src, N), _t(rs, n_dst, N), _t(rs, E, N), _t(rs, B * E, N)
    refs = [t.double().requires_grad_(True) for t in (x, w, b, ps, pd, pe)]
    xr, wr, br, psr, pdr, per = refs
    z = br + psr.reshape(B, n_src, N)[:, s].reshape(B * E, N) + pdr[d].repeat(B, 1) + per.repeat(B, 1)
    if with_product:
        z = z + xr @ wr.t()
    ref = torch.relu(z)
    ref.backward(dout.double())
    devs = [t.to(DEV).requires_grad_(True) for t in (x, w, b, ps, pd, pe)]
    xd, wd, bd, psd, pdd, ped = devs
    ident = torch.arange(E + 1, dtype=torch.int32, device=DEV)
    meta = ((plan.src, n_src, plan.src_sorted()), (plan.dst, 0, (None, plan.dst_ptr())), (None, 0, (None, ident)))
    out = wide._LinearGather.apply(xd if with_product else None, wd if with_product else None, bd, True, B * E, E, meta, psd, pdd, ped)
    out.backward(dout.to(DEV))
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() / ref.abs().max().item() < 2e-6
    pairs = [(bd, br), (psd, psr), (pdd, pdr), (ped, per)] + ([(xd, xr), (wd, wr)] if with_product else [])
    for got, want in pairs:
        assert _rel(got.grad, want.grad) < 1e-4


def test_gather_and_segment_sum_at_any_width_and_their_gradients():
    rs = np.random.RandomState(8)
    B, n_src, n_dst, E, W = 3, 40, 25, 333, 300
    src = rs.randint(0, n_src, size=E)
    dst = np.where(rs.rand(E) < 0.3, 7, rs.randint(0, n_dst, size=E))
    plan = plan_from_coo(src, dst, n_src, n_dst).to(DEV)
    table, shared, rows = _t(rs, B * n_src, W), _t(rs, n_src, W), _t(rs, B * E, W)
    s, d = plan.src.long().cpu(), plan.dst.long().cpu()
    # gather per sample / shared, with gradient
    for tb, rows_pb in ((table, n_src), (shared, 0)):
        td = tb.to(DEV).requires_grad_(True)
        got = wide._Gather.apply(td, plan.src, B, rows_pb, E, plan.src_sorted())
        tr = tb.double().requires_grad_(True)
        ref = tr.reshape(B, n_src, W)[:, s].reshape(B * E, W) if rows_pb else tr[s].repeat(B, 1)
        assert torch.equal(got.detach().cpu().double(), ref.detach())
        gw_ = _t(rs, B * E, W)
        got.backward(gw_.to(DEV))
        ref.backward(gw_.double())
        assert _rel(td.grad, tr.grad) < 1e-5
    # segment sum by destination, with gradient
    rd = rows.to(DEV).requires_grad_(True)
    got = wide._SegmentSum.apply(rd, plan, B)
    rr = rows.double().requires_grad_(True)
    ref = torch.zeros(B, n_dst, W, dtype=torch.float64).index_add(1, d, rr.reshape(B, E, W)).reshape(B * n_dst, W)
    assert (got.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-4
    g2 = _t(rs, B * n_dst, W)
    got.backward(g2.to(DEV))
    ref.backward(g2.double())
    assert _rel(rd.grad, rr.grad) < 1e-6
    again = wide._SegmentSum.apply(rd.detach(), plan, B)
    assert torch.equal(again, got.detach())  # one summation order: bitwise reproducible


@pytest.mark.parametrize("i,o,h,layers,norm", [(102, 320, 384, 2, "LayerNorm"), (300, 78, 300, 2, None), (1024, 1024, 1024, 2, "LayerNorm"),
                                                (2, 512, 64, 1, "LayerNorm"), (700, 100, 128, 3, "LayerNorm")])
def test_wide_mlp_forward_and_backward(i, o, h, layers, norm):
    m = gw.MLP(i, o, h, layers, norm)
    deterministic_fill_(m, seed=i + o)
    rs = np.random.RandomState(h)
    x, dy = _t(rs, 333, i), _t(rs, 333, o)
    ref = {"m." + k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    y_ref = om.mlp(ref, "m", xr)
    y_ref.backward(dy.double())
    m = m.to(DEV)
    with torch.no_grad():
        y = m(x.to(DEV))
    assert y.shape == (333, o)
    _close(y, y_ref, what=f"wide MLP {i}->{h}x{layers}->{o} (inference)")
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    y.backward(dy.to(DEV))
    worst = {}
    _check_param_grads(m, ref, "m.", worst)
    assert _rel(xd.grad, xr.grad) < 2e-3


def test_wide_graph_processor_random_coo():
    gp = gw.GraphProcessor(mp_iterations=2, in_dim_node=320, in_dim_edge=288, hidden_dim_node=384, hidden_dim_edge=300)
    deterministic_fill_(gp, seed=4)
    ref = {"gp." + k: v.detach().double().requires_grad_(True) for k, v in gp.state_dict().items()}
    rs = np.random.RandomState(5)
    n, e = 150, 900
    x, ea = _t(rs, n, 320), _t(rs, e, 288)
    ei = torch.from_numpy(np.stack([rs.randint(0, n, size=e), np.where(rs.rand(e) < 0.2, 3, rs.randint(0, n, size=e))]).astype(np.int64))
    gx, ge = _t(rs, n, 320), _t(rs, e, 288)
    xr, er = x.double().requires_grad_(True), ea.double().requires_grad_(True)
    xo_r, eo_r = om.graph_processor(ref, "gp", xr, ei, er)
    ((xo_r * gx.double()).sum() + (eo_r * ge.double()).sum()).backward()
    gp = gp.to(DEV)
    with torch.no_grad():
        xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
    assert xo.shape == (n, 320) and eo.shape == (e, 288)
    _close(xo, xo_r, what="wide GraphProcessor nodes")
    _close(eo, eo_r, what="wide GraphProcessor edges")
    xd, ed = x.to(DEV).requires_grad_(True), ea.to(DEV).requires_grad_(True)
    xo, eo = gp(xd, ei.to(DEV), ed)
    ((xo * gx.to(DEV)).sum() + (eo * ge.to(DEV)).sum()).backward()
    worst = {}
    _check_param_grads(gp, ref, "gp.", worst)
    assert _rel(xd.grad, xr.grad) < 2e-3 and _rel(ed.grad, er.grad) < 2e-3


def _wide_forecaster(lat_lons, **over):
    kw = dict(feature_dim=20, aux_dim=5, node_dim=320, edge_dim=288, num_blocks=2, hidden_dim_processor_node=384,
              hidden_dim_processor_edge=300, hidden_dim_decoder=272)
    kw.update(over)
    model = gw.GraphWeatherForecaster(lat_lons, **kw)
    deterministic_fill_(model, seed=9)
    return model


def test_wide_forecaster_matches_oracle_forward_and_backward():
    lat_lons = regular_lat_lons(15.0)
    model = _wide_forecaster(lat_lons)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    rs = np.random.RandomState(1)
    feats, dy = _t(rs, 2, len(lat_lons), 25), _t(rs, 2, len(lat_lons), 20)
    y_ref = om.forecaster_forward(ref, g64, feats.double(), feature_dim=20)
    (y_ref * dy.double()).sum().backward()
    model = model.to(DEV)
    with torch.no_grad():
        y = model(feats.to(DEV))
    res = feats[..., :20]
    _close(y.cpu() - res, y_ref.detach().float() - res, what="wide forecaster (inference)")
    model.train()
    y = model(feats.to(DEV))
    (y * dy.to(DEV)).sum().backward()
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)
    # compositional API at the real widths (tests/test_model.py:106-119)
    with torch.no_grad():
        x, ei, ea = model.encoder(feats.to(DEV))
        assert x.shape[1] == 320 and ea.shape[1] == 288
        out = model.decoder(model.processor(x, ei, ea), feats.to(DEV)[..., :20])
    _close(out, y, rel=1e-5, what="wide compositional")
    # checkpointing (forecast.py use_checkpointing, processor.py:70-81) changes memory, not results
    model.processor.set_checkpoint_segments(1)
    model.zero_grad()
    y2 = model(feats.to(DEV))
    (y2 * dy.to(DEV)).sum().backward()
    assert torch.equal(y2, y)
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)
    # inference caches of batch-independent embeddings / products follow the parameter versions
    model.eval()
    with torch.no_grad():
        y_a = model(feats.to(DEV))
        assert torch.equal(y_a, model(feats.to(DEV)))
        for prm in (model.decoder.edge_encoder.model[0].weight, model.encoder.h3_nodes, model.processor.graph_processor.blocks[0].edge_model.edge_mlp.model[0].weight):
            prm.add_(0.05)
        y_b = model(feats.to(DEV))
    assert (y_b - y_a).abs().max().item() > 1e-4
    model.train()
    _close(model(feats.to(DEV)).detach() - feats.to(DEV)[..., :20], y_b - feats.to(DEV)[..., :20], rel=1e-5, what="wide caches after a weight update")
    # no bf16 form of the wide path: loud
    with pytest.raises(NotImplementedError, match="bf16"):
        model.set_compute_dtype(torch.bfloat16)
        with torch.no_grad():
            model(feats.to(DEV))


def test_reference_training_script_widths_one_step():
    """The constructor call of the reference's train/run.py:479-501 as written - lat_lons as an [N, 2] numpy array, 605 + 40
    input channels, 605 outputs, nodes / edges / hidden layers / decoder 1024 wide, 6 blocks - on a 10 degree grid: forward
    against the oracle, then its loop body (:509-521: zero_grad, forward, NormalizedMSELoss, backward, torch.optim.AdamW.step)."""
    lat_lons = np.array(np.meshgrid(np.arange(-90.0, 90.0, 10.0), np.arange(0.0, 360.0, 10.0))).T.reshape(-1, 2)
    model = gw.GraphWeatherForecaster(lat_lons, edge_dim=1024, hidden_dim_processor_edge=1024, node_dim=1024,
                                      hidden_dim_processor_node=1024, hidden_dim_decoder=1024, feature_dim=605, aux_dim=40, num_blocks=6)
    deterministic_fill_(model, seed=2)
    ref = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = model.encoder.graphs.as_oracle_dict()
    rs = np.random.RandomState(2)
    n = len(lat_lons)
    inputs, labels = _t(rs, 1, n, 645), _t(rs, 1, n, 605)
    y_ref = om.forecaster_forward(ref, g, inputs, feature_dim=605)
    model = model.to(DEV)
    with torch.no_grad():
        y = model(inputs.to(DEV))
    res = inputs[..., :605]
    _close(y.cpu() - res, y_ref.float() - res, what="train/run.py forecaster (1024 wide, 645 -> 605)")
    criterion = gw.NormalizedMSELoss(lat_lons=lat_lons, feature_variance=[0.0] * 605, device=DEV).to(DEV)
    optimizer = torch.optim.AdamW(model.parameters(), lr=2e-5)
    model.train()
    losses = []
    for _ in range(4):
        optimizer.zero_grad()
        outputs = model(inputs.to(DEV))
        loss = criterion(outputs, labels.to(DEV))
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_wide_regional_forecaster_forward_and_backward():
    """RegionalForecaster (regional_forecast.py:234-298) at widths above 256: forward and every parameter gradient against the oracle."""
    rs = np.random.RandomState(17)
    n = 230
    lat_lons = [(float(a), float(b)) for a, b in zip(rs.uniform(35, 65, n), rs.uniform(-15, 30, n))]
    model = gw.RegionalForecasterConfig(feature_dim=20, aux_dim=5, node_dim=320, edge_dim=288, num_blocks=2, hidden_dim_processor_node=384,
                                        hidden_dim_processor_edge=300, hidden_dim_decoder=272, enable_nudging=True).build()
    deterministic_fill_(model, seed=12)
    feats, ctx, dy = _t(rs, 2, n, 25), _t(rs, 2, n, 20), _t(rs, 2, n, 20)
    enc, _, lat, h3_idx = model.graph_builder(lat_lons)
    g = {"enc_edge_index": enc.edge_index, "enc_edge_attr": enc.edge_attr, "lat_edge_index": lat.edge_index,
         "lat_edge_attr": lat.edge_attr, "h3_indices": h3_idx}
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    y_ref = om.regional_forward(ref, om.graphs_to_dtype(g, torch.float64), feats.double(), 20, global_context=ctx.double(), lat_lons=lat_lons)
    (y_ref * dy.double()).sum().backward()
    model = model.to(DEV)
    with torch.no_grad():
        _close(model(feats.to(DEV), lat_lons, global_context=ctx.to(DEV)), y_ref, what="wide regional (inference)")
    model.train()
    y = model(feats.to(DEV), lat_lons, global_context=ctx.to(DEV))
    _close(y, y_ref, what="wide regional (training forward)")
    (y * dy.to(DEV)).sum().backward()
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)


def test_wide_assimilator_forward_and_backward():
    """GraphWeatherAssimilator (analysis.py:52-150) at widths above 256, on the observations of the golden case."""
    from .test_oracle import _assimilator_setup

    out_lat_lons, llh, feats, g = _assimilator_setup()
    model = gw.GraphWeatherAssimilator(output_lat_lons=out_lat_lons, analysis_dim=24, node_dim=320, edge_dim=288, num_blocks=2,
                                       hidden_dim_processor_node=384, hidden_dim_processor_edge=300, hidden_dim_decoder=272)
    deterministic_fill_(model, seed=6)
    ref = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    rs = np.random.RandomState(3)
    dy = _t(rs, 1, 648, 24)
    y_ref = om.assimilator_forward(ref, om.graphs_to_dtype(g, torch.float64), feats.double(), 24)
    (y_ref * dy.double()).sum().backward()
    model = model.to(DEV)
    with torch.no_grad():
        _close(model(feats.to(DEV), llh.to(DEV)), y_ref, what="wide assimilator (inference)")
    model.train()
    y = model(feats.to(DEV), llh.to(DEV))
    (y * dy.to(DEV)).sum().backward()
    worst = {}
    _check_param_grads(model, ref, "", worst, bar=4e-3)
