"""Parity of the split-operand mode (``set_compute_dtype(model, "bf16x3")``, include/gw_amd.h: GW_DTYPE_BF16X3, csrc/gw_split.hip)
against the SAME references that gate the fp32 kernels: the golden vectors of the reference's own classes, the CPU oracle at
operator level and at every BASELINE size.  north_star's bar is 1e-3 of the tensor / delta scale; this mode is asserted at the
fp32 tests' own 2e-4 (a dropped cross term would show as ~2e-3, plain bf16 as ~1e-2)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd import _lib, ops  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features  # noqa: E402
from oracle import chunked as oc  # noqa: E402
from oracle import reference_math as om  # noqa: E402

from .helpers import pack_linear_bf16_ref  # noqa: E402

DEV = "cuda:0"
X3_REL = 2e-4        # asserted; north_star's bar is 1e-3
NORTH_STAR = 1e-3
X3 = gw.BF16X3


def _rel(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _close(a, ref, what, rel=X3_REL):
    r = _rel(a, ref)
    print(f"[x3] {what}: max-rel {r:.2e}")
    assert r <= rel, f"{what}: {r:.3e} > {rel:.1e}"
    return r


def _x3(m):
    return gw.set_compute_dtype(m, X3)


def _bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("n_out,k_total,k_lo,k_hi", [(256, 768, 256, 512), (256, 102, 0, 102), (78, 128, 0, 128), (256, 2, 0, 2)])
def test_pack_linear_x3_matches_layout_statement(n_out, k_total, k_lo, k_hi):
    """out[s][half][tile][lane][i]: half 0 = bf16(w), half 1 = bf16(w - bf16(w)) in the bf16 stream's element order."""
    rs = np.random.RandomState(0)
    w = rs.standard_normal((n_out, k_total)).astype(np.float32)
    L = _lib.lib()
    nbytes = L.gw_packed_bytes_bf16x3(n_out, k_lo, k_hi)
    assert nbytes == 2 * L.gw_packed_bytes_bf16(n_out, k_lo, k_hi)
    out = torch.zeros(nbytes // 2, dtype=torch.bfloat16, device=DEV)
    _lib.check(L.gw_pack_linear_bf16x3(torch.from_numpy(w).to(DEV).data_ptr(), n_out, k_total, k_lo, k_hi, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "pack x3")
    ref = pack_linear_bf16_ref(w, k_lo, k_hi)  # [steps, ntp, 64, 8] fp32 values
    hi = _bf16_round(ref)
    lo = _bf16_round(ref - hi)
    got = out.float().cpu().numpy().reshape(ref.shape[0], 2, ref.shape[1], 64, 8)
    assert np.array_equal(got[:, 0], hi)
    assert np.array_equal(got[:, 1], lo)
    # 16 significant bits: hi + lo reproduces w to 2^-16 relative
    nz = np.abs(ref) > 0
    assert (np.abs(got[:, 0] + got[:, 1] - ref)[nz] <= np.abs(ref)[nz] * 2.0 ** -16).all()


@pytest.mark.parametrize("tag,i,o,h,norm", [("node_enc", 102, 256, 256, "LayerNorm"), ("edge_enc", 2, 256, 256, "LayerNorm"),
                                            ("node_dec", 256, 78, 128, None)])
def test_mlp_matches_reference_golden(golden_dir, tag, i, o, h, norm):
    g = np.load(os.path.join(golden_dir, f"mlp_{tag}.npz"))
    m = gw.MLP(i, o, h, 2, norm)
    deterministic_fill_(m, seed=11)
    m = _x3(m.to(DEV))
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]).to(DEV))
    _close(y, torch.from_numpy(g["y"]), f"mlp {tag} golden")


@pytest.mark.parametrize("rows", [1, 63, 64, 65, 129, 1000])
def test_mlp_ragged_row_counts(rows):
    m = gw.MLP(256, 256, 256, 2, "LayerNorm")
    deterministic_fill_(m, seed=2)
    x = torch.from_numpy(np.random.RandomState(rows).standard_normal((rows, 256)).astype(np.float32))
    ref = om.mlp({"m." + k: v for k, v in m.state_dict().items()}, "m", x)
    m = _x3(m.to(DEV))
    with torch.no_grad():
        y = m(x.to(DEV))
    _close(y, ref, f"mlp rows={rows}")


@pytest.mark.parametrize("n_out,hidden,k", [(24, 128, 256), (1, 128, 256), (64, 256, 256), (80, 128, 256), (256, 256, 17), (256, 256, 128)])
def test_mlp_shapes(n_out, hidden, k):
    m = gw.MLP(k, n_out, hidden, 2, None if n_out <= 80 else "LayerNorm")
    deterministic_fill_(m, seed=n_out + k)
    x = torch.from_numpy(np.random.RandomState(n_out).standard_normal((333, k)).astype(np.float32))
    ref = om.mlp({"m." + kk: v for kk, v in m.state_dict().items()}, "m", x)
    m = _x3(m.to(DEV))
    with torch.no_grad():
        y = m(x.to(DEV))
    _close(y, ref, f"mlp {k}->{hidden}->{n_out}")


def test_mlp_more_hidden_layers_and_large_values():
    """4 hidden layers (the middle-layer loop) and activations spanning 1e-4 .. 1e4: the split keeps fp32's exponent range."""
    m = gw.MLP(64, 256, 256, 4, "LayerNorm")
    deterministic_fill_(m, seed=4)
    rs = np.random.RandomState(1)
    x = torch.from_numpy((rs.standard_normal((77, 64)) * 10.0 ** rs.uniform(-4, 4, size=(77, 1))).astype(np.float32))
    ref = om.mlp({"m." + k: v for k, v in m.state_dict().items()}, "m", x)
    m = _x3(m.to(DEV))
    with torch.no_grad():
        y = m(x.to(DEV))
    _close(y, ref, "mlp 4 hidden layers, wide dynamic range")


def test_graph_processor_random_coo_matches_reference_golden(golden_dir):
    """The compositional path on an arbitrary COO graph: projected node operands, raw per-edge features, residual, e' rows."""
    g = np.load(os.path.join(golden_dir, "graph_processor_random.npz"))
    gp = gw.GraphProcessor(mp_iterations=2, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256)
    deterministic_fill_(gp, seed=3)
    gp = _x3(gp.to(DEV))
    rs = np.random.RandomState(123)
    x = torch.from_numpy(rs.standard_normal((500, 256)).astype(np.float32)).to(DEV)
    ea = torch.from_numpy(rs.standard_normal((3000, 256)).astype(np.float32)).to(DEV)
    ei = torch.from_numpy(g["edge_index"]).to(DEV)
    with torch.no_grad():
        xo, eo = gp(x, ei, ea)
    _close(xo, torch.from_numpy(g["x_out"]), "random graph x (golden)")
    _close(eo[::5], torch.from_numpy(g["e_out_rows"]), "random graph e (golden)")


@pytest.mark.parametrize("deterministic", [False, True])
def test_graph_processor_edge_cases(deterministic):
    """empty edge list, isolated nodes, one hub destination longer than a tile (skewed segment); atomics and carry records."""
    gp = gw.GraphProcessor(mp_iterations=1, in_dim_node=256, in_dim_edge=256, hidden_dim_node=256, hidden_dim_edge=256)
    deterministic_fill_(gp, seed=8)
    p = {"gp." + k: v.clone() for k, v in gp.state_dict().items()}
    gp = _x3(gp.to(DEV))
    gw.set_deterministic(gp, deterministic)
    rs = np.random.RandomState(3)
    n = 70
    x = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    for e in (0, 5, 300, 1000):
        src = rs.randint(0, n, size=e)
        dst = np.where(rs.rand(e) < 0.7, 7, rs.randint(0, n, size=e)) if e else np.zeros(0, dtype=np.int64)
        ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
        ea = torch.from_numpy(rs.standard_normal((e, 256)).astype(np.float32))
        xr, er = om.graph_processor(p, "gp", x, ei, ea)
        with torch.no_grad():
            xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
            xo2, _ = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
        _close(xo, xr, f"edge case E={e} x (det={deterministic})")
        if e:
            _close(eo, er, f"edge case E={e} e")
        if deterministic:
            assert torch.equal(xo, xo2), "deterministic mode must be bitwise reproducible"


def test_mode_is_really_split_and_fp32_path_untouched():
    """x3 differs from fp32 by a few 1e-6 .. 1e-5 (so it is not the fp32 kernel), bf16 by ~1e-3 .. 1e-2 (so it is not plain bf16)."""
    m = gw.MLP(256, 256, 256, 2, "LayerNorm")
    deterministic_fill_(m, seed=5)
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((4096, 256)).astype(np.float32))
    ref = om.mlp({"m." + k: v.double() for k, v in m.state_dict().items()}, "m", x.double())
    m = m.to(DEV)
    with torch.no_grad():
        y32 = m(x.to(DEV)).cpu()
        y3 = _x3(m)(x.to(DEV)).cpu()
        y16 = gw.set_compute_dtype(m, torch.bfloat16)(x.to(DEV)).cpu()
        y32b = gw.set_compute_dtype(m, torch.float32)(x.to(DEV)).cpu()
    r32, r3, r16 = _rel(y32, ref), _rel(y3, ref), _rel(y16, ref)
    print(f"[x3] MLP vs fp64 oracle: fp32 {r32:.2e}, bf16x3 {r3:.2e}, bf16 {r16:.2e}")
    assert torch.equal(y32, y32b)
    assert r32 < 5e-6 and r3 < X3_REL and r16 > 20 * r3
    assert not torch.equal(y3, y32)


@pytest.mark.parametrize("tag,step,batch", [("10deg_b2", 10.0, 2), ("5deg_b1", 5.0, 1)])
def test_forecaster_matches_reference_golden(golden_dir, tag, step, batch):
    """The reference's own GraphWeatherForecaster outputs (tests/golden, oracle/gen_golden.py) in bf16x3 - the cases of
    tests/test_gpu_parity.py::test_forecaster_matches_reference_golden."""
    g = np.load(os.path.join(golden_dir, f"forecaster_{tag}.npz"))
    lat_lons = regular_lat_lons(step)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    model.set_compute_dtype(X3)
    feats = seeded_features(batch, len(lat_lons), 102, seed=42)
    fd = feats.to(DEV)
    y_ref = torch.from_numpy(g["out"])
    start = feats[..., :78]
    with torch.no_grad():
        x = model.encoder.encode(fd)
        _close(x[::37], torch.from_numpy(g["enc_x_rows"]), f"{tag} encoder mesh rows (golden)")
        y = model(fd).cpu()
        y_again = model(fd).cpu()
    r = _close(y - start, y_ref - start, f"forecaster {tag} delta (golden)")
    assert r <= NORTH_STAR
    assert _rel(y_again, y) <= 1e-5  # (atomics order)
    # the compositional API (Encoder -> Processor -> Decoder called separately, tests/test_model.py:106-119) gives the same forecast
    with torch.no_grad():
        x2, ei, ea = model.encoder(fd)
        xp = model.processor(x2, ei, ea)
        _close(xp[::37], torch.from_numpy(g["proc_x_rows"]), f"{tag} processor rows (golden)")
        y2 = model.decoder(xp, fd[..., :78]).cpu()
    _close(y2 - start, y_ref - start, f"forecaster {tag} compositional delta (golden)")


def test_forecaster_deterministic_mode_is_bitwise_and_equal_to_atomics():
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    model.set_compute_dtype(X3)
    fd = seeded_features(2, len(lat_lons), 102, seed=42).to(DEV)
    with torch.no_grad():
        y_atomic = model(fd)
        model.set_deterministic(True)
        y_a, y_b = model(fd), model(fd)
    assert torch.equal(y_a, y_b)
    assert _rel(y_a - fd[..., :78], y_atomic - fd[..., :78]) <= 1e-5


def _l2(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


GRAD_MAX_REL, GRAD_L2_REL = 2e-2, 6e-3  # fp32 path: 2e-2 / 3e-3 (tests/test_gpu_round2.py).  Split products carry 1e-5 instead of
#                                          1e-7 per product: the gradients downstream of the node encoder's first layer (ReLU gates
#                                          flipping inside long cancelling sums) reach 3.8e-3 l2 and vary by 1e-3 run to run with
#                                          the order of the atomics - the l2 bar of this mode is twice the fp32 path's


def test_training_step_gradients_fixed_bars_against_the_fp64_oracle():
    """Mixed-precision training (bf16x3 forward with fp32 activation saves, split input-gradient products and weight-gradient
    GEMMs, fp32 LayerNorm / ReLU backward): all 215 gradients of the 10 degree forecaster + NormalizedMSELoss against the
    oracle's fp64 autograd at fixed bars (the fp32 path's max-rel bar, twice its l2 bar), and against the fp32 kernels' own
    gradients."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    ref64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    g64 = om.graphs_to_dtype(model.encoder.graphs.as_oracle_dict(), torch.float64)
    feats = seeded_features(2, len(lat_lons), 102, seed=42)
    rs = np.random.RandomState(7)
    target = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78)).astype(np.float32))
    var = torch.from_numpy((rs.rand(78) + 0.5).astype(np.float32))
    om.normalized_mse_loss(om.forecaster_forward(ref64, g64, feats.double()), target.double(), lat_lons, var.double(), True).backward()
    model = model.to(DEV).train()
    crit = gw.NormalizedMSELoss(var.tolist(), lat_lons, normalize=True)
    loss32 = crit(model(feats.to(DEV)), target.to(DEV))
    loss32.backward()
    g32 = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    model.set_compute_dtype(X3)
    loss3 = crit(model(feats.to(DEV)), target.to(DEV))
    loss3.backward()
    assert abs(loss3.item() - loss32.item()) <= 1e-4 * abs(loss32.item())
    # encoder.h3_nodes is the one ill-conditioned gradient of this model: a long cancelling sum over every use of the mesh
    # embedding, in which ReLU gates that flip between two summation orders move single entries - the oracle's OWN fp32 autograd
    # differs from its fp64 autograd by 7e-3 (max-rel) there, i.e. a 1e-7 product error is amplified 1e5 x.  Split products carry
    # 1e-5: that tensor gets its own stated bar; the other 214 tensors keep the fp32 path's bars.
    H3, H3_MAX_REL, H3_L2_REL = "encoder.h3_nodes", 1.5e-1, 1.5e-2  # (measured 6.9e-2 max-rel, 5.1e-3 .. 7.2e-3 l2 over runs)
    worst, worst_h3, worst32, bad = (0, 0, ""), (0, 0), (0, 0, ""), []
    for k, p in model.named_parameters():
        m, l = _rel(p.grad, ref64[k].grad), _l2(p.grad, ref64[k].grad)
        if k == H3:
            worst_h3 = (m, l)
            if m > H3_MAX_REL or l > H3_L2_REL:
                bad.append((k, m, l))
        else:
            if m > worst[0]:
                worst = (m, l, k)
            if m > GRAD_MAX_REL or l > GRAD_L2_REL:
                bad.append((k, m, l))
        m2, l2 = _rel(p.grad, g32[k]), _l2(p.grad, g32[k])
        if l2 > worst32[1] and k != H3:
            worst32 = (m2, l2, k)
    print(f"[x3 backward] worst vs fp64 autograd max-rel {worst[0]:.2e} (l2 {worst[1]:.2e}, {worst[2]}); {H3}: max-rel "
          f"{worst_h3[0]:.2e} l2 {worst_h3[1]:.2e}; vs the fp32 kernels' gradients l2 {worst32[1]:.2e} (max-rel {worst32[0]:.2e}, {worst32[2]})")
    assert not bad, bad[:6]
    assert worst32[1] <= GRAD_L2_REL


def test_training_loss_decreases_and_inference_kernels_see_the_new_weights():
    """A few AdamW steps in bf16x3 on the 30 degree forecaster: the loss falls, and the no_grad forward after the steps (fused
    inference path with its per-weight-version caches) equals the training-mode forward of the same weights."""
    lat_lons = regular_lat_lons(30.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=3)
    model = model.to(DEV).train()
    model.set_compute_dtype(X3)
    feats = seeded_features(2, len(lat_lons), 102, seed=1).to(DEV)
    target = (feats[..., :78] * 0.9).contiguous()
    crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons)
    opt = gw.AdamW(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(model(feats), target)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
    y_train = model(feats).detach()
    model.eval()
    with torch.no_grad():
        y_eval = model(feats)
    assert _rel(y_eval - feats[..., :78], y_train - feats[..., :78]) <= 1e-4


def test_bfloat16_still_refuses_autograd():
    m = gw.set_compute_dtype(gw.MLP(256, 256, 256, 2, "LayerNorm").to(DEV), torch.bfloat16)
    with pytest.raises(NotImplementedError):
        m(torch.randn(8, 256, device=DEV))


def _row_sample(lat_lons, n, seed):
    G = len(lat_lons)
    rs = np.random.RandomState(seed)
    lat = np.asarray([ll[0] for ll in lat_lons])
    polar = np.concatenate([np.argsort(lat)[:20], np.argsort(-lat)[:20]])
    rows = np.unique(np.concatenate([rs.choice(G, size=n - polar.size, replace=False), polar]))
    return torch.from_numpy(rows).long()


def test_c2_and_c3_one_degree_against_the_oracle():
    """BASELINE.json configs[1] and configs[2] sizes (1 degree, 64 800 nodes; batch 2 and batch 16) in bf16x3 against the fp32
    oracle on a fixed row sample (incl. the polar rows), samples 0, 1 (batch 2) and 0, 15 (batch 16)."""
    lat_lons = regular_lat_lons(1.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = model.encoder.graphs.as_oracle_dict()
    feats = seeded_features(16, len(lat_lons), 102, seed=42)
    rows = _row_sample(lat_lons, 1500, seed=1)
    ref = oc.forecaster_rows(sd, g, feats[[0, 1, 15]], rows)  # [3, R, 78] fp32 oracle
    model = model.to(DEV).eval()
    model.set_compute_dtype(X3)
    fd = feats.to(DEV)
    with torch.no_grad():
        y2 = model(fd[:2].contiguous())[:, rows.to(DEV)].cpu()
        y16 = model(fd)[[0, 15]][:, rows.to(DEV)].cpu()
    start = feats[[0, 1, 15]][:, rows, :78]
    d_ref = ref - start
    r2 = _rel(y2 - start[:2], d_ref[:2])
    r16 = _rel(y16 - start[[0, 2]], d_ref[[0, 2]])
    print(f"[parity] bf16x3 1deg: B=2 max-rel {r2:.2e}; B=16 max-rel {r16:.2e} of the delta scale ({rows.numel()} rows)")
    assert r2 <= X3_REL and r16 <= X3_REL


def test_c5_quarter_degree_against_the_chunked_oracle():
    """BASELINE.json configs[4]: 0.25 degree, mesh resolution 3, batch 1, in bf16x3 against the slab-wise oracle."""
    lat_lons = regular_lat_lons(0.25)
    model = gw.GraphWeatherForecaster(lat_lons, resolution=3)
    deterministic_fill_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = model.encoder.graphs.as_oracle_dict()
    G = len(lat_lons)
    feats = torch.from_numpy(np.random.RandomState(5).standard_normal((1, G, 102)).astype(np.float32))
    rows = _row_sample(lat_lons, 600, seed=2)
    ref = oc.forecaster_rows(sd, g, feats, rows, slab=1 << 17)
    model = model.to(DEV).eval()
    model.set_compute_dtype(X3)
    with torch.no_grad():
        y = model(feats.to(DEV))[:, rows.to(DEV)].cpu()
    start = feats[:, rows, :78]
    r = _rel(y - start, ref - start)
    print(f"[parity] bf16x3 C5 0.25deg res 3 B=1: max-rel {r:.2e} on {rows.numel()} rows")
    assert r <= X3_REL


def test_graphcast_wrapper_golden():
    """graphcast/model.py: decoder head 256 wide -> node update and head as two launches in this mode."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphCast(lat_lons, efficient_batching=True)
    deterministic_fill_(model, seed=5)
    feats = seeded_features(2, len(lat_lons), 78, seed=9)
    model = model.to(DEV).eval()
    gw.set_compute_dtype(model, X3)
    with torch.no_grad():
        y = model(feats.to(DEV)).cpu()
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphcast_10deg_b2.npz"))
    _close(y - feats, torch.from_numpy(gold["y"]) - feats, "GraphCast delta (reference golden)")


def test_assimilator_golden(golden_dir):
    """GraphWeatherAssimilator (analysis.py:52-150): observation graph built per call, 24-feature head."""
    from .test_oracle import _assimilator_setup

    gold = np.load(os.path.join(golden_dir, "assimilator_10deg.npz"))
    out_lat_lons, llh, feats, g = _assimilator_setup()
    model = gw.GraphWeatherAssimilator(output_lat_lons=out_lat_lons, analysis_dim=24)
    deterministic_fill_(model, seed=6)
    model = model.to(DEV).eval()
    gw.set_compute_dtype(model, X3)
    with torch.no_grad():
        y = model(feats.to(DEV), llh.to(DEV))
    _close(y, torch.from_numpy(gold["y"]), "assimilator (reference golden)")


# ---- widths other than 256: zero-padded on the same split kernels (masked LayerNorm statistics, gw_mlp_weights.ln_width) -----------
@pytest.mark.parametrize("i,o,h,layers,norm", [(16, 128, 128, 2, "LayerNorm"), (102, 32, 32, 2, "LayerNorm"), (200, 64, 96, 3, "LayerNorm"),
                                                (32, 12, 32, 2, "LayerNorm"), (157, 1, 64, 1, None), (64, 100, 40, 1, "LayerNorm"),
                                                (256, 78, 32, 1, None)])
def test_mlp_any_width(i, o, h, layers, norm):
    """The cases of tests/test_gpu_narrow.py::test_mlp_any_width_forward_and_backward (the reference's MLP defaults are 128 wide,
    graph_net_block.py:20-28), inference and training forward + backward in bf16x3."""
    m = gw.MLP(i, o, h, layers, norm)
    deterministic_fill_(m, seed=i + o)
    rs = np.random.RandomState(h)
    x = torch.from_numpy(rs.standard_normal((333, i)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((333, o)).astype(np.float32))
    ref = {"m." + k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    y_ref = om.mlp(ref, "m", xr)
    y_ref.backward(dy.double())
    m = _x3(m.to(DEV))
    with torch.no_grad():
        y = m(x.to(DEV))
    assert y.shape == (333, o)
    _close(y, y_ref, f"MLP {i}->{h}x{layers}->{o} (inference)")
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    _close(y, y_ref, f"MLP {i}->{h}x{layers}->{o} (training forward)")
    y.backward(dy.to(DEV))
    if not (o == 1 and norm):
        for k, p in m.named_parameters():
            assert _rel(p.grad, ref["m." + k].grad) < 2e-3, k
        assert _rel(xd.grad, xr.grad) < 2e-3


def test_graph_processor_reference_defaults_and_narrow_forecaster():
    """``GraphProcessor()`` at the reference's default widths (128, graph_net_block.py:234-244) on a random COO graph, and a
    forecaster with node 64 / edge 48 / hidden 96 and 40 / decoder hidden 32 (single hidden layer in the edge MLPs), in bf16x3."""
    gp = gw.GraphProcessor(mp_iterations=3)
    deterministic_fill_(gp, seed=4)
    p = {"gp." + k: v.clone() for k, v in gp.state_dict().items()}
    rs = np.random.RandomState(5)
    n, e = 150, 900
    x = torch.from_numpy(rs.standard_normal((n, 128)).astype(np.float32))
    ea = torch.from_numpy(rs.standard_normal((e, 128)).astype(np.float32))
    ei = torch.from_numpy(np.stack([rs.randint(0, n, size=e), np.where(rs.rand(e) < 0.2, 3, rs.randint(0, n, size=e))]).astype(np.int64))
    xo_r, eo_r = om.graph_processor(p, "gp", x, ei, ea)
    gp = _x3(gp.to(DEV))
    with torch.no_grad():
        xo, eo = gp(x.to(DEV), ei.to(DEV), ea.to(DEV))
    assert xo.shape == (n, 128) and eo.shape == (e, 128)
    _close(xo, xo_r, "GraphProcessor(128) nodes")
    _close(eo, eo_r, "GraphProcessor(128) edges")
    lat_lons = regular_lat_lons(15.0)
    kw = dict(feature_dim=20, aux_dim=5, node_dim=64, edge_dim=48, num_blocks=2, hidden_dim_processor_node=96,
              hidden_dim_processor_edge=40, hidden_layers_processor_node=2, hidden_layers_processor_edge=1, hidden_dim_decoder=32,
              hidden_layers_decoder=2)
    model = gw.GraphWeatherForecaster(lat_lons, **kw)
    deterministic_fill_(model, seed=9)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    feats = torch.from_numpy(np.random.RandomState(1).standard_normal((2, len(lat_lons), 25)).astype(np.float32))
    y_ref = om.forecaster_forward(sd, model.encoder.graphs.as_oracle_dict(), feats, feature_dim=20)
    model = model.to(DEV).eval()
    model.set_compute_dtype(X3)
    with torch.no_grad():
        y = model(feats.to(DEV)).cpu()
    res = feats[..., :20]
    _close(y - res, y_ref - res, "narrow forecaster delta")


def test_regional_forecaster_small_config():
    """The reference's tests/test_regional_forecast.py small configuration (32 wide, LayerNorm on the 12-feature head) in bf16x3
    against the fp32 kernels on the same weights."""
    cfg = gw.RegionalForecasterConfig(feature_dim=12, aux_dim=4, node_dim=32, edge_dim=32, num_blocks=2, hidden_dim_processor_node=32,
                                      hidden_dim_processor_edge=32, hidden_dim_decoder=32)
    model = cfg.build()
    deterministic_fill_(model, seed=2)
    model = model.to(DEV).eval()
    lat_lons = [(51.5, -0.1), (52.0, 0.5), (53.0, -1.0), (54.0, -2.0), (50.0, -3.0)]
    feats = torch.from_numpy(np.random.RandomState(4).standard_normal((2, 5, 16)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        ref = model(feats, lat_lons)
        gw.set_compute_dtype(model, X3)
        out = model(feats, lat_lons)
    assert out.shape == (2, 5, 12)
    _close(out - feats[..., :12], ref - feats[..., :12], "regional small config vs the fp32 kernels")


@pytest.mark.parametrize("part", ["encoder", "processor", "decoder"])
def test_one_stage_in_bf16x3_and_the_rest_in_fp32(part):
    """``set_compute_dtype`` on a sub-module: the stages hand fp32 rows to each other in every mode, so any mix of fp32 and
    bf16x3 stages is a valid model (the fused forward falls back to stage-by-stage hand-over where the dtypes differ)."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=0)
    model = model.to(DEV).eval()
    fd = seeded_features(2, len(lat_lons), 102, seed=42).to(DEV)
    with torch.no_grad():
        ref = model(fd)
        gw.set_compute_dtype(getattr(model, part), X3)
        out = model(fd)
        gw.set_compute_dtype(model, torch.float32)
        back = model(fd)
    _close(out - fd[..., :78], ref - fd[..., :78], f"{part} in bf16x3, the rest fp32")
    assert not torch.equal(out, ref)
    assert _rel(back, ref) <= 1e-5
