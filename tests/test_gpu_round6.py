"""Round-6 GPU tests: the row-split node update of mesh-sized launches (csrc/gw_noders.hip) - NodeProcessor.forward behind
scatter_sum, graph_net_block.py:184-193 - against a float64 statement of the same MLP and against the 64-column kernels on
the same rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import graph_weather_amd as gw  # noqa: E402
from graph_weather_amd import ops  # noqa: E402
from graph_weather_amd.ops import Operand  # noqa: E402
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features  # noqa: E402

DEV = "cuda:0"
X3 = ops.BF16X3


def _rel(a, ref):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _node_mlp_fp64(blk, x, agg, x_res):
    """graph_net_block.py:189-191 in float64: LayerNorm(MLP(cat[x, agg])) + x."""
    m = blk.node_model.node_mlp
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    h = torch.cat([x, agg], dim=1).cpu().double()
    h = torch.relu(h @ sd["model.0.weight"].T + sd["model.0.bias"])
    h = torch.relu(h @ sd["model.2.weight"].T + sd["model.2.bias"])
    h = h @ sd["model.4.weight"].T + sd["model.4.bias"]
    h = torch.nn.functional.layer_norm(h, (256,), sd["model.5.weight"], sd["model.5.bias"], eps=1e-5)
    return h + x_res.cpu().double()


def _blocks(dtype):
    torch.manual_seed(0)
    blk = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
    nxt = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
    deterministic_fill_(blk, seed=13)
    deterministic_fill_(nxt, seed=14)
    gw.set_compute_dtype(blk, dtype)
    gw.set_compute_dtype(nxt, dtype)
    return blk.to(DEV), nxt.to(DEV)


# rows per batch element x batch: 1, 2 and 3 column groups per workgroup (<= 4 096, <= 8 192, <= 12 288 columns), ragged
# against 16 / 32 / 48, one column, the mesh itself at batch 1 and 2
SIZES = [(1, 1), (15, 1), (777, 3), (4096, 1), (4097, 1), (5882, 1), (2731, 3), (5882, 2), (12288, 1), (6143, 2)]


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 2e-6), (X3, 3e-5)])
@pytest.mark.parametrize("n,B", SIZES)
def test_row_split_node_update_against_float64(dtype, bar, n, B):
    blk, nxt = _blocks(dtype)
    g = torch.Generator(device="cpu").manual_seed(n * 7 + B)
    x = torch.randn(B * n, 256, generator=g).to(DEV)
    agg = (2.0 * torch.randn(B * n, 256, generator=g)).to(DEV)
    pm_n, pm_e = blk.node_model.node_mlp.packed(), nxt.edge_model.edge_mlp.packed()
    ref = _node_mlp_fp64(blk, x, agg, x)
    out = ops.node_update_forward(pm_n, B * n, n, Operand(x, n, 256), Operand(x, n, 256), Operand(agg, n, 256))
    zero = torch.full((B * n, 256), 7.0, device=DEV)
    out2, (ps, pd) = ops.node_update_forward(pm_n, B * n, n, Operand(x, n, 256), Operand(x, n, 256), Operand(agg, n, 256),
                                             post_w=[pm_e.w1[0], pm_e.w1[1]], zero_rows=zero)
    torch.cuda.synchronize()
    r = _rel(out, ref)
    print(f"[row-split node update {dtype} n={n} B={B}] max-rel vs float64 {r:.2e}")
    assert r <= bar
    assert torch.equal(out, out2)
    assert (zero == 0).all()
    w1 = nxt.edge_model.edge_mlp.state_dict()["model.0.weight"].detach().cpu().double()
    ps_ref = out.cpu().double() @ w1[:, 0:256].T
    pd_ref = out.cpu().double() @ w1[:, 256:512].T
    assert _rel(ps, ps_ref) <= bar and _rel(pd, pd_ref) <= bar


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 0.0), (X3, 0.0)])
def test_row_split_node_update_equals_the_64_column_kernel_on_the_same_rows(dtype, bar):
    """The same rows as the head of a launch too large for the row-split form (> 12 288 columns: chain_kernel / chainx3_kernel,
    64 columns per workgroup): products and LayerNorm sums are added in the same order - BITWISE the same rows, so a
    checkpointed segment (inference kernels in the forward, replay with activation saves on the 64-column kernels in the
    backward: autograd.RecomputeFunction) sees the values of its first run."""
    blk, nxt = _blocks(dtype)
    n_small, n_big = 5882, 12288 + 640
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n_big, 256, generator=g).to(DEV)
    agg = torch.randn(n_big, 256, generator=g).to(DEV)
    pm_n, pm_e = blk.node_model.node_mlp.packed(), nxt.edge_model.edge_mlp.packed()
    big, (ps_b, pd_b) = ops.node_update_forward(pm_n, n_big, n_big, Operand(x, n_big, 256), Operand(x, n_big, 256),
                                                Operand(agg, n_big, 256), post_w=[pm_e.w1[0], pm_e.w1[1]])
    xs, as_ = x[:n_small].contiguous(), agg[:n_small].contiguous()
    small, (ps_s, pd_s) = ops.node_update_forward(pm_n, n_small, n_small, Operand(xs, n_small, 256), Operand(xs, n_small, 256),
                                                  Operand(as_, n_small, 256), post_w=[pm_e.w1[0], pm_e.w1[1]])
    # ... and with the node operand as a cached product (the encoder's node update): the same order of additions there too
    px = torch.randn(n_big, 256, generator=g).to(DEV)
    big_p = ops.node_update_forward(pm_n, n_big, n_big, Operand(px, n_big, 256, projected=True), Operand(x, n_big, 256),
                                    Operand(agg, n_big, 256))
    small_p = ops.node_update_forward(pm_n, n_small, n_small, Operand(px[:n_small].contiguous(), n_small, 256, projected=True),
                                      Operand(xs, n_small, 256), Operand(as_, n_small, 256))
    torch.cuda.synchronize()
    r = max(_rel(small, big[:n_small]), _rel(ps_s, ps_b[:n_small]), _rel(pd_s, pd_b[:n_small]), _rel(small_p, big_p[:n_small]))
    print(f"[row-split vs 64-column kernel {dtype}] max-rel {r:.2e}")
    assert r <= bar


@pytest.mark.parametrize("dtype", [torch.float32, X3])
def test_row_split_node_update_without_layernorm_and_with_indexed_rows(dtype):
    """norm_type=None (graph_net_block.py:50-59) and operands addressed through an index (rows of a larger table)."""
    torch.manual_seed(1)
    blk = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, None)
    deterministic_fill_(blk, seed=21)
    gw.set_compute_dtype(blk, dtype)
    blk = blk.to(DEV)
    n = 1000
    table = torch.randn(3 * n, 256, device=DEV)
    idx = torch.randperm(3 * n, device=DEV)[:n].to(torch.int32)
    agg = torch.randn(n, 256, device=DEV)
    pm = blk.node_model.node_mlp.packed()
    xi = Operand(table, 0, 256, index=idx)
    out = ops.node_update_forward(pm, n, n, xi, xi, Operand(agg, n, 256))
    x = table[idx.long()]
    sd = {k: v.detach().cpu().double() for k, v in blk.node_model.node_mlp.state_dict().items()}
    h = torch.cat([x, agg], dim=1).cpu().double()
    h = torch.relu(h @ sd["model.0.weight"].T + sd["model.0.bias"])
    h = torch.relu(h @ sd["model.2.weight"].T + sd["model.2.bias"])
    ref = h @ sd["model.4.weight"].T + sd["model.4.bias"] + x.cpu().double()
    assert _rel(out, ref) <= (2e-6 if dtype == torch.float32 else 3e-5)


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 2e-6), (X3, 3e-5)])
@pytest.mark.parametrize("n,B", [(5882, 2), (900, 1)])
def test_row_split_node_update_with_a_projected_or_absent_node_operand(dtype, bar, n, B):
    """The encoder's mesh rows enter their node update as a cached, batch-shared product x . Wx^T (``projected``: a gather-add,
    no matrix pass, encoder.py:235-241 through the layer-1 split) and the decoder's grid rows are zeros
    (assimilator_decoder.py:84: operand absent)."""
    blk, _ = _blocks(dtype)
    g = torch.Generator(device="cpu").manual_seed(n + B)
    x = torch.randn(n, 256, generator=g).to(DEV)  # shared by the batch
    agg = torch.randn(B * n, 256, generator=g).to(DEV)
    sd = {k: v.detach().cpu().double() for k, v in blk.node_model.node_mlp.state_dict().items()}
    px = (x.cpu().double() @ sd["model.0.weight"][:, :256].T).float().to(DEV)  # the cached product rows

    def ref(xin, with_res):
        h = torch.relu(xin @ sd["model.0.weight"][:, :256].T + agg.cpu().double() @ sd["model.0.weight"][:, 256:].T + sd["model.0.bias"])
        h = torch.relu(h @ sd["model.2.weight"].T + sd["model.2.bias"])
        h = h @ sd["model.4.weight"].T + sd["model.4.bias"]
        h = torch.nn.functional.layer_norm(h, (256,), sd["model.5.weight"], sd["model.5.bias"], eps=1e-5)
        return h + (xin if with_res else 0.0)

    pm = blk.node_model.node_mlp.packed()
    xb = x.cpu().double().repeat(B, 1)
    out_p = ops.node_update_forward(pm, B * n, n, Operand(px, 0, 256, projected=True), Operand(x, 0, 256), Operand(agg, n, 256))
    out_z = ops.node_update_forward(pm, B * n, n, ops.ZERO, ops.ZERO, Operand(agg, n, 256))
    torch.cuda.synchronize()
    rp, rz = _rel(out_p, ref(xb, True)), _rel(out_z, ref(torch.zeros_like(xb), False))
    print(f"[row-split node update {dtype} n={n} B={B}] projected x {rp:.2e}, absent x {rz:.2e}")
    assert rp <= bar and rz <= bar


def _forecaster(deg=10.0, seed=0):
    lat_lons = regular_lat_lons(deg)
    model = gw.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(model, seed=seed)
    return model.to(DEV).eval(), lat_lons


@pytest.mark.parametrize("dtype", [torch.float32, X3])
def test_eval_forward_replays_itself_from_a_hip_graph_and_follows_weight_updates(dtype):
    """graphed.AutoGraph behind ``GraphWeatherForecaster.forward``: in eval() under no_grad() the third call of a shape is the
    first replay.  A replay returns its own tensor (nothing a later call overwrites), equals the eager forward (bitwise in
    deterministic mode), and a weight update between two calls is followed - eager again for two calls, then a new capture;
    grad mode, training mode and ``auto_graph = False`` stay eager."""
    model, lat_lons = _forecaster()
    model.set_compute_dtype(dtype)
    model.set_deterministic(True)
    a = seeded_features(2, len(lat_lons), 102, seed=1).to(DEV)
    b = seeded_features(2, len(lat_lons), 102, seed=2).to(DEV)
    with torch.no_grad():
        ref_a, ref_b = model._forward_eager(a), model._forward_eager(b)
        ys = [model(a), model(b), model(a), model(b), model(a)]
        auto = model.__dict__["_auto"]
        assert auto._fg is not None and auto._fg.captures == 1  # calls 3-5 replayed one capture
        for y, ref in zip(ys, (ref_a, ref_b, ref_a, ref_b, ref_a)):
            assert torch.equal(y, ref)
        assert ys[2].data_ptr() != ys[4].data_ptr()  # clones, not the graph's buffer
        for p in model.parameters():
            p.mul_(1.01)
        new_a = model._forward_eager(a)
        assert not torch.equal(new_a, ref_a)
        outs = [model(a) for _ in range(4)]  # 2 eager calls, capture, replay
        assert all(torch.equal(o, new_a) for o in outs)
        assert auto._fg.captures == 2 and auto._fg.pinned  # the same buffer on every call since the update: captured on it
        c = seeded_features(1, len(lat_lons), 102, seed=3).to(DEV)  # another shape: counted afresh
        assert torch.equal(model(c), model._forward_eager(c))
        model.auto_graph = False
        n = auto._fg.captures
        assert torch.equal(model(a), new_a) and auto._fg.captures == n
        model.auto_graph = True
    # grad mode: the autograd path, untouched by the graph
    model.set_deterministic(False)
    x = a.clone().requires_grad_(True)
    model(x).sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_forward_graph_reads_a_stable_caller_buffer_in_place():
    """graphed.ForwardGraph: a caller that hands over one and the same buffer (a rollout's input, a staging buffer) gets a graph
    captured ON that buffer after ``pin_after`` calls - no copy of the batch per step; another buffer sends the graph back to
    its own input for good (it never writes into a caller's tensor)."""
    model, lat_lons = _forecaster()
    model.set_deterministic(True)
    fg = gw.ForwardGraph(model, pin_after=3)
    buf = seeded_features(2, len(lat_lons), 102, seed=5).to(DEV)
    other = seeded_features(2, len(lat_lons), 102, seed=6).to(DEV)
    with torch.no_grad():
        fg(buf)
        fg(buf)
        assert not fg.pinned and fg.captures == 1
        y = fg(buf).clone()
        assert fg.pinned and fg.captures == 2 and fg.input is buf
        assert torch.equal(y, model._forward_eager(buf))
        buf.copy_(other)  # the next step's data written into the same buffer: read in place
        keep = other.clone()
        y2 = fg(buf).clone()
        assert fg.captures == 2 and torch.equal(y2, model._forward_eager(other)) and not torch.equal(y2, y)
        y3 = fg(other).clone()  # a different buffer: back to an own input buffer, the caller's tensors untouched
        assert not fg.pinned and fg.captures == 3 and torch.equal(y3, y2)
        assert torch.equal(buf, keep) and torch.equal(other, keep)
        for _ in range(4):
            fg(buf)
        assert not fg.pinned and fg.captures == 3  # no second attempt


def test_rollout_through_the_automatic_graph_equals_the_eager_rollout():
    """rollout.py (train/run.py:506-528 loop shape): step t's output becomes step t + 1's input in ONE reused buffer - the
    automatic graph pins that buffer; every step equals the eager rollout."""
    from graph_weather_amd.rollout import rollout

    model, lat_lons = _forecaster()
    model.set_deterministic(True)
    feats = seeded_features(1, len(lat_lons), 102, seed=9).to(DEV)
    model.auto_graph = False
    ref = rollout(model, feats, 6)
    model.auto_graph = True
    out = rollout(model, feats, 6)
    assert len(out) == 6 and all(torch.equal(o, r) for o, r in zip(out, ref))
    assert model.__dict__["_auto"]._fg.pinned


def test_graphcast_wrapper_replays_itself_too():
    """graphcast/model.py:264-286 behind the same policy: the reference's own fast entry point (its benchmark script times this
    wrapper) replays one HIP graph in eval() under no_grad(); gradient mode keeps the checkpointed autograd path."""
    lat_lons = regular_lat_lons(10.0)
    model = gw.GraphCast(lat_lons, efficient_batching=True)
    deterministic_fill_(model, seed=2)
    model = model.to(DEV).eval()
    gw.set_deterministic(model, True)
    x = seeded_features(2, len(lat_lons), 78, seed=4).to(DEV)
    with torch.no_grad():
        ref = model._forward_eager(x)
        ys = [model(x) for _ in range(5)]
    auto = model.__dict__["_auto"]
    assert auto._fg is not None and auto._fg.captures == 1 and auto._fg.pinned
    assert all(torch.equal(y, ref) for y in ys)
    gw.set_deterministic(model, False)
    model.train()
    xg = x.clone().requires_grad_(True)
    model(xg).sum().backward()
    assert torch.isfinite(xg.grad).all()


def test_automatic_graph_under_inference_mode_leaves_the_generator_and_later_captures_intact():
    """torch.cuda.graph updates the default generator's graph-safe state tensors in place at the start of every capture: were they
    created under a caller's ``torch.inference_mode()`` (the automatic graph capturing there), the next capture in normal mode
    would fail half-way and every later ``torch.rand`` on the device would raise.  The capture therefore runs with inference
    mode locally off: a replayed forward under inference_mode, device RNG afterwards, a second capture in normal mode."""
    model, lat_lons = _forecaster()
    x = seeded_features(1, len(lat_lons), 102, seed=8).to(DEV)
    with torch.inference_mode():
        ys = [model(x) for _ in range(4)]
    auto = model.__dict__["_auto"]
    assert auto.enabled and auto._fg is not None and auto._fg.captures == 1
    with torch.no_grad():
        ref = model._forward_eager(x)
    assert all((y - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() for y in ys)
    r = torch.rand(16, device=DEV) + torch.randn(16, device=DEV)  # the device generator still works
    assert torch.isfinite(r).all()
    other, _ = _forecaster(seed=1)
    fg = gw.ForwardGraph(other)
    with torch.no_grad():
        y = fg(x).clone()
        assert fg.captures == 1 and (y - other._forward_eager(x)).abs().max().item() <= 1e-5 * y.abs().max().item()
    assert torch.isfinite(torch.rand(4, device=DEV)).all()


@pytest.mark.parametrize("rows,B,n_out,with_res,x_mode", [(64800, 2, 78, True, "zero"), (1000, 3, 78, True, "proj"), (77, 1, 37, False, "zero"),
                                                           (4097, 1, 80, True, "raw")])
def test_fp32_node_update_with_head_is_bitwise_the_two_launches(rows, B, n_out, with_res, x_mode):
    """gw_node_update_head_forward with fp32 weights (v17): NodeProcessor.forward of the decoder block (graph_net_block.py:189-191)
    + node_decoder (assimilator_decoder.py:197) + the Decoder residual (decoder.py:93) in ONE launch - the [rows, 256] table
    between them is never written.  Same arithmetic in the same order as gw_node_update_forward followed by gw_mlp_forward."""
    torch.manual_seed(rows + n_out)
    blk = gw.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
    head = gw.MLP(256, n_out, 128, 2, None)
    deterministic_fill_(blk, seed=31)
    deterministic_fill_(head, seed=32)
    blk, head = blk.to(DEV), head.to(DEV)
    n = rows * B
    agg = torch.randn(n, 256, device=DEV)
    feats = torch.randn(n, 102, device=DEV)
    pm, ph = blk.node_model.node_mlp.packed(), head.packed()
    if x_mode == "zero":
        x = ops.ZERO
    elif x_mode == "proj":
        x = Operand(torch.randn(rows, 256, device=DEV), 0, 256, projected=True)  # a cached batch-shared product table
    else:
        x = Operand(torch.randn(n, 256, device=DEV), rows, 256)
    res = Operand(feats, rows, n_out) if with_res else None
    with torch.no_grad():
        fused = ops.node_update_head_forward(pm, ph, n, rows, x, Operand(agg, rows, 256), res)
        xg = ops.node_update_forward(pm, n, rows, x, ops.ZERO, Operand(agg, rows, 256))
        two = head.run(xg, n, rows, residual=res)
    torch.cuda.synchronize()
    assert fused.shape == (n, n_out) and torch.equal(fused, two[:, :n_out])


@pytest.mark.parametrize("dtype", [torch.float32, X3])
@pytest.mark.parametrize("B", [1, 2])
def test_cold_forward_rebuilds_the_decoder_tables_on_a_side_stream_with_the_same_result(dtype, B):
    """AssimilatorDecoder.prefetch_tables: right after a weight update the decoder's edge embedding and its layer-1 product are
    rebuilt on a side stream beside encoder and processor (assimilator_decoder.py:175-177 recomputes them every forward); the
    forecast is that of the warm forward with the same weights, call after call, and a warm forward starts no side work."""
    model, lat_lons = _forecaster(deg=5.0, seed=3)
    model.set_compute_dtype(dtype)
    model.set_deterministic(True)
    model.auto_graph = False
    x = seeded_features(B, len(lat_lons), 102, seed=12).to(DEV)
    with torch.no_grad():
        model(x)
        assert model.decoder.prefetch_tables(x.device) is None  # warm: nothing to do
        for trial in range(3):
            for p in model.parameters():
                p.mul_(1.0 + 0.01 * (trial + 1))
            assert not model.decoder._cache.fresh("dec_e", None)
            cold = model(x)  # tables rebuilt on the side stream inside this call
            warm = model(x)
            assert torch.equal(cold, warm), trial
            assert model.decoder.prefetch_tables(x.device) is None
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows", [1, 64, 1000, 4097])
def test_chain_backward_on_split_operands_against_torch_and_the_single_products(rows):
    """gw_mlp_chain_backward_bf16x3 (ABI v18): d1 = (d W2) * (h1 > 0), dz0 = (d1 W1) * (h0 > 0) and the fan products dz0 W0[:, block]
    in one launch on split operands - against torch fp64 at the split mode's accuracy (operands carry 16 significant bits) and
    against the single-layer launches it replaces (the same three-MFMA products on the same split of the same rows: equal to
    fp32 rounding of the intermediate rows, which the single launches read back from memory as the identical fp32 values)."""
    from graph_weather_amd import _lib, autograd as ag

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(rows)
    W2, W1 = (torch.randn(256, 256, generator=g) / 16).to(DEV), (torch.randn(256, 256, generator=g) / 16).to(DEV)
    W0 = (torch.randn(256, 768, generator=g) / 16).to(DEV)
    d = torch.randn(rows, 256, generator=g).to(DEV)
    h1, h0 = torch.randn(rows, 256, generator=g).relu().to(DEV), torch.randn(rows, 256, generator=g).relu().to(DEV)
    n = int(L.gw_packed_bytes_bf16x3(256, 0, 256)) // 2
    blocks = [(W2, 0), (W1, 0), (W0, 0), (W0, 256), (W0, 512)]
    buf = torch.empty(len(blocks) * n, dtype=torch.int16, device=DEV)
    ops.pack_many(_lib.DTYPE_BF16X3, [(W.data_ptr() + 4 * lo, 1, int(W.shape[1]), 256, 256, buf[i * n:].data_ptr())
                                      for i, (W, lo) in enumerate(blocks)], [], st)
    pk = [buf[i * n:(i + 1) * n] for i in range(len(blocks))]
    dd = d.double()
    r1 = (dd @ W2.double()) * (h1 > 0)
    r0 = (r1 @ W1.double()) * (h0 > 0)
    # the launches this kernel replaces
    s1 = ops.project_forward([pk[0]], Operand(d, rows, 256), rows, rows, relu_mask=h1)[0]
    s0 = ops.project_forward([pk[1]], Operand(s1, rows, 256), rows, rows, relu_mask=h0)[0]
    for n_chain in (1, 2):
        for n_fan in (0, 1, 3):
            outs = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(n_chain)]
            fouts = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(n_fan)]
            chain = [(pk[0], h1, outs[0])] + ([(pk[1], h0, outs[1])] if n_chain == 2 else [])
            ag.chain_backward(d, chain, [(pk[2 + s], fouts[s]) for s in range(n_fan)])
            torch.cuda.synchronize()
            last = r0 if n_chain == 2 else r1
            want = [r1, r0][:n_chain] + [last @ W0.double()[:, 256 * s:256 * (s + 1)] for s in range(n_fan)]
            for got, ref in zip(outs + fouts, want):
                scale = ref.abs().max().item() + 1e-12
                assert (got.double() - ref).abs().max().item() <= 5e-5 * scale
            assert torch.equal(outs[0], s1)
            if n_chain == 2:
                assert torch.equal(outs[1], s0)
            slast = s0 if n_chain == 2 else s1
            for s in range(n_fan):
                single = ops.project_forward([pk[2 + s]], Operand(slast, rows, 256), rows, rows)[0]
                assert torch.equal(fouts[s], single)


@pytest.mark.parametrize("x3", [True, False])
@pytest.mark.parametrize("rows", [1, 63, 1000, 4097])
def test_layernorm_backward_as_the_prologue_of_the_input_gradient_chain(rows, x3):
    """gw_mlp_ln_chain_backward (ABI v19): the gradient at the LayerNorm's input, d gamma, d beta and the chain's rows from ONE launch
    (split and fp32 streams) against gw_layernorm_backward followed by gw_mlp_chain_backward[_bf16x3] (same formulas, other summation orders: fp32 rounding)
    and against torch's fp64 LayerNorm backward; d gamma / d beta are accumulated onto what the buffers hold; rows past the end of a
    ragged last tile add nothing to them."""
    from graph_weather_amd import _lib, autograd as ag

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(100 + rows)
    W2, W1 = (torch.randn(256, 256, generator=g) / 16).to(DEV), (torch.randn(256, 256, generator=g) / 16).to(DEV)
    W0 = (torch.randn(256, 512, generator=g) / 16).to(DEV)
    dn = torch.randn(rows, 256, generator=g).to(DEV)
    y = (torch.randn(rows, 256, generator=g) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV)
    h1, h0 = torch.randn(rows, 256, generator=g).relu().to(DEV), torch.randn(rows, 256, generator=g).relu().to(DEV)
    n = int(L.gw_packed_bytes_bf16x3(256, 0, 256)) // 2 if x3 else int(L.gw_packed_floats(256, 0, 256))
    blocks = [(W2, 0), (W1, 0), (W0, 0), (W0, 256)]
    buf = torch.empty(len(blocks) * n, dtype=torch.int16 if x3 else torch.float32, device=DEV)
    ops.pack_many(_lib.DTYPE_BF16X3 if x3 else _lib.DTYPE_F32, [(W.data_ptr() + 4 * lo, 1, int(W.shape[1]), 256, 256, buf[i * n:].data_ptr())
                                                                for i, (W, lo) in enumerate(blocks)], [], st)
    pk = [buf[i * n:(i + 1) * n] for i in range(len(blocks))]

    def run(fused):
        dg, db = torch.full((256,), 2.0, device=DEV), torch.full((256,), -3.0, device=DEV)
        outs = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(2)]
        fouts = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(2)]
        chain = [(pk[0], h1, outs[0]), (pk[1], h0, outs[1])]
        fan = [(pk[2], fouts[0]), (pk[3], fouts[1])]
        if fused:
            d = torch.full((rows, 256), float("nan"), device=DEV)
            ag.chain_backward(dn, chain, fan, ln=(y, gamma, dg, db, d))
        else:
            d = ag.layernorm_backward(dn, y, gamma, dg, db)
            ag.chain_backward(d, chain, fan)
        torch.cuda.synchronize()
        return [d, dg, db] + outs + fouts

    one, two = run(True), run(False)
    for a_, b_, name in zip(one, two, ["d", "dgamma", "dbeta", "d1", "dz0", "fan0", "fan1"]):
        scale = b_.abs().max().item() + 1e-12
        err = (a_.double() - b_.double()).abs().max().item() / scale
        assert err <= 2e-5, (name, err)
    # torch fp64
    yd = y.double().cpu().requires_grad_(True)
    gd = gamma.double().cpu().requires_grad_(True)
    bd = torch.zeros(256, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(yd, (256,), gd, bd, 1e-5).backward(dn.double().cpu())
    assert (one[0].double().cpu() - yd.grad).abs().max().item() <= 1e-5 * yd.grad.abs().max().item()
    assert (one[1].double().cpu() - 2.0 - gd.grad).abs().max().item() <= 1e-5 * (gd.grad.abs().max().item() + 2.0)
    assert (one[2].double().cpu() + 3.0 - bd.grad).abs().max().item() <= 1e-5 * (bd.grad.abs().max().item() + 3.0)
    with pytest.raises(RuntimeError):  # a LayerNorm without its buffers
        ag.chain_backward(dn, [(pk[0], h1, one[3])], [], ln=(y, gamma, None, one[2], one[0]))
    # the other two extras of the launch, with and without the LayerNorm in front: Linear_0's bias gradient (column sums of dz0,
    # accumulated) and a fan product that another gradient of the same tensor joins before the store
    addend = torch.randn(rows + 3, 260, generator=g).to(DEV)[:rows, :256]  # (its own leading dimension)
    for with_ln in (True, False):
        cs = torch.full((256,), 5.0, device=DEV)
        outs = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(2)]
        fouts = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(2)]
        d_in = dn if with_ln else two[0]
        ln = (y, gamma, torch.zeros(256, device=DEV), torch.zeros(256, device=DEV), torch.empty(rows, 256, device=DEV)) if with_ln else None
        ag.chain_backward(d_in, [(pk[0], h1, outs[0]), (pk[1], h0, outs[1])], [(pk[2], fouts[0]), (pk[3], fouts[1])], ln=ln, colsum=cs,
                          fan_add=[None, addend])
        torch.cuda.synchronize()
        ref = one if with_ln else two
        tol = 2e-5 if with_ln else 0.0  # (without the norm the launch is the plain chain: the same bits)
        for got, want, name in ((outs[0], ref[3], "d1"), (outs[1], ref[4], "dz0"), (fouts[0], ref[5], "fan0"), (fouts[1], ref[6] + addend, "fan1+add")):
            assert (got.double() - want.double()).abs().max().item() <= tol * (want.abs().max().item() + 1e-12) + (1e-6 if name == "fan1+add" else 0.0), name
        want_cs = 5.0 + outs[1].double().sum(0)
        assert (cs.double() - want_cs).abs().max().item() <= 1e-5 * (want_cs.abs().max().item() + 1.0)


@pytest.mark.parametrize("x3", [True, False])
def test_chain_launch_gathers_its_input_gradient_in_place(x3):
    """gw_mlp_ln_chain_backward with dn_idx / dn_add: the launch reads row dagg[b * n_dst + dst[k]] + de_out[b * E + k] itself - the
    same bits as the launch on the materialised gather (gw_gather_rows), for every output, with the input gradient joining a fan
    product (fan_add_dn_mask; with de_out the two rows are added to the product one after the other: fp32 rounding) and a ragged last tile; the C entry refuses a gather without the LayerNorm in front."""
    from graph_weather_amd import _lib, autograd as ag

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(77)
    B, n_dst, E = 3, 50, 333
    rows = B * E
    W2, W1 = (torch.randn(256, 256, generator=g) / 16).to(DEV), (torch.randn(256, 256, generator=g) / 16).to(DEV)
    W0 = (torch.randn(256, 256, generator=g) / 16).to(DEV)
    dagg = torch.randn(B * n_dst, 256, generator=g).to(DEV)
    dst = torch.sort(torch.randint(0, n_dst, (E,), generator=g)).values.to(torch.int32).to(DEV)
    de_out = torch.randn(rows, 256, generator=g).to(DEV)
    y = (torch.randn(rows, 256, generator=g) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV)
    h1, h0 = torch.randn(rows, 256, generator=g).relu().to(DEV), torch.randn(rows, 256, generator=g).relu().to(DEV)
    n = int(L.gw_packed_bytes_bf16x3(256, 0, 256)) // 2 if x3 else int(L.gw_packed_floats(256, 0, 256))
    blocks = [(W2, 0), (W1, 0), (W0, 0)]
    buf = torch.empty(len(blocks) * n, dtype=torch.int16 if x3 else torch.float32, device=DEV)
    ops.pack_many(_lib.DTYPE_BF16X3 if x3 else _lib.DTYPE_F32, [(W.data_ptr() + 4 * lo, 1, int(W.shape[1]), 256, 256, buf[i * n:].data_ptr())
                                                                for i, (W, lo) in enumerate(blocks)], [], st)
    pk = [buf[i * n:(i + 1) * n] for i in range(len(blocks))]
    for add in (de_out, None):
        res = []
        for in_place in (True, False):
            dg, db = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
            d = torch.full((rows, 256), float("nan"), device=DEV)
            outs = [torch.full((rows, 256), float("nan"), device=DEV) for _ in range(2)]
            fout = torch.full((rows, 256), float("nan"), device=DEV)
            chain = [(pk[0], h1, outs[0]), (pk[1], h0, outs[1])]
            if in_place:
                ag.chain_backward(dagg, chain, [(pk[2], fout)], ln=(y, gamma, dg, db, d), fan_add=[ag.FAN_ADD_INPUT], gather=(dst, n_dst, add))
            else:
                dn = ag.gather_rows(dagg, n_dst, dst, B, E, add)
                ag.chain_backward(dn, chain, [(pk[2], fout)], ln=(y, gamma, dg, db, d), fan_add=[dn])
            torch.cuda.synchronize()
            res.append([d, dg, db] + outs + [fout])
        for a_, b_, name in zip(res[0], res[1], ["d", "dgamma", "dbeta", "d1", "dz0", "fan+dn"]):
            if name in ("dgamma", "dbeta"):  # (atomics: the order of the workgroups)
                assert (a_ - b_).abs().max().item() <= 1e-4 * (b_.abs().max().item() + 1e-6), name
            elif name == "fan+dn" and add is not None:  # (product + dagg row + de_out row against product + their rounded sum)
                assert (a_ - b_).abs().max().item() <= 1e-6 * b_.abs().max().item(), name
            else:
                assert torch.equal(a_, b_), (name, add is not None)
    with pytest.raises(RuntimeError):
        ag.chain_backward(dagg, [(pk[0], h1, torch.empty(rows, 256, device=DEV))], [], gather=(dst, n_dst, None))
