"""Round-3 GPU tests: the team-pipelined bf16 edge kernel (csrc/gw_edge16t.hip) in the forms the small operator tests of
test_gpu_edge16.py do not reach (batch chunks with the shared layer-1 part cached per chunk; the form without residual),
the optimizer state round trip of the flat AdamW, and a GraphCast whose processor alone is wider than the fused kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from graph_weather_amd import ops  # noqa: E402
from graph_weather_amd.ops import Operand, PackedMLP  # noqa: E402

DEV = "cuda:0"


def _edge_mlp(rs):
    W0 = torch.from_numpy((rs.standard_normal((256, 768)) / 16).astype(np.float32))
    W1 = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32))
    W2 = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32))
    b = [torch.from_numpy((0.1 * rs.standard_normal(256)).astype(np.float32)) for _ in range(3)]
    gamma = torch.from_numpy((1 + 0.1 * rs.standard_normal(256)).astype(np.float32))
    beta = torch.from_numpy((0.1 * rs.standard_normal(256)).astype(np.float32))
    return PackedMLP([W0.to(DEV), W1.to(DEV), W2.to(DEV)], [x.to(DEV) for x in b], (gamma.to(DEV), beta.to(DEV)),
                     ((0, 256), (256, 512), (512, 768)), torch.bfloat16)


@pytest.mark.timeout(600)
def test_team_kernel_batch_chunks_agree_with_the_lock_step_kernel():
    """Decoder-shaped launch large enough (>= 16 units per workgroup with 2 samples per unit) that the team kernel walks the batch
    in chunks and caches the batch-shared layer-1 part (b1 + We.e rows, fp16) per chunk: against the deterministic lock-step
    kernel on the same operands (same bf16 products; differences = fp32 summation order, the fp16 cache in front of a bf16
    rounding, v_rsq - a few 1e-3 of the scale, where an indexing mistake in the chunk walk would be O(1))."""
    rs = np.random.RandomState(5)
    B, n_src, n_dst = 2, 3000, 38000
    E = 64 * 4200 + 17  # 4201 edge blocks: chunks of 2 samples leave 4201 units >= 16 x 256 workgroups
    pm = _edge_mlp(rs)
    g = torch.Generator(device="cpu").manual_seed(3)
    dst = torch.sort(torch.randint(0, n_dst, (E,), generator=g)).values.int().to(DEV)
    src = torch.randint(0, n_src, (E,), generator=g).int().to(DEV)
    ps = torch.randn((B * n_src, 256), generator=g).to(DEV)
    pe = torch.randn((E, 256), generator=g).to(DEV)
    e = torch.randn((E, 256), generator=g).to(DEV)
    tiles = ops.edge_rows_to_tiles(e, 1, E, E)
    args = (pm, B, src, dst, Operand(ps, n_src, 256, projected=True), ops.ZERO, Operand(pe, 0, 256, projected=True))
    agg_team = torch.zeros((B * n_dst, 256), device=DEV)
    ops.edge_update_forward(*args, ops.ZERO, n_dst, agg_team, None)  # team kernel: no residual ...
    e_sum = torch.zeros((n_dst, 256), device=DEV).index_add_(0, dst.long(), e.to(torch.bfloat16).float())  # (test arithmetic)
    agg_team = agg_team.reshape(B, n_dst, 256) + e_sum[None]  # ... + the per-destination sums of the bf16-rounded e
    agg_lock = torch.zeros((B * n_dst, 256), device=DEV)
    ops.edge_update_forward(*args, Operand(tiles, 0, 256, tiles=True), n_dst, agg_lock, None, deterministic=True)
    torch.cuda.synchronize()
    agg_lock = agg_lock.reshape(B, n_dst, 256)
    scale = agg_lock.abs().max().item()
    err = (agg_team - agg_lock).abs().max().item()
    print(f"[team chunks] E={E} B={B}: team vs lock-step max {err:.3e} of scale {scale:.3e}")
    assert 0 < err <= 1e-2 * scale


def test_fp16_product_rows_equal_fp32_rows_of_the_same_values():
    """GW_LAYOUT_ROWS_F16: the layer-1 node products as fp16 rows give what fp32 rows holding the same (fp16-representable) values
    give - in the team kernel's gather (decoder form) and in the layer-1 kernel of a block with per-sample edge tiles."""
    rs = np.random.RandomState(21)
    B, n_src, n_dst, E = 3, 90, 64, 1100
    pm = _edge_mlp(rs)
    dst_np = np.sort(rs.randint(0, n_dst, size=E))
    dst = torch.from_numpy(dst_np.astype(np.int32)).to(DEV)
    src = torch.from_numpy(rs.randint(0, n_src, size=E).astype(np.int32)).to(DEV)
    ps16 = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32)).to(DEV).half()
    pd16 = torch.from_numpy(rs.standard_normal((B * n_dst, 256)).astype(np.float32)).to(DEV).half()
    pe = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32)).to(DEV)
    # decoder form: one per-sample table, shared edge products, no residual
    res = []
    for tab in (ps16, ps16.float()):
        agg = torch.zeros((B * n_dst, 256), device=DEV)
        ops.edge_update_forward(pm, B, src, dst, Operand(tab, n_src, 256, projected=True), ops.ZERO, Operand(pe, 0, 256, projected=True),
                                ops.ZERO, n_dst, agg, None)
        res.append(agg)
    torch.cuda.synchronize()
    assert (res[0] - res[1]).abs().max().item() <= 2e-5 * res[1].abs().max().item()
    # processor form: per-sample edge tiles as the raw operand and residual, both node tables per sample
    e = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32)).to(DEV)
    tiles = ops.edge_rows_to_tiles(e, B, E, E)
    res = []
    for a_, b_ in ((ps16, pd16), (ps16.float(), pd16.float())):
        agg = torch.zeros((B * n_dst, 256), device=DEV)
        e_out = torch.empty(ops.edge_tiles_bytes(B, E), dtype=torch.uint8, device=DEV)
        ops.edge_update_forward(pm, B, src, dst, Operand(a_, n_src, 256, projected=True), Operand(b_, n_dst, 256, projected=True),
                                Operand(tiles, E, 256, tiles=True), Operand(tiles, E, 256, tiles=True), n_dst, agg, e_out)
        res.append((agg, e_out))
    torch.cuda.synchronize()
    assert (res[0][0] - res[1][0]).abs().max().item() <= 2e-5 * res[1][0].abs().max().item()
    assert torch.equal(res[0][1], res[1][1])  # e' tiles: bitwise (same products, same order inside a tile)
    # fp16 rows are a format of projected operands of the resident-weight path only
    with pytest.raises(RuntimeError):
        ops.edge_update_forward(pm, B, src, dst, Operand(ps16, n_src, 256, projected=False), ops.ZERO, Operand(pe, 0, 256, projected=True),
                                ops.ZERO, n_dst, res[0][0], None)


def test_team_kernel_without_residual_plus_segment_sums_of_e():
    """sum(LN(.) + e) = sum(LN(.)) + sum(e): the launch without residual (what the decoder uses: e' is dropped and e is batch
    shared) plus the per-destination sums of the bf16-rounded e equals the launch with the residual tiles."""
    rs = np.random.RandomState(9)
    B, n_src, n_dst, E = 3, 70, 50, 900
    pm = _edge_mlp(rs)
    dst_np = np.sort(np.where(rs.rand(E) < 0.3, 7, rs.randint(0, n_dst, size=E)))
    dst = torch.from_numpy(dst_np.astype(np.int32)).to(DEV)
    src = torch.from_numpy(rs.randint(0, n_src, size=E).astype(np.int32)).to(DEV)
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32)).to(DEV)
    pe = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32)).to(DEV)
    e = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32)).to(DEV)
    args = (pm, B, src, dst, Operand(ps, n_src, 256, projected=True), ops.ZERO, Operand(pe, 0, 256, projected=True))
    agg_res = torch.zeros((B * n_dst, 256), device=DEV)
    ops.edge_update_forward(*args, Operand(ops.edge_rows_to_tiles(e, 1, E, E), 0, 256, tiles=True), n_dst, agg_res, None)
    agg_no = torch.zeros((B * n_dst, 256), device=DEV)
    ops.edge_update_forward(*args, ops.ZERO, n_dst, agg_no, None)
    torch.cuda.synchronize()
    e_sum = torch.zeros((n_dst, 256), device=DEV).index_add_(0, dst.long(), e.to(torch.bfloat16).float())  # (test arithmetic)
    want = agg_res.reshape(B, n_dst, 256)
    got = agg_no.reshape(B, n_dst, 256) + e_sum[None]
    err = (got - want).abs().max().item()
    # (the residual form runs on the lock-step kernel, the other on the team kernel: same bf16 products; fp16 cache of the shared
    #  layer-1 part, v_rsq and summation order differ - a few 1e-3 of the scale; a dropped or doubled residual would be O(1))
    assert err <= 1e-2 * want.abs().max().item(), err
    # e' cannot be produced without its residual operand
    with pytest.raises(RuntimeError):
        ops.edge_update_forward(*args, ops.ZERO, n_dst, agg_no, torch.empty((B * E, 256), device=DEV))


def test_flat_adamw_state_dict_round_trip():
    """The flat optimizer state (moments of the whole model + the bias-correction step) survives state_dict() / load_state_dict():
    a resumed optimizer takes the same next step as the one that kept running."""
    import graph_weather_amd as gw
    from graph_weather_amd import sharding as sh

    def make():
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 5)).to(DEV)
        flat = sh.FlatGradients(m.parameters())
        return m, flat, gw.AdamW(m.parameters(), lr=1e-2, flat=flat)

    x, y = torch.randn(16, 12, device=DEV), torch.randn(16, 5, device=DEV)

    def step(m, flat, opt):
        flat.zero_()
        (m(x) - y).square().mean().backward()
        opt.step()

    m1, f1, o1 = make()
    for _ in range(3):
        step(m1, f1, o1)
    sd = o1.state_dict()
    assert sd["gw_flat_state"]["step"] == 3 and sd["gw_flat_state"]["exp_avg"].abs().sum().item() > 0
    m2, f2, o2 = make()
    with torch.no_grad():
        f2.param.copy_(f1.param)
    o2.load_state_dict(sd)
    step(m1, f1, o1)
    step(m2, f2, o2)
    torch.cuda.synchronize()
    assert torch.equal(f1.param, f2.param)
    m3, f3, o3 = make()
    with pytest.raises(ValueError):
        o3.load_state_dict({k: v for k, v in sd.items() if k != "gw_flat_state"})


@pytest.mark.timeout(600)
def test_graphcast_with_a_wide_processor_only():
    """A GraphCast whose processor hidden width alone exceeds 256 (node / edge tables stay 256 wide): wide versus fused is decided
    once for the whole model - the forward runs (it used to hand a 256-padded table to the wide blocks) and agrees with the
    oracle."""
    import graph_weather_amd as gw
    from graph_weather_amd.layers import Processor
    from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features
    from oracle import reference_math as om

    ll = regular_lat_lons(30.0)
    m = gw.GraphCast(ll, input_dim=78, output_dim=78, hidden_dim=256, num_processor_blocks=2)
    m.processor = Processor(input_dim=256, edge_dim=256, num_blocks=2, hidden_dim_processor_node=320, hidden_dim_processor_edge=320)
    deterministic_fill_(m, seed=4)
    feats = seeded_features(2, len(ll), 78, seed=6)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    want = om.forecaster_forward(state, m.encoder.graphs.as_oracle_dict(), feats)
    m = m.to(DEV).eval()
    with torch.no_grad():
        got = m(feats.to(DEV))
    torch.cuda.synchronize()
    scale = (want - feats).abs().max().item()
    err = (got.cpu() - want).abs().max().item()
    assert err <= 2e-4 * max(scale, 1.0), (err, scale)


def test_pack_many_equals_the_single_item_packers_bitwise():
    """gw_pack_many (ABI v13: every slice / vector of an MLP in one launch, items addressed by strides) produces the streams of
    gw_pack_linear / gw_pack_linear_bf16 / gw_pad_vector bit for bit: column slices, a head with 78 rows, narrow K variants, and
    the transposed block of the backward (against packing an explicitly transposed copy)."""
    from graph_weather_amd import _lib, ops

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(3)
    W0 = torch.randn(256, 768, device=DEV)
    Wh = torch.randn(78, 128, device=DEV)
    Wk = torch.randn(256, 102, device=DEV)
    W2 = torch.randn(256, 2, device=DEV)
    vs = [torch.randn(n, device=DEV) for n in (256, 78, 128, 5)]
    cases = [(W0, 0, 256), (W0, 256, 512), (W0, 512, 768), (Wh, 0, 128), (Wk, 0, 102), (W2, 0, 2)]
    for bf16 in (False, True):
        dt = _lib.DTYPE_BF16 if bf16 else _lib.DTYPE_F32
        want, got, mats = [], [], []
        for w, lo, hi in cases:
            if bf16:
                nb = L.gw_packed_bytes_bf16(int(w.shape[0]), lo, hi)
                a = torch.zeros(nb // 2, dtype=torch.bfloat16, device=DEV)
                b = torch.ones_like(a)
                _lib.check(L.gw_pack_linear_bf16(w.data_ptr(), int(w.shape[0]), int(w.shape[1]), lo, hi, a.data_ptr(), st), "pack")
            else:
                n = L.gw_packed_floats(int(w.shape[0]), lo, hi)
                a = torch.zeros(n, device=DEV)
                b = torch.ones_like(a)
                _lib.check(L.gw_pack_linear(w.data_ptr(), int(w.shape[0]), int(w.shape[1]), lo, hi, a.data_ptr(), st), "pack")
            want.append(a)
            got.append(b)
            mats.append((w.data_ptr() + 4 * lo, int(w.shape[1]), 1, int(w.shape[0]), hi - lo, b.data_ptr()))
        # the transposed 256 x 256 block W0[:, 256:512]^T, in place, against a packed explicit transpose
        Wt = W0[:, 256:512].t().contiguous()
        if bf16:
            a = torch.zeros(L.gw_packed_bytes_bf16(256, 0, 256) // 2, dtype=torch.bfloat16, device=DEV)
            _lib.check(L.gw_pack_linear_bf16(Wt.data_ptr(), 256, 256, 0, 256, a.data_ptr(), st), "pack")
        else:
            a = torch.zeros(L.gw_packed_floats(256, 0, 256), device=DEV)
            _lib.check(L.gw_pack_linear(Wt.data_ptr(), 256, 256, 0, 256, a.data_ptr(), st), "pack")
        b = torch.ones_like(a)
        want.append(a)
        got.append(b)
        mats.append((W0.data_ptr() + 4 * 256, 1, 768, 256, 256, b.data_ptr()))
        vwant, vgot, vecs = [], [], []
        for v in vs:
            a = torch.zeros(L.gw_padded_n(int(v.shape[0])), device=DEV)
            b = torch.ones_like(a)
            _lib.check(L.gw_pad_vector(v.data_ptr(), int(v.shape[0]), a.data_ptr(), st), "pad")
            vwant.append(a)
            vgot.append(b)
            vecs.append((v.data_ptr(), int(v.shape[0]), b.data_ptr()))
        ops.pack_many(dt, mats, vecs, st)
        torch.cuda.synchronize()
        for a, b in zip(want + vwant, got + vgot):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    # an output head: 78 rows packed as the 80 rows its kernel variant walks == packing the explicitly zero-padded matrix / vector
    Wh80 = torch.cat([Wh, Wh.new_zeros(2, 128)])
    b78 = torch.randn(78, device=DEV)
    b80 = torch.cat([b78, b78.new_zeros(2)])
    for n_small in (78, 32):
        Ws_, bs_ = Wh[:n_small].contiguous(), b78[:n_small].contiguous()
        W80 = torch.cat([Ws_, Ws_.new_zeros(80 - n_small, 128)])
        v80 = torch.cat([bs_, bs_.new_zeros(80 - n_small)])
        a = torch.zeros(L.gw_packed_floats(80, 0, 128), device=DEV)
        _lib.check(L.gw_pack_linear(W80.data_ptr(), 80, 128, 0, 128, a.data_ptr(), st), "pack")
        b = torch.ones_like(a)
        va = torch.zeros(L.gw_padded_n(80), device=DEV)
        _lib.check(L.gw_pad_vector(v80.data_ptr(), 80, va.data_ptr(), st), "pad")
        vb = torch.ones_like(va)
        ops.pack_many(_lib.DTYPE_F32, [(Ws_.data_ptr(), 128, 1, n_small, 128, b.data_ptr(), 80)], [(bs_.data_ptr(), n_small, vb.data_ptr(), 80)], st)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(va, vb)
    # more items than one launch takes: chunked by the host wrapper
    many = [(W0.data_ptr(), 768, 1, 256, 256, torch.empty(L.gw_packed_floats(256, 0, 256), device=DEV)) for _ in range(20)]
    ops.pack_many(_lib.DTYPE_F32, [m[:5] + (m[5].data_ptr(),) for m in many], [], st)
    torch.cuda.synchronize()
    ref = torch.empty(L.gw_packed_floats(256, 0, 256), device=DEV)
    _lib.check(L.gw_pack_linear(W0.data_ptr(), 256, 768, 0, 256, ref.data_ptr(), st), "pack")
    torch.cuda.synchronize()
    assert all(torch.equal(m[5], ref) for m in many)
    with pytest.raises(RuntimeError):
        ops.pack_many(7, [many[0][:5] + (many[0][5].data_ptr(),)], [], st)


@pytest.mark.parametrize("rows", [1, 64, 1000, 4097])
def test_chain_backward_kernel_against_torch(rows):
    """gw_mlp_chain_backward (ABI v13): d1 = (d W2) * (h1 > 0), dz0 = (d1 W1) * (h0 > 0), fan products dz0 W0[:, block] - one
    launch - against the same products in torch fp64 (ragged row counts; 1, 2 chain links; 0..3 fan blocks)."""
    from graph_weather_amd import _lib, autograd as ag, ops

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(rows)
    W2, W1 = (torch.randn(256, 256, generator=g) / 16).to(DEV), (torch.randn(256, 256, generator=g) / 16).to(DEV)
    W0 = (torch.randn(256, 768, generator=g) / 16).to(DEV)
    d = torch.randn(rows, 256, generator=g).to(DEV)
    h1, h0 = torch.randn(rows, 256, generator=g).relu().to(DEV), torch.randn(rows, 256, generator=g).relu().to(DEV)
    n = int(L.gw_packed_floats(256, 0, 256))
    blocks = [(W2, 0), (W1, 0), (W0, 0), (W0, 256), (W0, 512)]
    buf = torch.empty(len(blocks) * n, device=DEV)
    ops.pack_many(_lib.DTYPE_F32, [(W.data_ptr() + 4 * lo, 1, int(W.shape[1]), 256, 256, buf[i * n:].data_ptr())
                                   for i, (W, lo) in enumerate(blocks)], [], st)
    pk = [buf[i * n:(i + 1) * n] for i in range(len(blocks))]
    dd = d.double()
    r1 = (dd @ W2.double()) * (h1 > 0)
    r0 = (r1 @ W1.double()) * (h0 > 0)
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
                assert (got.double() - ref).abs().max().item() <= 2e-5 * scale
    with pytest.raises(RuntimeError):
        ag.chain_backward(d, [], [])


def test_node_update_takes_fp16_projected_rows_and_other_layouts_are_rejected():
    """bf16 node update (+ head): a projected x operand as fp16 rows gives the result of the same values as fp32 rows; layouts an
    entry point does not read are refused instead of being misread as fp32 (fp16 rows with fp32 weights, as a raw operand, as agg)."""
    import numpy as np
    from graph_weather_amd import ops
    from graph_weather_amd.ops import Operand, PackedMLP

    rs = np.random.RandomState(11)
    def mk(dims, norm):
        ws = [torch.from_numpy((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)).to(DEV) for i in range(3)]
        bs = [torch.from_numpy((0.1 * rs.standard_normal(dims[i + 1])).astype(np.float32)).to(DEV) for i in range(3)]
        ln = (torch.ones(dims[3], device=DEV), torch.zeros(dims[3], device=DEV)) if norm else None
        return ws, bs, ln
    node, head = mk([512, 256, 256, 256], True), mk([256, 128, 128, 78], False)
    rows, B = 300, 2
    n = rows * B
    xp = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32)).to(DEV).half()  # (values exactly representable in fp16)
    agg = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    for dt in (torch.bfloat16,):
        pm = PackedMLP(node[0], node[1], node[2], ((0, 256), (256, 512)), dt)
        hd = PackedMLP(head[0], head[1], None, ((0, 256),), dt)
        a = ops.node_update_forward(pm, n, rows, Operand(xp.float(), 0, 256, projected=True), ops.ZERO, Operand(agg, rows, 256))
        b = ops.node_update_forward(pm, n, rows, Operand(xp, 0, 256, projected=True), ops.ZERO, Operand(agg, rows, 256))
        c = ops.node_update_head_forward(pm, hd, n, rows, Operand(xp.float(), 0, 256, projected=True), Operand(agg, rows, 256), None)
        d = ops.node_update_head_forward(pm, hd, n, rows, Operand(xp, 0, 256, projected=True), Operand(agg, rows, 256), None)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(c, d)
    pm32 = PackedMLP(node[0], node[1], node[2], ((0, 256), (256, 512)), torch.float32)
    with pytest.raises(RuntimeError):
        ops.node_update_forward(pm32, n, rows, Operand(xp, 0, 256, projected=True), ops.ZERO, Operand(agg, rows, 256))
    with pytest.raises(RuntimeError):  # fp16 rows are a format of projected operands
        ops.node_update_forward(pm, n, rows, Operand(xp, 0, 256), ops.ZERO, Operand(agg, rows, 256))
