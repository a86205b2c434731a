"""Out-of-bounds audit of the C-ABI launches: every buffer handed to the library sits between two 64 KiB canary regions
filled with 0xFF bytes (a NaN as fp32 / bf16, -1 as an index).  A write outside a buffer breaks a canary; a read outside
one pulls a NaN into an MFMA (NaN x 0 is still NaN) or a wild index into a gather, so the guarded run no longer matches the
plain one.  Shapes are ragged on purpose: rows and edges that are not multiples of the 16 / 64 / 128-column tiles, K = 102 and
K = 78 operands (padded K-steps of the packed layer-1 stream), one-tile and sub-tile inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from graph_weather_amd import ops  # noqa: E402
from graph_weather_amd.ops import Operand, PackedMLP  # noqa: E402

DEV = "cuda:0"
PAD = 1 << 16
X3 = ops.BF16X3  # split-operand mode (csrc/gw_split.hip): the same entry points, fp32 rows everywhere


class Guards:
    """wrap(t): a copy of ``t`` on the device, between canaries; check(): the canaries are intact."""

    def __init__(self, on: bool):
        self.on = on
        self.bufs = []

    def wrap(self, t):
        if t is None:
            return None
        t = t.to(DEV).contiguous()
        if not self.on:
            return t.clone()
        nbytes = t.numel() * t.element_size()
        raw = torch.full((2 * PAD + nbytes,), 0xFF, dtype=torch.uint8, device=DEV)
        inner = raw[PAD:PAD + nbytes].view(t.dtype).view(t.shape)
        inner.copy_(t)
        self.bufs.append((raw, nbytes))
        return inner

    def wrap_packed(self, pm: PackedMLP) -> PackedMLP:
        pm.w1 = [self.wrap(w) for w in pm.w1]
        for name in ("b1", "w_mid", "b_mid", "w_out", "b_out", "gamma", "beta"):
            setattr(pm, name, self.wrap(getattr(pm, name)))
        return pm

    def check(self):
        torch.cuda.synchronize()
        for i, (raw, n) in enumerate(self.bufs):
            assert bool((raw[:PAD] == 0xFF).all()), f"buffer {i} ({n} bytes): write BEFORE the buffer"
            assert bool((raw[PAD + n:] == 0xFF).all()), f"buffer {i} ({n} bytes): write PAST the buffer"


def _mlp(rs, in_dim, hidden, out, norm, dtype):
    dims = [in_dim, hidden, hidden, out]
    ws = [torch.from_numpy((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(3)]
    bs = [torch.from_numpy((0.1 * rs.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    ln = (torch.from_numpy((1 + 0.1 * rs.standard_normal(out)).astype(np.float32)),
          torch.from_numpy((0.1 * rs.standard_normal(out)).astype(np.float32))) if norm else None
    return ws, bs, ln


def _packed(g, mlp, splits, dtype):
    ws, bs, ln = mlp
    pm = PackedMLP([w.to(DEV) for w in ws], [b.to(DEV) for b in bs], None if ln is None else (ln[0].to(DEV), ln[1].to(DEV)),
                   splits, dtype)
    return g.wrap_packed(pm)


def _same(a, b, what, tol):
    assert torch.isfinite(b).all(), f"{what}: non-finite values in the guarded run (a read outside a buffer)"
    scale = a.abs().max().item() + 1e-30
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{what}: guarded run differs from the plain run by {err:.3e}"


def _both(fn):
    """fn(guards) -> dict of output tensors.  Plain run, guarded run, canaries, equality (atomics: summation order only)."""
    plain = {k: v.float().cpu() for k, v in fn(Guards(False)).items()}
    g = Guards(True)
    guarded = fn(g)
    g.check()
    for k, v in guarded.items():
        _same(plain[k], v.float().cpu(), k, 2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
@pytest.mark.parametrize("rows,in_dim,hidden,out,norm", [(1000, 102, 256, 256, True), (77, 78, 256, 256, True), (1, 256, 256, 256, True),
                                                        (1000, 256, 128, 78, False), (130, 31, 256, 256, True), (453, 2, 256, 256, True),
                                                        (129, 256, 256, 256, False)])
def test_mlp_forward_stays_inside_its_buffers(dtype, rows, in_dim, hidden, out, norm):
    rs = np.random.RandomState(rows + in_dim)
    mlp = _mlp(rs, in_dim, hidden, out, norm, dtype)
    x = torch.from_numpy(rs.standard_normal((rows, in_dim)).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((rows, 102)).astype(np.float32)) if out == 78 else None

    def run(g):
        pm = _packed(g, mlp, ((0, in_dim),), dtype)
        o = g.wrap(torch.zeros(rows, out))
        ops.mlp_forward(pm, Operand(g.wrap(x), rows, in_dim), rows, rows,
                        residual=None if res is None else Operand(g.wrap(res), rows, out), out=o)
        return {"out": o}

    _both(run)


def _graph(rs, n_src, n_dst, E):
    dst = np.sort(np.where(rs.rand(E) < 0.3, n_dst // 2, rs.randint(0, n_dst, size=E))).astype(np.int32)
    if E > 2:
        dst[-1] = n_dst - 1  # the last destination row is written
    src = rs.randint(0, n_src, size=E).astype(np.int32)
    src[0] = n_src - 1       # ... and the last source row is read
    return torch.from_numpy(src), torch.from_numpy(dst)


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
@pytest.mark.parametrize("B,n_src,n_dst,E", [(2, 50, 40, 333), (1, 7, 5, 64), (3, 9, 4, 5), (2, 30, 30, 129)])
def test_edge_update_raw_operands_stays_inside_its_buffers(dtype, deterministic, B, n_src, n_dst, E):
    """General kernels (encoder form): gathered raw node rows, raw per-sample edge rows."""
    rs = np.random.RandomState(E)
    mlp = _mlp(rs, 768, 256, 256, True, dtype)
    src, dst = _graph(rs, n_src, n_dst, E)
    xs = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))
    xd = torch.from_numpy(rs.standard_normal((n_dst, 256)).astype(np.float32))
    e = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), dtype)
        agg = g.wrap(torch.zeros(B * n_dst, 256))
        e_out = g.wrap(torch.zeros(B * E, 256))
        s, d = g.wrap(src), g.wrap(dst)
        ev = g.wrap(e)
        ops.edge_update_forward(pm, B, s, d, Operand(g.wrap(xs), n_src, 256, index=s), Operand(g.wrap(xd), 0, 256, index=d),
                                Operand(ev, E, 256), Operand(ev, E, 256), n_dst, agg, e_out, deterministic=deterministic)
        return {"agg": agg, "e_out": e_out}

    if deterministic and dtype == torch.float32:
        # the general fp32 kernel has no carry records (the forecaster projects the node operands first in that mode): loud
        with pytest.raises(RuntimeError, match="deterministic"):
            run(Guards(False))
        return
    _both(run)


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
@pytest.mark.parametrize("B,n_src,n_dst,E,use_dst,want_e", [(2, 50, 40, 333, True, True), (1, 7, 5, 64, False, False),
                                                            (3, 9, 4, 5, True, True), (16, 20, 30, 1000, False, False)])
def test_edge_update_projected_operands_stays_inside_its_buffers(dtype, deterministic, B, n_src, n_dst, E, use_dst, want_e):
    """Fast kernels (decoder / first processor block form): every layer-1 operand is a table of projected rows."""
    rs = np.random.RandomState(E + 1)
    mlp = _mlp(rs, 768, 256, 256, True, dtype)
    src, dst = _graph(rs, n_src, n_dst, E)
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))
    pd = torch.from_numpy(rs.standard_normal((n_dst, 256)).astype(np.float32))
    pe = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))
    er = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), dtype)
        agg = g.wrap(torch.zeros(B * n_dst, 256))
        e_out = g.wrap(torch.zeros(B * E, 256)) if want_e else None
        ops.edge_update_forward(pm, B, g.wrap(src), g.wrap(dst), Operand(g.wrap(ps), n_src, 256, projected=True),
                                Operand(g.wrap(pd), 0, 256, projected=True) if use_dst else ops.ZERO,
                                Operand(g.wrap(pe), 0, 256, projected=True), Operand(g.wrap(er), 0, 256), n_dst, agg, e_out,
                                deterministic=deterministic)
        return {"agg": agg} if e_out is None else {"agg": agg, "e_out": e_out}

    _both(run)


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("B,n_src,n_dst,E,out_kind", [(2, 50, 40, 333, "tiles"), (1, 7, 5, 64, "rows"), (3, 9, 4, 5, "tiles"),
                                                      (4, 30, 30, 700, None)])
def test_edge_update_tile_operands_stays_inside_its_buffers(deterministic, B, n_src, n_dst, E, out_kind):
    """Processor-block form of the bf16 path: per-sample bf16 edge tiles in, tiles (or rows, or nothing) out."""
    rs = np.random.RandomState(E + 2)
    mlp = _mlp(rs, 768, 256, 256, True, torch.bfloat16)
    src, dst = _graph(rs, n_src, n_dst, E)
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))
    pd = torch.from_numpy(rs.standard_normal((B * n_dst, 256)).astype(np.float32))
    e = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
        tiles = g.wrap(ops.edge_rows_to_tiles(e.to(DEV), B, E, E))
        agg = g.wrap(torch.zeros(B * n_dst, 256))
        if out_kind == "tiles":
            e_out = g.wrap(torch.zeros(ops.edge_tiles_bytes(B, E), dtype=torch.uint8))
        else:
            e_out = g.wrap(torch.zeros(B * E, 256)) if out_kind == "rows" else None
        et = Operand(tiles, E, 256, tiles=True)
        ops.edge_update_forward(pm, B, g.wrap(src), g.wrap(dst), Operand(g.wrap(ps), n_src, 256, projected=True),
                                Operand(g.wrap(pd), n_dst, 256, projected=True), et, et, n_dst, agg, e_out,
                                deterministic=deterministic)
        out = {"agg": agg}
        if e_out is not None:
            out["e_out"] = e_out.view(torch.bfloat16) if out_kind == "tiles" else e_out
        return out

    _both(run)


def test_rows_to_tiles_stays_inside_its_buffers():
    rs = np.random.RandomState(5)
    for B, E, shared in ((3, 130, False), (1, 64, False), (2, 5, True), (2, 1000, False)):
        rows = torch.from_numpy(rs.standard_normal(((1 if shared else B) * E, 256)).astype(np.float32))

        def run(g):
            r = g.wrap(rows)
            t = ops.edge_rows_to_tiles(r, B, E, 0 if shared else E)
            return {"tiles": t.view(torch.bfloat16)}

        _both(run)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
@pytest.mark.parametrize("rows,B,n_post", [(1000, 2, 3), (77, 1, 2), (5882, 1, 0), (130, 2, 1)])
def test_node_update_with_post_products_stays_inside_its_buffers(dtype, rows, B, n_post):
    rs = np.random.RandomState(rows)
    mlp = _mlp(rs, 512, 256, 256, True, dtype)
    post = [torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32)) for _ in range(n_post)]
    n = rows * B
    x = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    a = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512)), dtype)
        L = ops._lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        pw = []
        for w in post:  # packed [256, 256] slices of the next block's layer 1
            wd = w.to(DEV)
            if dtype == X3:
                buf = torch.empty(L.gw_packed_bytes_bf16x3(256, 0, 256) // 2, dtype=torch.int16, device=DEV)
                ops._lib.check(L.gw_pack_linear_bf16x3(wd.data_ptr(), 256, 256, 0, 256, buf.data_ptr(), st), "pack")
            elif dtype == torch.bfloat16:
                buf = torch.empty(L.gw_packed_bytes_bf16(256, 0, 256) // 2, dtype=torch.bfloat16, device=DEV)
                ops._lib.check(L.gw_pack_linear_bf16(wd.data_ptr(), 256, 256, 0, 256, buf.data_ptr(), st), "pack")
            else:
                buf = torch.empty(L.gw_packed_floats(256, 0, 256), dtype=torch.float32, device=DEV)
                ops._lib.check(L.gw_pack_linear(wd.data_ptr(), 256, 256, 0, 256, buf.data_ptr(), st), "pack")
            pw.append(g.wrap(buf))
        xv = g.wrap(x)
        out = g.wrap(torch.zeros(n, 256))
        zero = g.wrap(torch.ones(n, 256)) if n_post else None
        r = ops.node_update_forward(pm, n, rows, Operand(xv, rows, 256), Operand(xv, rows, 256), Operand(g.wrap(a), rows, 256),
                                    out=out, post_w=pw if n_post else None, zero_rows=zero)
        res = {"out": out}
        if n_post:
            for i, p in enumerate(r[1]):
                res[f"post{i}"] = p
            res["zero"] = zero
        return res

    _both(run)


def _pack_bf16(w, dtype=torch.bfloat16):
    L = ops._lib.lib()
    wd = w.to(DEV).contiguous()
    n, k = int(wd.shape[0]), int(wd.shape[1])
    st = torch.cuda.current_stream().cuda_stream
    if dtype == X3:
        buf = torch.empty(L.gw_packed_bytes_bf16x3(n, 0, k) // 2, dtype=torch.int16, device=DEV)
        ops._lib.check(L.gw_pack_linear_bf16x3(wd.data_ptr(), n, k, 0, k, buf.data_ptr(), st), "pack")
        return buf
    buf = torch.empty(L.gw_packed_bytes_bf16(n, 0, k) // 2, dtype=torch.bfloat16, device=DEV)
    ops._lib.check(L.gw_pack_linear_bf16(wd.data_ptr(), n, k, 0, k, buf.data_ptr(), st), "pack")
    return buf


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("B,n_src,n_dst,E,use_e", [(2, 50, 40, 333, True), (1, 7, 5, 64, True), (3, 9, 4, 5, False), (16, 20, 30, 1000, True),
                                                   (4, 33, 17, 129, True)])
def test_team_edge_update_without_residual_stays_inside_its_buffers(half, B, n_src, n_dst, E, use_e):
    """Round-3 form of the decoder / encoder edge update (csrc/gw_edge16t.hip, gathered, e_res.k == 0): one per-sample table of
    projected rows (fp32 or fp16), the batch-shared per-edge products, nothing written but the aggregate."""
    rs = np.random.RandomState(E + 3)
    mlp = _mlp(rs, 768, 256, 256, True, torch.bfloat16)
    src, dst = _graph(rs, n_src, n_dst, E)
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))
    pe = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
        agg = g.wrap(torch.zeros(B * n_dst, 256))
        tab = g.wrap(ps.half() if half else ps)
        ops.edge_update_forward(pm, B, g.wrap(src), g.wrap(dst), Operand(tab, n_src, 256, projected=True), ops.ZERO,
                                Operand(g.wrap(pe), 0, 256, projected=True) if use_e else ops.ZERO, ops.ZERO, n_dst, agg, None)
        return {"agg": agg}

    _both(run)


@pytest.mark.parametrize("half,dtype", [(False, torch.bfloat16), (True, torch.bfloat16), (False, X3)])
@pytest.mark.parametrize("rows,B,in_dim,want_out", [(1000, 2, 102, False), (77, 1, 78, True), (333, 3, 128, False), (1, 1, 40, True)])
def test_mlp_with_post_products_stays_inside_its_buffers(half, dtype, rows, B, in_dim, want_out):
    """gw_mlp_post_forward (node encoder + the first layer-1 product of the encoder's edge MLP, one launch), bf16 / bf16x3."""
    rs = np.random.RandomState(rows + in_dim)
    mlp = _mlp(rs, in_dim, 256, 256, True, dtype)
    post = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32))
    n = rows * B
    x = torch.from_numpy(rs.standard_normal((n, in_dim)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, in_dim),), dtype)
        pw = g.wrap(_pack_bf16(post, dtype))
        out = g.wrap(torch.zeros(n, 256)) if want_out else None
        po = g.wrap(torch.zeros(n, 256, dtype=torch.float16 if half else torch.float32))
        ops.mlp_post_forward(pm, Operand(g.wrap(x), rows, in_dim), n, rows, [pw], post_half=half, out=out, post_out=[po])
        res = {"post": po}
        if want_out:
            res["out"] = out
        return res

    _both(run)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3])
@pytest.mark.parametrize("rows,B,n_out,with_res", [(1000, 2, 78, True), (77, 1, 78, False), (333, 3, 37, True), (1, 1, 80, True)])
def test_node_update_with_head_stays_inside_its_buffers(dtype, rows, B, n_out, with_res):
    """gw_node_update_head_forward (decoder node update + output head + residual, one launch), bf16; the x operand is the projected
    batch-shared table the round-3 decoder feeds (rows_per_batch 0)."""
    rs = np.random.RandomState(rows + n_out)
    mlp = _mlp(rs, 512, 256, 256, True, dtype)
    head = _mlp(rs, 256, 128, n_out, False, dtype)
    n = rows * B
    xp = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32))
    a = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32))
    feats = torch.from_numpy(rs.standard_normal((n, 102)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512)), dtype)
        hd = _packed(g, head, ((0, 256),), dtype)
        out = g.wrap(torch.zeros(n, n_out))
        ops.node_update_head_forward(pm, hd, n, rows, Operand(g.wrap(xp), 0, 256, projected=True), Operand(g.wrap(a), rows, 256),
                                     Operand(g.wrap(feats), rows, n_out) if with_res else None, out=out)
        return {"out": out}

    _both(run)


def test_node_update_with_head_refuses_a_head_packed_for_another_width():
    """gw_node_update_head_forward streams 8 K-steps of the head's first Linear and 80 rows of its last one: a head packed for another
    input width (gw_mlp_weights.k_in) or with fewer output rows (out_rows) is refused instead of being read past its end."""
    rs = np.random.RandomState(9)
    mlp = _mlp(rs, 512, 256, 256, True, torch.bfloat16)
    n = 50
    xp = torch.zeros(n, 256, device=DEV)
    a = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    g = Guards(False)
    pm = _packed(g, mlp, ((0, 256), (256, 512)), torch.bfloat16)
    narrow = _packed(g, _mlp(rs, 128, 128, 78, False, torch.bfloat16), ((0, 128),), torch.bfloat16)  # a 128-wide head
    with pytest.raises(RuntimeError, match="k_in"):
        ops.node_update_head_forward(pm, narrow, n, n, Operand(xp, 0, 256, projected=True), Operand(a, n, 256), None)
    ok = _packed(g, _mlp(rs, 256, 128, 78, False, torch.bfloat16), ((0, 256),), torch.bfloat16)
    ok.out_rows = 78  # as if the caller had packed the last Linear with its natural 78 rows
    with pytest.raises(RuntimeError, match="out_rows"):
        ops.node_update_head_forward(pm, ok, n, n, Operand(xp, 0, 256, projected=True), Operand(a, n, 256), None)


@pytest.mark.parametrize("rows,n_chain,n_fan", [(1000, 2, 3), (77, 1, 0), (1, 2, 1), (4097, 2, 2)])
def test_chain_backward_stays_inside_its_buffers(rows, n_chain, n_fan):
    """gw_mlp_chain_backward: gradient rows, ReLU masks, transposed packs and every output between canaries."""
    from graph_weather_amd import autograd as ag

    rs = np.random.RandomState(rows + n_fan)
    L = ops._lib.lib()
    Ws = [torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32)) for _ in range(n_chain + n_fan)]
    d = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32))
    hs = [torch.from_numpy(np.maximum(rs.standard_normal((rows, 256)), 0).astype(np.float32)) for _ in range(n_chain)]

    def run(g):
        n = int(L.gw_packed_floats(256, 0, 256))
        pk = []
        for w in Ws:
            wd = w.to(DEV)
            buf = torch.empty(n, device=DEV)
            ops.pack_many(ops._lib.DTYPE_F32, [(wd.data_ptr(), 1, 256, 256, 256, buf.data_ptr())], [], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            pk.append(g.wrap(buf))
        outs = [g.wrap(torch.zeros(rows, 256)) for _ in range(n_chain + n_fan)]
        ag.chain_backward(g.wrap(d), [(pk[i], g.wrap(hs[i]), outs[i]) for i in range(n_chain)],
                          [(pk[n_chain + s], outs[n_chain + s]) for s in range(n_fan)])
        return {f"o{i}": o for i, o in enumerate(outs)}

    _both(run)


def _seg_graph(rs, n_src, n_dst, degrees):
    from graph_weather_amd.graphs import plan_from_coo

    deg = rs.choice(degrees, size=n_dst)
    deg[-1] = max(int(deg[-1]), 1)  # the last destination row is written
    dst = np.repeat(np.arange(n_dst), deg)
    src = rs.randint(0, n_src, size=dst.size)
    src[0] = n_src - 1           # ... and the last source row is read
    return plan_from_coo(src, dst, n_src, n_dst)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("B,n_src,n_dst,degrees,kind", [(2, 50, 41, [7, 7, 6], "bf16k"), (1, 7, 3, [2], "fp32"), (3, 19, 77, [1, 2, 3], "bf16k"),
                                                        (16, 20, 25, [0, 5, 40, 64], "fp32"), (2, 33, 9, [3, 70, 150, 64], "split")])
def test_segment_tile_edge_update_stays_inside_its_buffers(half, B, n_src, n_dst, degrees, kind):
    """Round-4 gather form on segment-aligned tiles (GW_EDGE_SEGMENT_TILES; aggregate written as fp32 rows, as bf16 rows in K
    order, or - runs longer than a tile - added by atomics): padded index arrays, padded per-edge rows, a destination table whose
    rows without edges are never touched."""
    rs = np.random.RandomState(n_dst + 5)
    mlp = _mlp(rs, 768, 256, 256, True, torch.bfloat16)
    plan = _seg_graph(rs, n_src, n_dst, degrees)
    seg = plan.seg_tiles(split=(kind == "split"))
    assert seg is not None and seg.split == (kind == "split")
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))
    pe = seg.pad_rows(torch.from_numpy(rs.standard_normal((plan.num_edges, 256)).astype(np.float32)))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
        agg = g.wrap(torch.zeros(B * n_dst, 256, dtype=torch.bfloat16 if kind == "bf16k" else torch.float32))
        tab = g.wrap(ps.half() if half else ps)
        ops.edge_update_forward(pm, B, g.wrap(seg.src), g.wrap(seg.dst), Operand(tab, n_src, 256, projected=True), ops.ZERO,
                                Operand(g.wrap(pe), 0, 256, projected=True), ops.ZERO, n_dst, agg, None, segment_tiles=True,
                                segment_split=seg.split)
        return {"agg": agg}

    _both(run)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("B,n,degrees,want_e", [(2, 41, [7, 7, 6], True), (1, 3, [2], False), (3, 77, [4, 5, 6, 7], True), (16, 25, [7], False)])
def test_processor_form_on_segment_tiles_stays_inside_its_buffers(half, B, n, degrees, want_e):
    """Round-4 processor-block form (edge16_l1_kernel + csrc/gw_edge16p.hip): padded per-sample edge tiles as operand and
    residual, both node products gathered, e' tiles written or dropped, the aggregate accumulated in place."""
    from .helpers import edge_tiles_from_rows

    rs = np.random.RandomState(n + 11)
    mlp = _mlp(rs, 768, 256, 256, True, torch.bfloat16)
    plan = _seg_graph(rs, n, n, degrees)
    seg = plan.seg_tiles()
    assert seg is not None and seg.max_slots <= 16
    E, P = plan.num_edges, seg.n_pad
    ps = torch.from_numpy(rs.standard_normal((B * n, 256)).astype(np.float32))
    pd = torch.from_numpy(rs.standard_normal((B * n, 256)).astype(np.float32))
    e = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32))
    e_pad = seg.pad_batched_rows(e, B).reshape(B, P, 256)
    tiles_host = edge_tiles_from_rows(e_pad).view(torch.uint8).reshape(-1)
    agg0 = torch.from_numpy(rs.standard_normal((B * n, 256)).astype(np.float32))

    def run(g):
        pm = _packed(g, mlp, ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
        tiles = g.wrap(tiles_host)
        agg = g.wrap(agg0)
        e_out = g.wrap(torch.zeros(ops.edge_tiles_bytes(B, P), dtype=torch.uint8)) if want_e else None
        et = Operand(tiles, P, 256, tiles=True)
        ops.edge_update_forward(pm, B, g.wrap(seg.src), g.wrap(seg.dst), Operand(g.wrap(ps.half() if half else ps), n, 256, projected=True),
                                Operand(g.wrap(pd.half() if half else pd), n, 256, projected=True), et, et, n, agg, e_out,
                                segment_tiles=True)
        out = {"agg": agg}
        if e_out is not None:
            out["e_out"] = e_out.view(torch.bfloat16)
        return out

    _both(run)
