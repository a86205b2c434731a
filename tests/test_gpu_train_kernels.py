"""Building blocks of the backward pass (csrc/gw_train.hip) against plain torch fp64 references on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from graph_weather_amd import _lib  # noqa: E402

DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rel(a, ref):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)


@pytest.mark.parametrize("m,n,k", [(1000, 256, 256), (77, 102, 256), (5, 2, 256), (300, 78, 128), (64, 64, 4), (129, 130, 7)])
def test_gemm_nn(m, n, k):
    rs = np.random.RandomState(m + n)
    a = torch.from_numpy(rs.standard_normal((m, k)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((k, n + 3)).astype(np.float32))  # ldb > n
    c = torch.full((m, n + 5), 7.0, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    L = _lib.lib()
    _lib.check(L.gw_gemm_f32(_lib.GEMM_NN, m, n, k, ad.data_ptr(), k, bd.data_ptr(), n + 3, c.data_ptr(), n + 5, None, _st()), "gemm")
    ref = a.double() @ b[:, :n].double()
    assert _rel(c[:, :n], ref) < 2e-6
    assert torch.all(c[:, n:] == 7.0)  # nothing written outside


@pytest.mark.parametrize("m,n,k", [(256, 256, 5000), (256, 102, 4097), (78, 128, 300), (256, 2, 10000), (16, 16, 1)])
def test_gemm_tn_accumulates(m, n, k):
    rs = np.random.RandomState(k)
    a = torch.from_numpy(rs.standard_normal((k, m)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((k, n)).astype(np.float32))
    c0 = torch.from_numpy(rs.standard_normal((m, n)).astype(np.float32))
    c = c0.to(DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    L = _lib.lib()
    cs = torch.ones(m, device=DEV)
    _lib.check(L.gw_gemm_f32(_lib.GEMM_TN, m, n, k, ad.data_ptr(), m, bd.data_ptr(), n, c.data_ptr(), n, cs.data_ptr(), _st()), "gemm")
    ref = c0.double() + a.double().t() @ b.double()
    assert _rel(c, ref) < 1e-5
    assert _rel(cs, 1.0 + a.double().sum(0)) < 1e-5  # fused column sums of A (bias gradient)


@pytest.mark.parametrize("m,n,k,ld_extra", [(256, 256, 5000, 0), (256, 128, 4097, 4), (128, 256, 31, 0), (256, 256, 100000, 0),
                                            (256, 102, 4097, 0),
                                            # one, two, three stages of 32 rows and their ragged neighbours (the loads run two stages ahead)
                                            (128, 128, 1, 0), (256, 256, 32, 0), (256, 256, 33, 0), (128, 256, 64, 0), (256, 128, 65, 0),
                                            (256, 256, 96, 0), (256, 256, 127, 0), (256, 256, 256 + 95, 0), (256, 256, 300007, 0),
                                            # ragged m / n, odd leading dimensions
                                            (78, 128, 5000, 0), (80, 128, 333, 1), (256, 78, 4097, 3), (1, 1, 70, 0), (129, 257, 1000, 1),
                                            (102, 256, 129600, 0)])
def test_gemm_tn_on_split_operands(m, n, k, ld_extra):
    """GW_GEMM_TN_BF16X3 (weight gradients of the mixed-precision training step): the fp32 TN sums to ~1e-5 on operands of mixed
    sign and magnitude (a transposed operand or a k permutation that differs between A and B would be an O(1) error); ragged k,
    strided operands (any leading dimension: the loads are 4-byte), a 768-wide destination written at a column offset; shapes
    that are not multiples of 128 (the 102 input / 78 output features of the path) run the ragged form of the same kernel."""
    rs = np.random.RandomState(k + n)
    a = torch.from_numpy((rs.standard_normal((k, m + ld_extra)) * 10.0 ** rs.uniform(-2, 2, size=(k, 1))).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((k, n + ld_extra)).astype(np.float32))
    c0 = torch.from_numpy(rs.standard_normal((m, 768)).astype(np.float32))
    c = c0.to(DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    L = _lib.lib()
    cs = torch.ones(m, device=DEV)
    col0 = 256
    _lib.check(L.gw_gemm_f32(_lib.GEMM_TN_BF16X3, m, n, k, ad.data_ptr(), m + ld_extra, bd.data_ptr(), n + ld_extra,
                             c.data_ptr() + 4 * col0, 768, cs.data_ptr(), _st()), "gemm x3")
    prod = a[:, :m].double().t() @ b[:, :n].double()
    ref = c0.double().clone()
    ref[:, col0:col0 + n] += prod
    err = (c.cpu().double() - ref).abs().max().item() / prod.abs().max().item()
    print(f"[gemm tn x3] m={m} n={n} k={k}: max err / max |product sum| = {err:.2e}")
    assert err < 5e-5
    assert torch.equal(c.cpu()[:, :col0], c0[:, :col0]) and torch.equal(c.cpu()[:, col0 + n:], c0[:, col0 + n:])
    assert _rel(cs, 1.0 + a[:, :m].double().sum(0)) < 1e-5


def test_relu_backward_and_bias_grad():
    rs = np.random.RandomState(1)
    rows, w = 1000, 200
    dh = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32))
    h = torch.relu(torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float32)))
    dz = torch.zeros(rows, 256, device=DEV)
    db = torch.ones(256, device=DEV)
    L = _lib.lib()
    dhd, hd = dh.to(DEV), h.to(DEV)
    _lib.check(L.gw_relu_backward(rows, w, dhd.data_ptr(), 256, hd.data_ptr(), 256, dz.data_ptr(), 256, db.data_ptr(), _st()), "relu")
    ref = (dh * (h > 0))[:, :w]
    assert torch.equal(dz[:, :w].cpu(), ref)
    assert _rel(db[:w], 1.0 + ref.double().sum(0)) < 1e-5
    assert torch.all(db[w:] == 1.0) and torch.all(dz[:, w:] == 0)
    # no mask: plain column sums (bias gradient of the last Linear)
    db2 = torch.zeros(256, device=DEV)
    _lib.check(L.gw_relu_backward(rows, 256, dhd.data_ptr(), 256, None, 0, None, 0, db2.data_ptr(), _st()), "colsum")
    assert _rel(db2, dh.double().sum(0)) < 1e-5


def test_layernorm_backward():
    rs = np.random.RandomState(2)
    rows = 777
    y = torch.from_numpy(rs.standard_normal((rows, 256)).astype(np.float64) * 2 + 0.3).requires_grad_(True)
    gamma = torch.from_numpy(1 + 0.1 * rs.standard_normal(256)).requires_grad_(True)
    beta = torch.zeros(256, dtype=torch.float64, requires_grad=True)
    dn = torch.from_numpy(rs.standard_normal((rows, 256)))
    torch.nn.functional.layer_norm(y, (256,), gamma, beta, 1e-5).backward(dn)
    L = _lib.lib()
    yd, gd, dnd = y.detach().float().to(DEV), gamma.detach().float().to(DEV), dn.float().to(DEV)
    dy = torch.empty(rows, 256, device=DEV)
    dg = torch.zeros(256, device=DEV)
    dbt = torch.zeros(256, device=DEV)
    _lib.check(L.gw_layernorm_backward(rows, 256, dnd.data_ptr(), 256, yd.data_ptr(), 256, gd.data_ptr(), dy.data_ptr(), 256,
                                       dg.data_ptr(), dbt.data_ptr(), _st()), "ln")
    assert _rel(dy, y.grad) < 1e-5
    assert _rel(dg, gamma.grad) < 1e-5
    assert _rel(dbt, beta.grad) < 1e-5


def test_gather_and_segment_sum_are_duals():
    rs = np.random.RandomState(3)
    B, N, E = 2, 50, 400
    idx = np.sort(rs.randint(0, N, size=E)).astype(np.int32)  # destination-sorted like the plans
    idx[:60] = 7  # one long segment, some empty ones
    idx = np.sort(idx)
    table = torch.from_numpy(rs.standard_normal((B * N, 256)).astype(np.float32))
    add = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32))
    L = _lib.lib()
    out = torch.empty(B * E, 256, device=DEV)
    td, ad, idd = table.to(DEV), add.to(DEV), torch.from_numpy(idx).to(DEV)
    _lib.check(L.gw_gather_rows(B, E, td.data_ptr(), N, idd.data_ptr(), ad.data_ptr(), out.data_ptr(), _st()), "gather")
    ref = table.reshape(B, N, 256)[:, idx.astype(np.int64)].reshape(B * E, 256) + add
    assert torch.equal(out.cpu(), ref)
    # shared table (rows_per_batch = 0), no add
    _lib.check(L.gw_gather_rows(B, E, td.data_ptr(), 0, idd.data_ptr(), None, out.data_ptr(), _st()), "gather shared")
    assert torch.equal(out.cpu(), table[:N][idx.astype(np.int64)].repeat(B, 1))
    # segment sum through a permutation: rows in arbitrary order, perm sorts them by segment
    order = rs.permutation(E).astype(np.int32)           # rows[order[i]] is the i-th row in segment order
    seg_of_sorted = idx.astype(np.int64)
    ptr = np.zeros(N + 1, dtype=np.int32)
    np.add.at(ptr, seg_of_sorted + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    rows = torch.from_numpy(rs.standard_normal((B * E, 256)).astype(np.float32))
    rd, pd, od = rows.to(DEV), torch.from_numpy(ptr).to(DEV), torch.from_numpy(order).to(DEV)
    o = torch.full((B * N, 256), 3.0, device=DEV)
    _lib.check(L.gw_segment_sum_rows(B, B, N, rd.data_ptr(), E, od.data_ptr(), pd.data_ptr(), o.data_ptr(), 1, _st()), "segsum")
    ref = torch.full((B, N, 256), 3.0, dtype=torch.float64)
    r3 = rows.reshape(B, E, 256).double()
    for i in range(E):
        ref[:, seg_of_sorted[i]] += r3[:, order[i]]
    assert _rel(o, ref.reshape(B * N, 256)) < 1e-5
    # summed over the batch as well, overwrite mode, identity permutation
    o1 = torch.full((N, 256), 9.0, device=DEV)
    _lib.check(L.gw_segment_sum_rows(B, 1, N, rd.data_ptr(), E, None, pd.data_ptr(), o1.data_ptr(), 0, _st()), "segsum shared")
    ref1 = torch.zeros(N, 256, dtype=torch.float64)
    for i in range(E):
        ref1[seg_of_sorted[i]] += r3[:, i].sum(0)
    assert _rel(o1, ref1) < 1e-5


def test_adamw_matches_torch():
    rs = np.random.RandomState(4)
    p0 = torch.from_numpy(rs.standard_normal(10000).astype(np.float32))
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.to(DEV)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    L = _lib.lib()
    for step in range(1, 4):
        g = torch.from_numpy(rs.standard_normal(10000).astype(np.float32))
        ref.grad = g.clone()
        opt.step()
        gd = g.to(DEV)
        _lib.check(L.gw_adamw_step(p.numel(), p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-3, 0.9, 0.999, 1e-8,
                                   0.01, step, _st()), "adamw")
    assert _rel(p, ref.detach()) < 1e-6


def test_nmse_backward_matches_autograd():
    from oracle import reference_math as om

    lat_lons = [(lat, lon) for lat in range(-90, 90, 10) for lon in range(0, 360, 10)]
    rs = np.random.RandomState(5)
    pred = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78))).requires_grad_(True)
    target = torch.from_numpy(rs.standard_normal((2, len(lat_lons), 78)))
    var = torch.from_numpy(rs.rand(78) + 0.5)
    loss = om.normalized_mse_loss(pred, target, lat_lons, var, normalize=True)
    loss.backward()
    lats = sorted(set(lat for lat, _ in lat_lons))
    w = torch.tensor([np.cos(lat * np.pi / 180.0) for lat in lats], dtype=torch.float32, device=DEV)
    L = _lib.lib()
    pd, td = pred.detach().float().to(DEV), target.float().to(DEV)
    iv = (1.0 / var).float().to(DEV)
    dl = torch.full((1,), 1.0, device=DEV)
    dp = torch.empty_like(pd)
    _lib.check(L.gw_normalized_mse_backward(pd.data_ptr(), td.data_ptr(), iv.data_ptr(), 0, w.data_ptr(), len(lats), 2, len(lat_lons), 78,
                                            dl.data_ptr(), dp.data_ptr(), _st()), "nmse bwd")
    assert _rel(dp, pred.grad) < 1e-5
