"""Host logic of the MLP backward (graph_weather_amd/autograd.py: _mlp_chain_backward and the fused chain launch's extras) on the
CPU: every C-ABI call it makes is replaced by a torch restatement of what that entry point computes (include/gw_amd.h:
gw_mlp_ln_chain_backward, gw_gemm_f32 TN, gw_layernorm_backward, gw_gather_rows, gw_relu_backward), and the parameter / input
gradients it assembles are compared with torch's fp64 autograd of the reference MLP (graph_net_block.py:45-77) for every route
the function can take: fused launch with the LayerNorm prologue, with a gathered input gradient read in place, without
LayerNorm, the unfused fallback; Linear_0's bias gradient from the caller's GEMM, from the chain launch, from the fallback pass;
a joining gradient as rows and as the launch's own input."""
import types

import pytest
import torch

from graph_weather_amd import autograd as ag

R, B, NDST = 37, 2, 11  # edge rows per batch element, batch, destination rows


def _ln_bwd64(dn, y, gamma):
    yd = y.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros_like(gd).requires_grad_(True)
    torch.nn.functional.layer_norm(yd, (yd.shape[1],), gd, bd, 1e-5).backward(dn.double())
    return yd.grad, gd.grad, bd.grad


class _Calls:
    def __init__(self):
        self.names = []


@pytest.fixture
def restated(monkeypatch):
    calls = _Calls()

    def packed_t(mlp, layer, W, lo, hi):
        return None if getattr(mlp, "no_packs", False) else W[:, lo:hi].detach().clone()  # "stream" of the block: d @ pt

    def chain_backward(d, chain, fan, ln=None, colsum=None, fan_add=None, gather=None):
        calls.names.append("chain" + ("+ln" if ln is not None else "") + ("+gather" if gather is not None else "")
                           + ("+colsum" if colsum is not None else ""))
        rows = d
        if gather is not None:
            idx, tab_rows, add = gather
            n = int(idx.numel())
            b = torch.arange(int(chain[0][2].shape[0]) // n).repeat_interleave(n)
            rows = d[b * tab_rows + idx.long().repeat(len(b) // n)]
            if add is not None:
                rows = rows + add
        cur = rows.double()
        if ln is not None:
            y, gamma, dgamma, dbeta, dy = ln
            gy, gg, gb = _ln_bwd64(rows, y, gamma)
            dgamma += gg.float()
            dbeta += gb.float()
            dy.copy_(gy.float())
            cur = gy
        for pt, mask, out in chain:
            cur = (cur @ pt.double()) * (mask > 0)
            out.copy_(cur.float())
        if colsum is not None:
            colsum += cur.sum(0).float()
        for i, (pt, out) in enumerate(fan):
            v = cur @ pt.double()
            extra = fan_add[i] if fan_add else None
            if extra is ag.FAN_ADD_INPUT:
                v = v + rows.double()
            elif extra is not None:
                v = v + extra.double()
            out.copy_(v.float())

    def gemm_tn_acc(a, b, c, c_col0=0, colsum=None, x3=False):
        calls.names.append("gemm" + ("+colsum" if colsum is not None else ""))
        c[:, c_col0:c_col0 + b.shape[1]] += (a.double().t() @ b.double()).float()
        if colsum is not None:
            colsum += a.double().sum(0).float()

    def layernorm_backward(dn, y, gamma, dgamma, dbeta, width=0):
        calls.names.append("ln_bwd")
        gy, gg, gb = _ln_bwd64(dn, y, gamma)
        dgamma += gg.float()
        dbeta += gb.float()
        return gy.float()

    def relu_backward(dh, h, db):
        calls.names.append("relu_bwd")
        if h is not None:
            dh.mul_((h > 0).float())
        if db is not None:
            db += dh.double().sum(0).float()
        return dh

    def input_grad(mlp, layer, d, W, lo, hi, relu_of=None):
        calls.names.append("single")
        out = (d.double() @ W[:, lo:hi].double()).float()
        return out * (relu_of > 0) if relu_of is not None else out

    def gather_rows(table, rows_pb, idx, batch, n_idx, add=None):
        calls.names.append("gather")
        b = torch.arange(batch).repeat_interleave(n_idx)
        out = table[b * rows_pb + idx.long().repeat(batch)]
        return out + add if add is not None else out.clone()

    for name, fn in (("_packed_transposed", packed_t), ("chain_backward", chain_backward), ("gemm_tn_acc", gemm_tn_acc),
                     ("layernorm_backward", layernorm_backward), ("relu_backward", relu_backward), ("input_grad", input_grad),
                     ("gather_rows", gather_rows)):
        monkeypatch.setattr(ag, name, fn)
    return calls


def _case(norm: bool, seed: int):
    g = torch.Generator().manual_seed(seed)
    rows = B * R
    p = {"W0": torch.randn(256, 512, generator=g) / 20, "b0": 0.1 * torch.randn(256, generator=g),
         "W1": torch.randn(256, 256, generator=g) / 16, "b1": 0.1 * torch.randn(256, generator=g),
         "W2": torch.randn(256, 256, generator=g) / 16, "b2": 0.1 * torch.randn(256, generator=g)}
    if norm:
        p["gamma"], p["beta"] = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    xa, xb = torch.randn(rows, 256, generator=g), torch.randn(rows, 256, generator=g)
    dagg = torch.randn(B * NDST, 256, generator=g)
    dst = torch.sort(torch.randint(0, NDST, (R,), generator=g)).values.to(torch.int32)
    de_out = torch.randn(rows, 256, generator=g)
    # the reference in fp64
    q = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xad, xbd = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    h0 = torch.relu(torch.cat([xad, xbd], 1) @ q["W0"].t() + q["b0"])
    h1 = torch.relu(h0 @ q["W1"].t() + q["b1"])
    y = h1 @ q["W2"].t() + q["b2"]
    out = torch.nn.functional.layer_norm(y, (256,), q["gamma"], q["beta"], 1e-5) if norm else y
    bidx = torch.arange(B).repeat_interleave(R)
    dn = dagg[bidx * NDST + dst.long().repeat(B)] + de_out
    out.backward(dn.double())
    saved = types.SimpleNamespace(hidden=[h0.detach().float(), h1.detach().float()], pre_norm=y.detach().float())
    weights = [p[k] for k in ("W0", "b0", "W1", "b1", "W2", "b2")] + ([p["gamma"], p["beta"]] if norm else [])
    want = {k: v.grad for k, v in q.items()}
    want["xa"], want["xb"] = xad.grad, xbd.grad
    return weights, saved, (xa, xb), (dagg, dst, de_out, dn), want


def _close(a, b, name):
    err = (a.double() - b).abs().max().item() / (b.abs().max().item() + 1e-30)
    assert err < 2e-5, (name, err)


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("route", ["fused", "fused_gathered", "fallback", "fallback_gathered"])
@pytest.mark.parametrize("bias0", ["caller", "in_chain", "here"])
@pytest.mark.parametrize("join", [None, "rows", "input"])
def test_mlp_backward_routes_give_the_reference_gradients(restated, norm, route, bias0, join):
    weights, saved, (xa, xb), (dagg, dst, de_out, dn), want = _case(norm, seed=5)
    mlp = types.SimpleNamespace(compute_dtype=torch.float32, no_packs=route.startswith("fallback"))
    gathered = route.endswith("gathered")
    dout = ag.GatheredRows(dagg, NDST, dst, B, de_out) if gathered else dn.clone()
    grads = [None] * len(weights)
    fo: dict = {}
    fan = [(0, 256), (256, 512)]
    joining = torch.randn(B * R, 256, generator=torch.Generator().manual_seed(9))
    fan_add = None if join is None else {(256, 512): joining if join == "rows" else ag.FAN_ADD_INPUT}
    dz0, _ = ag._mlp_chain_backward(dout, saved, weights, norm, weights[-2] if norm else None, grads, mlp, 256, fan=fan, fan_out=fo,
                                    bias0_by_caller=bias0 != "here", bias0_in_chain=bias0 == "in_chain", fan_add=fan_add)
    # the caller's part: Linear_0's weight gradient from its operands, the bias gradient where it is still pending
    gW0 = grads[0]
    pending = None if (bias0 == "here" or fo.get("bias0_done")) else grads[1]
    if bias0 == "in_chain":
        if pending is not None:
            ag.relu_backward(dz0, None, pending)  # (EdgeUpdateFunction's fallback when the launch did not sum the columns)
        ag.gemm_tn_acc(dz0, xa, gW0, c_col0=0)
    else:
        ag.gemm_tn_acc(dz0, xa, gW0, c_col0=0, colsum=pending)
    ag.gemm_tn_acc(dz0, xb, gW0, c_col0=256)
    names = ["W0", "b0", "W1", "b1", "W2", "b2"] + (["gamma", "beta"] if norm else [])
    for n_, g_ in zip(names, grads):
        _close(g_, want[n_], n_)
    _close(fo[(0, 256)], want["xa"], "d xa")
    added = (256, 512) in fo.get("added", ())
    base = fo[(256, 512)].double()
    if join is not None and not added:  # the launch did not take it: the caller adds (as EdgeUpdateFunction returns de_res)
        base = base + (joining.double() if join == "rows" else dn.double())
    ref = want["xb"] + (0 if join is None else (joining.double() if join == "rows" else dn.double()))
    _close(base, ref, "d xb (+ joined)")
    # route bookkeeping: what ran
    fused = not route.startswith("fallback")
    assert any(c.startswith("chain") for c in restated.names) == fused
    if fused and norm:
        assert "ln_bwd" not in restated.names and any("+ln" in c for c in restated.names)
    if gathered:
        in_place = fused and norm
        assert (dout._rows is None) == in_place and any("+gather" in c for c in restated.names) == in_place
        assert ("gather" in restated.names) == (not in_place)
    if join is not None:
        assert added == fused
    assert bool(fo.get("bias0_done")) == (fused and bias0 == "in_chain")
