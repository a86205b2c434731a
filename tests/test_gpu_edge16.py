"""The bf16 edge update with register-resident weights (csrc/gw_edge16.hip) against a float64 emulation of exactly its
arithmetic: bf16-rounded weights and activations (round to nearest even), exact products, fp32-style sums.  Layout or
indexing mistakes show up as O(1) errors; what remains is summation order (~1e-6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from graph_weather_amd import ops  # noqa: E402
from graph_weather_amd.ops import Operand, PackedMLP  # noqa: E402

DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("case", ["decoder", "encoder", "tiny"])
def test_edge16_matches_bf16_emulation(case):
    rs = np.random.RandomState({"decoder": 1, "encoder": 2, "tiny": 3}[case])
    B, n_src, n_dst, E = (3, 50, 40, 333) if case != "tiny" else (2, 4, 3, 5)
    W0 = torch.from_numpy((rs.standard_normal((256, 768)) / 16).astype(np.float32))
    W1 = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32))
    W2 = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32))
    b0, b1, b2 = (torch.from_numpy((0.1 * rs.standard_normal(256)).astype(np.float32)) for _ in range(3))
    gamma = torch.from_numpy((1 + 0.1 * rs.standard_normal(256)).astype(np.float32))
    beta = torch.from_numpy((0.1 * rs.standard_normal(256)).astype(np.float32))
    dst = np.sort(np.where(rs.rand(E) < 0.4, n_dst // 2, rs.randint(0, n_dst, size=E)))  # one long segment spanning tiles
    src = rs.randint(0, n_src, size=E)
    ps = torch.from_numpy(rs.standard_normal((B * n_src, 256)).astype(np.float32))   # per-sample projected source rows
    pd = torch.from_numpy(rs.standard_normal((n_dst, 256)).astype(np.float32))       # batch-shared projected destination rows
    pe = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))           # batch-shared projected edge rows
    e_res = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))
    use_dst = case != "decoder"

    # ---- emulation ----
    z1 = b0.double() + ps.double().reshape(B, n_src, 256)[:, src] + pe.double()[None]
    if use_dst:
        z1 = z1 + pd.double()[dst][None]
    h1 = _bf(torch.relu(z1).float())
    h2 = _bf(torch.relu(h1 @ _bf(W1).t() + b1.double()).float())
    o = h2 @ _bf(W2).t() + b2.double()
    y = torch.nn.functional.layer_norm(o, (256,), gamma.double(), beta.double(), 1e-5) + e_res.double()[None]
    agg_ref = torch.zeros(B, n_dst, 256, dtype=torch.float64)
    agg_ref.index_add_(1, torch.from_numpy(dst), y)

    # ---- kernel ----
    pm = PackedMLP([W0.to(DEV), W1.to(DEV), W2.to(DEV)], [b0.to(DEV), b1.to(DEV), b2.to(DEV)], (gamma.to(DEV), beta.to(DEV)),
                   ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
    agg = torch.zeros((B * n_dst, 256), device=DEV)
    e_out = torch.empty((B * E, 256), device=DEV) if case != "decoder" else None
    x_dst = Operand(pd.to(DEV), 0, 256, projected=True) if use_dst else ops.ZERO
    ops.edge_update_forward(pm, B, torch.from_numpy(src.astype(np.int32)).to(DEV), torch.from_numpy(dst.astype(np.int32)).to(DEV),
                            Operand(ps.to(DEV), n_src, 256, projected=True), x_dst, Operand(pe.to(DEV), 0, 256, projected=True),
                            Operand(e_res.to(DEV), 0, 256), n_dst, agg, e_out)
    torch.cuda.synchronize()
    scale = agg_ref.abs().max().item()
    err = (agg.cpu().double().reshape(B, n_dst, 256) - agg_ref).abs().max().item()
    assert err < 3e-4 * scale, f"{case}: aggregate max err {err:.3e} vs scale {scale:.3e}"
    if e_out is not None:
        err_e = (e_out.cpu().double().reshape(B, E, 256) - y).abs().max().item()
        # a bf16 rounding of one activation can fall the other way (fp32 sums here, float64 in the emulation): ~1e-3
        assert err_e < 1.5e-3 * y.abs().max().item(), f"{case}: e' max err {err_e:.3e}"


@pytest.mark.parametrize("use_dst,res_tiles", [(False, False), (True, False), (True, True)])
def test_edge16_against_the_oracle(use_dst, res_tiles):
    """The same launches against the ORACLE (oracle/reference_math.py: EdgeProcessor.forward + scatter_sum in full precision
    from the raw node / edge rows), not against a model of the kernel: the layer-1 products the kernel gathers are made from
    the raw rows here, so split, gather, both resident layers, LayerNorm, residual and segment sums are all inside the
    comparison.  Error budget of bf16 operands: 2e-2 of the output scale."""
    from graph_weather_amd.utils import deterministic_fill_
    from oracle import reference_math as om
    import graph_weather_amd as gw

    rs = np.random.RandomState(11 + int(use_dst))
    B, n_src, n_dst, E = 3, 60, 45, 700
    ep = gw.EdgeProcessor(256, 256, 256, 2, "LayerNorm")
    deterministic_fill_(ep, seed=17)
    p = {"blk.edge_model." + k: v.clone() for k, v in ep.state_dict().items()}
    x_src = torch.from_numpy(rs.standard_normal((B, n_src, 256)).astype(np.float32))
    x_dst = torch.from_numpy(rs.standard_normal((n_dst, 256)).astype(np.float32)) if use_dst else torch.zeros(n_dst, 256)
    e = torch.from_numpy(rs.standard_normal((E, 256)).astype(np.float32))  # batch-shared edge features (encoder / decoder case)
    dst = np.sort(np.where(rs.rand(E) < 0.35, n_dst // 3, rs.randint(0, n_dst, size=E)))
    src = rs.randint(0, n_src, size=E)
    st, dt = torch.from_numpy(src), torch.from_numpy(dst)
    # oracle, per sample
    e_ref, agg_ref = [], []
    for b in range(B):
        en = om.edge_processor(p, "blk.edge_model", x_src[b][st], x_dst[dt], e)
        e_ref.append(en)
        agg_ref.append(om.scatter_sum(en, dt, n_dst))
    e_ref, agg_ref = torch.stack(e_ref), torch.stack(agg_ref)
    # kernel: layer-1 products from the raw rows (fp64 product, fp32 table - what gw_project_forward hands over in fp32 mode)
    W0 = ep.edge_mlp.model[0].weight.detach().double()
    ps = (x_src.double().reshape(B * n_src, 256) @ W0[:, :256].t()).float()
    pd = (x_dst.double() @ W0[:, 256:512].t()).float()
    pe = (e.double() @ W0[:, 512:].t()).float()
    lin = [m for m in ep.edge_mlp.model if isinstance(m, torch.nn.Linear)]
    norm = ep.edge_mlp.model[-1]
    pm = PackedMLP([l.weight.detach().to(DEV) for l in lin], [l.bias.detach().to(DEV) for l in lin],
                   (norm.weight.detach().to(DEV), norm.bias.detach().to(DEV)), ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
    agg = torch.zeros((B * n_dst, 256), device=DEV)
    e_out = torch.empty((B * E, 256), device=DEV)
    ops.edge_update_forward(pm, B, st.int().to(DEV), dt.int().to(DEV), Operand(ps.to(DEV), n_src, 256, projected=True),
                            Operand(pd.to(DEV), 0, 256, projected=True) if use_dst else ops.ZERO,
                            Operand(pe.to(DEV), 0, 256, projected=True),
                            # residual: the batch-shared edge rows as fp32 rows, or as ONE shared set of bf16 edge tiles
                            Operand(ops.edge_rows_to_tiles(e.to(DEV), 1, E, E), 0, 256, tiles=True) if res_tiles else Operand(e.to(DEV), 0, 256),
                            n_dst, agg, e_out)
    torch.cuda.synchronize()
    err_e = (e_out.cpu().reshape(B, E, 256) - e_ref).abs().max().item() / e_ref.abs().max().item()
    err_a = (agg.cpu().reshape(B, n_dst, 256) - agg_ref).abs().max().item() / agg_ref.abs().max().item()
    print(f"[edge16 vs oracle] use_dst={use_dst} res_tiles={res_tiles}: e' max-rel {err_e:.2e}, aggregate max-rel {err_a:.2e}")
    assert 1e-5 < err_e <= 2e-2 and err_a <= 2e-2


def test_edge_rows_to_tiles_matches_the_layout_statement():
    """gw_edge_rows_to_tiles against the numpy statement of GW_LAYOUT_EDGE_TILES_BF16 (tests/helpers.py), ragged last tile."""
    from .helpers import edge_tiles_from_rows

    rs = np.random.RandomState(4)
    for B, E, shared in ((3, 130, False), (1, 64, False), (2, 5, True)):
        rows = torch.from_numpy(rs.standard_normal((1 if shared else B, E, 256)).astype(np.float32))
        tiles = ops.edge_rows_to_tiles(rows.reshape(-1, 256).to(DEV), B, E, 0 if shared else E)
        ref = edge_tiles_from_rows(rows.expand(B, E, 256) if shared else rows)
        got = tiles.cpu().view(torch.bfloat16).reshape(ref.shape)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("E,out_kind", [(333, "tiles"), (64, "rows"), (700, None), (5, "tiles")])
def test_edge16_tile_path_matches_bf16_emulation_and_oracle(E, out_kind):
    """Processor-block form of the bf16 edge update: the per-sample edge features arrive as bf16 edge tiles (raw layer-1
    operand on edge16_l1_kernel with W_e in LDS, and residual), node operands pre-projected, e' leaves as tiles / rows / not at
    all.  Against (a) a float64 emulation of exactly this arithmetic (bf16-rounded operands, exact products) and (b) the oracle
    on the raw rows (bf16 budget 2e-2)."""
    from graph_weather_amd.utils import deterministic_fill_
    from oracle import reference_math as om
    import graph_weather_amd as gw

    from .helpers import edge_rows_from_tiles, edge_tiles_from_rows

    rs = np.random.RandomState(E)
    B, n = 3, 50
    ep = gw.EdgeProcessor(256, 256, 256, 2, "LayerNorm")
    deterministic_fill_(ep, seed=23)
    p = {"blk.edge_model." + k: v.clone() for k, v in ep.state_dict().items()}
    x = torch.from_numpy(rs.standard_normal((B, n, 256)).astype(np.float32))
    e = torch.from_numpy(rs.standard_normal((B, E, 256)).astype(np.float32))
    dst = np.sort(np.where(rs.rand(E) < 0.3, n // 2, rs.randint(0, n, size=E)))
    src = rs.randint(0, n, size=E)
    st, dt = torch.from_numpy(src), torch.from_numpy(dst)
    lin = [m for m in ep.edge_mlp.model if isinstance(m, torch.nn.Linear)]
    norm = ep.edge_mlp.model[-1]
    W0 = lin[0].weight.detach().double()
    ps = (x.double().reshape(B * n, 256) @ W0[:, :256].t()).float()
    pd = (x.double().reshape(B * n, 256) @ W0[:, 256:512].t()).float()
    # (a) emulation
    e16 = _bf(e)
    z1 = lin[0].bias.detach().double() + ps.double().reshape(B, n, 256)[:, st] + pd.double().reshape(B, n, 256)[:, dt] + e16 @ _bf(W0[:, 512:].float()).t()
    h1 = _bf(torch.relu(z1).float())
    h2 = _bf(torch.relu(h1 @ _bf(lin[1].weight.detach()).t() + lin[1].bias.detach().double()).float())
    o = h2 @ _bf(lin[2].weight.detach()).t() + lin[2].bias.detach().double()
    y = torch.nn.functional.layer_norm(o, (256,), norm.weight.detach().double(), norm.bias.detach().double(), 1e-5) + e16
    agg_emu = torch.zeros(B, n, 256, dtype=torch.float64)
    agg_emu.index_add_(1, dt, y)
    # (b) oracle
    e_ref = torch.stack([om.edge_processor(p, "blk.edge_model", x[b][st], x[b][dt], e[b]) for b in range(B)])
    agg_ref = torch.stack([om.scatter_sum(e_ref[b], dt, n) for b in range(B)])
    # kernel
    pm = PackedMLP([l.weight.detach().to(DEV) for l in lin], [l.bias.detach().to(DEV) for l in lin],
                   (norm.weight.detach().to(DEV), norm.bias.detach().to(DEV)), ((0, 256), (256, 512), (512, 768)), torch.bfloat16)
    tiles = ops.edge_rows_to_tiles(e.reshape(B * E, 256).to(DEV), B, E, E)
    assert torch.equal(tiles.cpu().view(torch.bfloat16).reshape(-1), edge_tiles_from_rows(e).reshape(-1))
    agg = torch.zeros((B * n, 256), device=DEV)
    e_out = None
    if out_kind == "tiles":
        e_out = torch.full((ops.edge_tiles_bytes(B, E),), 255, dtype=torch.uint8, device=DEV)
    elif out_kind == "rows":
        e_out = torch.empty((B * E, 256), device=DEV)
    ops.edge_update_forward(pm, B, st.int().to(DEV), dt.int().to(DEV), Operand(ps.to(DEV), n, 256, projected=True),
                            Operand(pd.to(DEV), n, 256, projected=True), Operand(tiles, E, 256, tiles=True),
                            Operand(tiles, E, 256, tiles=True), n, agg, e_out)
    torch.cuda.synchronize()
    a = agg.cpu().double().reshape(B, n, 256)
    err_emu = (a - agg_emu).abs().max().item() / agg_emu.abs().max().item()
    err_orc = (a - agg_ref.double()).abs().max().item() / agg_ref.abs().max().item()
    print(f"[edge16 tiles] E={E} out={out_kind}: aggregate vs emulation {err_emu:.2e}, vs oracle {err_orc:.2e}")
    assert err_emu < 3e-4 and err_orc <= 2e-2
    if out_kind == "rows":
        err_e = (e_out.cpu().double().reshape(B, E, 256) - y).abs().max().item() / y.abs().max().item()
        assert err_e < 1.5e-3, err_e
    if out_kind == "tiles":
        neb = (E + 63) // 64
        got = e_out.cpu().view(torch.bfloat16).reshape(B, neb, 4, 8, 64, 8)
        rows = edge_rows_from_tiles(got, E).double()
        # bf16 of y: one unit in the last place of bf16 (2^-8 relative) where fp32 and float64 sums round differently
        assert ((rows - y).abs() <= 2.0 ** -7 * y.abs() + 1e-3 * y.abs().max()).all()
        pad = edge_rows_from_tiles(got, neb * 64)[:, E:]
        assert (pad == 0).all(), "padding edges of the last tile must be written as zeros"
