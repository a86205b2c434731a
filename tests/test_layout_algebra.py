"""CPU check of the index algebra the HIP kernels rely on: packed weight order, the K walk order
k(s,q) = 16*(s>>2) + 4*q + (s&3), and the fact that an accumulator tile is directly the next layer's B operand.
A 64-lane wave is emulated with the MFMA operand layouts documented for gfx950 (v_mfma_f32_16x16x4_f32)."""
import numpy as np

from .helpers import k_of, mfma_16x16x4_emulate, pack_linear_ref


def _layer(wp, in_regs, nt):
    """in_regs: [nsteps, 64] per-lane B operands. returns acc [nt, 4, 64]."""
    nsteps = min(wp.shape[0], len(in_regs))  # the pack may be zero-padded beyond the operand's K-steps (gw_packed_floats)
    acc = np.zeros((nt, 4, 64))
    for s in range(nsteps):
        for t in range(nt):
            acc[t] = mfma_16x16x4_emulate(wp[s, t >> 2, :, t & 3], in_regs[s], acc[t])
    return acc


def test_two_layer_chain_register_resident():
    rs = np.random.RandomState(0)
    k_in, hid, n_out = 32, 64, 40  # 8 steps, 4 hidden tiles, 3 output tiles (8 padded rows)
    w1 = rs.standard_normal((hid, k_in)).astype(np.float32)
    w2 = rs.standard_normal((n_out, hid)).astype(np.float32)
    x = rs.standard_normal((16, k_in)).astype(np.float32)  # 16 columns (edges), row major like the gathered tables
    # operand load: lane (j, q) holds in[s] = x[j, k(s, q)]
    in1 = np.zeros((k_in // 4, 64))
    for s in range(k_in // 4):
        for lane in range(64):
            in1[s, lane] = x[lane & 15, k_of(s, lane >> 4)]
    acc1 = _layer(pack_linear_ref(w1, 0, k_in), in1, hid // 16)
    # accumulator -> next operand without any data movement: in2[4 t + r] = relu(acc[t][r])
    in2 = np.maximum(acc1.reshape(hid // 16 * 4, 64), 0.0)
    nt_out = (n_out + 15) // 16
    acc2 = _layer(pack_linear_ref(w2, 0, hid), in2, nt_out)
    # read the result back with the store mapping of the kernels: feature = 16 t + 4 q + r
    y = np.zeros((16, 16 * nt_out))
    for lane in range(64):
        j, q = lane & 15, lane >> 4
        for t in range(nt_out):
            for r in range(4):
                y[j, 16 * t + 4 * q + r] = acc2[t, r, lane]
    ref = np.maximum(x @ w1.T, 0.0) @ w2.T
    np.testing.assert_allclose(y[:, :n_out], ref, rtol=1e-5, atol=1e-4)
    assert np.all(y[:, n_out:] == 0.0)  # padded output rows stay exactly zero


def test_split_layer1_equals_concat():
    """cat[x_s, x_d, e] @ W1^T == sum of the three column slices (graph_net_block.py:131-134)."""
    rs = np.random.RandomState(1)
    w = rs.standard_normal((16, 48)).astype(np.float32)
    xs = [rs.standard_normal((16, 16)).astype(np.float32) for _ in range(3)]
    acc = np.zeros((1, 4, 64))
    for i, x in enumerate(xs):
        wp = pack_linear_ref(w, 16 * i, 16 * i + 16)
        in1 = np.array([[x[lane & 15, k_of(s, lane >> 4)] for lane in range(64)] for s in range(4)])
        for s in range(4):
            acc[0] = mfma_16x16x4_emulate(wp[s, 0, :, 0], in1[s], acc[0])
    y = np.zeros((16, 16))
    for lane in range(64):
        for r in range(4):
            y[lane & 15, 4 * (lane >> 4) + r] = acc[0, r, lane]
    np.testing.assert_allclose(y, np.concatenate(xs, axis=1) @ w.T, rtol=1e-5, atol=1e-5)


def test_edge16_activation_exchange_layout():
    """csrc/gw_edge16.hip: four waves each own 64 output features of a layer (wave w, row tile t, lane (j, q) holds features
    64 w + 16 t + 4 q + r of column j after the MFMAs) and write them to LDS as the NEXT layer's B operand without a
    shuffle: 8-byte half (t & 1) of lane (j, q) in K-step s = 2 w + (t >> 1).  The consumer lane (j, q) of K-step s must
    then hold k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3) - the K order gw_pack_linear_bf16 packs the weights in.  The
    gather launch fills the same layout from row-major tables with eight lanes per 128-byte line."""
    from .helpers import k16_of

    rs = np.random.RandomState(1)
    h = rs.standard_normal((16, 256))  # [column j, feature]: what the producing layer computed for one 16-column group
    # ---- producer side (end of a layer in edge16_kernel) ----
    buf = np.full((8, 64, 8), np.nan)  # Hbuf[g][s][lane][i]
    for w in range(4):
        for t in range(4):
            for lane in range(64):
                j, q = lane & 15, lane >> 4
                vals = [h[j, 64 * w + 16 * t + 4 * q + r] for r in range(4)]  # accumulator tile t of wave w, rows 4q .. 4q+3
                s, half = 2 * w + (t >> 1), t & 1
                buf[s, lane, 4 * half:4 * half + 4] = vals
    assert not np.isnan(buf).any()
    # ---- consumer side: B fragment of K-step s in lane (j, q) ----
    for s in range(8):
        for lane in range(64):
            j, q = lane & 15, lane >> 4
            for i in range(8):
                assert buf[s, lane, i] == h[j, k16_of(s, q, i)]
    # ---- gather launch: lanes (row j' = 8 half_rows + lane >> 3, piece p = lane & 7) read features 32 s + 4 p .. + 3 ----
    buf2 = np.full((8, 64, 8), np.nan)
    for hr in range(2):
        for lane in range(64):
            j, piece = 8 * hr + (lane >> 3), lane & 7
            q, half = piece & 3, piece >> 2
            for s in range(8):
                buf2[s, 16 * q + j, 4 * half:4 * half + 4] = h[j, 32 * s + 4 * piece:32 * s + 4 * piece + 4]
    assert np.array_equal(buf, buf2)
    # ---- and the weights a wave keeps resident: rows 64 w .. 64 w + 63 = row tiles 4 w .. 4 w + 3 of the packed stream ----
    from .helpers import pack_linear_bf16_ref

    wmat = rs.standard_normal((256, 256)).astype(np.float32)
    packed = pack_linear_bf16_ref(wmat, 0, 256)  # [s][tile][lane][i]
    y = np.zeros((16, 256))
    for w in range(4):
        for t in range(4):
            tile = 4 * w + t
            for s in range(8):
                a = packed[s, tile]  # A fragment: lane (row = lane & 15, q) holds W[16 tile + row, k(s, q, i)]
                for lane_a in range(64):
                    row, q = lane_a & 15, lane_a >> 4
                    for j in range(16):
                        y[j, 16 * tile + row] += float(np.dot(a[lane_a], buf[s, 16 * q + j]))
    assert np.allclose(y, h @ wmat.T.astype(np.float64), atol=1e-9)
