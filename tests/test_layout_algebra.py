"""CPU check of the index algebra the HIP kernels rely on: packed weight order, the K walk order
k(s,q) = 16*(s>>2) + 4*q + (s&3), and the fact that an accumulator tile is directly the next layer's B operand.
A 64-lane wave is emulated with the MFMA operand layouts documented for gfx950 (v_mfma_f32_16x16x4_f32)."""
import numpy as np

from .helpers import k_of, mfma_16x16x4_emulate, pack_linear_ref


def _layer(wp, in_regs, nt):
    """in_regs: [nsteps, 64] per-lane B operands. returns acc [nt, 4, 64]."""
    nsteps = wp.shape[0]
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
