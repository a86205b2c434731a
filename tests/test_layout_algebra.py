"""CPU check of the index algebra the HIP kernels rely on: packed weight order, the K walk order
k(s,h) = 8*(s>>2) + 4*h + (s&3), and the fact that an accumulator tile is directly the next layer's B operand.
A 64-lane wave is emulated with the MFMA operand layouts documented for gfx950."""
import numpy as np

from .helpers import k_of, mfma_32x32x2_emulate, pack_linear_ref


def _layer(wp, in_regs, nt):
    """in_regs: [nsteps, 64] per-lane B operands. returns acc [nt, 16, 64]."""
    nsteps = wp.shape[0]
    acc = np.zeros((nt, 16, 64))
    for s in range(nsteps):
        for t in range(nt):
            a_lane = wp[s, t >> 2, :, t & 3]
            acc[t] = mfma_32x32x2_emulate(a_lane, in_regs[s], acc[t])
    return acc


def test_two_layer_chain_register_resident():
    rs = np.random.RandomState(0)
    k_in, hid, n_out = 24, 64, 40  # small: 12 steps, 2 hidden tiles, 2 output tiles (8 padded rows)
    w1 = rs.standard_normal((hid, k_in)).astype(np.float32)
    w2 = rs.standard_normal((n_out, hid)).astype(np.float32)
    x = rs.standard_normal((32, k_in)).astype(np.float32)  # 32 columns (edges), row major like the gathered tables
    # operand load: lane (j, h) holds in[s] = x[j, k(s, h)]
    in1 = np.zeros((k_in // 2, 64))
    for s in range(k_in // 2):
        for lane in range(64):
            in1[s, lane] = x[lane & 31, k_of(s, lane >> 5)]
    acc1 = _layer(pack_linear_ref(w1, 0, k_in), in1, hid // 32)
    # accumulator -> next operand without any data movement: in2[16 t + r] = relu(acc[t][r])
    in2 = np.maximum(acc1.reshape(hid // 32 * 16, 64), 0.0)
    acc2 = _layer(pack_linear_ref(w2, 0, hid), in2, 2)
    # read the result back with the store mapping of the kernels: feature = 32 t + 8 g + 4 h + r  (reg = 4 g + r)
    y = np.zeros((32, 64))
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for t in range(2):
            for g in range(4):
                for r in range(4):
                    y[j, 32 * t + 8 * g + 4 * h + r] = acc2[t, 4 * g + r, lane]
    ref = np.maximum(x @ w1.T, 0.0) @ w2.T
    np.testing.assert_allclose(y[:, :n_out], ref, rtol=1e-5, atol=1e-4)
    assert np.all(y[:, n_out:] == 0.0)  # padded output rows stay exactly zero


def test_split_layer1_equals_concat():
    """cat[x_s, x_d, e] @ W1^T == sum of the three column slices (graph_net_block.py:131-134)."""
    rs = np.random.RandomState(1)
    w = rs.standard_normal((32, 24)).astype(np.float32)
    xs = [rs.standard_normal((32, 8)).astype(np.float32) for _ in range(3)]
    acc = np.zeros((1, 16, 64))
    for i, x in enumerate(xs):
        wp = pack_linear_ref(w, 8 * i, 8 * i + 8)
        in1 = np.array([[x[lane & 31, k_of(s, lane >> 5)] for lane in range(64)] for s in range(4)])
        for s in range(4):
            acc[0] = mfma_32x32x2_emulate(wp[s, 0, :, 0], in1[s], acc[0])
    y = np.zeros((32, 32))
    for lane in range(64):
        for reg in range(16):
            y[lane & 31, (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)] = acc[0, reg, lane]
    np.testing.assert_allclose(y, np.concatenate(xs, axis=1) @ w.T, rtol=1e-5, atol=1e-5)
