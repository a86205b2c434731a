"""Input-gradient products: the single-layer form of the fused chain kernel (gw_project_forward) against the tiled NT GEMM of
gw_wide.hip, and the TN weight-gradient GEMM, at the row counts of the 1 degree training step (B = 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graph_weather_amd import _lib, ops, wide
from graph_weather_amd.autograd import gemm_tn_acc
from graph_weather_amd.ops import Operand

dev = "cuda:0"
L = _lib.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# the wide path's edge-level product: 453 600 decoder edges, 1024 -> 1024
for rows, k, n in ((453600, 1024, 1024), (41162, 1024, 1024), (64800, 102, 1024)):
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev) / 32
    t = timeit(lambda: wide.linear_forward(x, w, None, True), n=5)
    ref = torch.relu(x[:256].double() @ w.double().t())
    err = (wide.linear_forward(x, w, None, True)[:256].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"gemm_nt {rows} x {k} -> {n}: {t*1e3:8.1f} us ({2.0*rows*k*n/t/1e9:6.1f} TF/s), rel err {err:.1e}")

for rows in (82324, 907200):
    d = torch.randn(rows, 256, device=dev)
    h = torch.randn(rows, 256, device=dev)
    W = torch.randn(256, 256, device=dev) / 16
    Wt = W.t().contiguous()
    pt = torch.empty(L.gw_packed_floats(256, 0, 256), dtype=torch.float32, device=dev)
    _lib.check(L.gw_pack_linear(Wt.data_ptr(), 256, 256, 0, 256, pt.data_ptr(), torch.cuda.current_stream().cuda_stream), "pack")
    fl = 2.0 * rows * 256 * 256
    t1 = timeit(lambda: ops.project_forward([pt], Operand(d, rows, 256), rows, rows, weight_dtype=_lib.DTYPE_F32, relu_mask=h))
    t2 = timeit(lambda: wide.linear_forward(d, Wt, None, False))
    gw_ = torch.zeros(256, 256, device=dev)
    gb = torch.zeros(256, device=dev)
    t3 = timeit(lambda: gemm_tn_acc(d, h, gw_, colsum=gb))
    a = ops.project_forward([pt], Operand(d, rows, 256), rows, rows, weight_dtype=_lib.DTYPE_F32)[0]
    b = wide.linear_forward(d, Wt, None, False)
    err = (a - b).abs().max().item()
    print(f"rows {rows:7d}: chain single {t1*1e3:7.1f} us ({fl/t1/1e9:6.1f} TF/s) | gemm_nt {t2*1e3:7.1f} us ({fl/t2/1e9:6.1f} TF/s) | "
          f"gemm_tn {t3*1e3:7.1f} us ({fl/t3/1e9:6.1f} TF/s) | max diff {err:.2e}")
