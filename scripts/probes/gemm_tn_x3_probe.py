"""Weight-gradient GEMM on split operands (gw_gemm_f32, GW_GEMM_TN_BF16X3) alone, at the row counts of the 1 degree training step
(B = 2): decoder / encoder edges, their batch-summed tables, a processor block's edges, the mesh nodes.  Reports time, the
operand bytes per second (each operand row counted once) and the product rate.  GW_TN_X3_TUNE (tuning builds): 1 = no atomics,
2 = launch-order slabs (no XCD grouping), 4 = no MFMAs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graph_weather_amd.autograd import gemm_tn_acc

dev = "cuda:0"


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


print("GW_TN_X3_TUNE =", os.environ.get("GW_TN_X3_TUNE", "0"), " GW_TN_TARGET =", os.environ.get("GW_TN_TARGET", "1024"))
for rows in (907200, 453600, 164648, 82324, 23528, 11764):
    d = torch.randn(rows, 256, device=dev)
    h = torch.randn(rows, 256, device=dev)
    gw_ = torch.zeros(256, 256, device=dev)
    gb = torch.zeros(256, device=dev)
    t = timeit(lambda: gemm_tn_acc(d, h, gw_, colsum=gb, x3=True))
    t32 = timeit(lambda: gemm_tn_acc(d, h, gw_, colsum=gb, x3=False))
    print(f"rows {rows:7d}: x3 {t*1e3:7.1f} us  {2 * rows * 1024 / (t * 1e-3) / 1e12:6.2f} TB/s  {3 * 2.0 * rows * 65536 / (t * 1e-3) / 1e12:6.1f} TF/s (3 MFMA products) | fp32 {t32*1e3:7.1f} us")
