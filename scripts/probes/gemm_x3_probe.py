"""Host and device cost per launch of the TN GEMM entry (fp32 vs split operands)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graph_weather_amd import _lib

L = _lib.lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for rows in (256, 11764, 82324, 904960):
    a = torch.randn(rows, 256, device=dev)
    b = torch.randn(rows, 256, device=dev)
    c = torch.zeros(256, 256, device=dev)
    cs = torch.zeros(256, device=dev)
    for mode, name in ((_lib.GEMM_TN, "fp32"), (_lib.GEMM_TN_BF16X3, "x3")):
        for _ in range(3):
            _lib.check(L.gw_gemm_f32(mode, 256, 256, rows, a.data_ptr(), 256, b.data_ptr(), 256, c.data_ptr(), 256, cs.data_ptr(), st), "g")
        torch.cuda.synchronize()
        n = 200 if rows < 100000 else 30
        t0 = time.perf_counter()
        for _ in range(n):
            _lib.check(L.gw_gemm_f32(mode, 256, 256, rows, a.data_ptr(), 256, b.data_ptr(), 256, c.data_ptr(), 256, cs.data_ptr(), st), "g")
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        fl = 2.0 * rows * 256 * 256
        print(f"rows {rows:7d} {name:5s}: host issue {1e6 * (t1 - t0) / n:8.1f} us/launch, total {1e6 * (t2 - t0) / n:9.1f} us/launch = {fl / ((t2 - t0) / n) / 1e12:7.1f} TFLOP/s")
