"""Achieved HBM bandwidth of the memory-bound kernels of the path (gather / segment-sum duals of the message passing,
ReLU / LayerNorm backward) on the 1 degree decoder graph, B = 2: algorithmic bytes / HIP-event time vs 8 TB/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from graph_weather_amd import autograd as ag
from graph_weather_amd.graphs import build_forecast_graphs
from graph_weather_amd.utils import regular_lat_lons

dev = "cuda:0"
g = build_forecast_graphs(regular_lat_lons(1.0), 2)
plan = g.dec_plan.to(dev)
B, E, G, M = 2, plan.num_edges, g.num_grid, g.num_mesh
rows = B * E
x = torch.randn(rows, 256, device=dev)
h = torch.relu(torch.randn(rows, 256, device=dev))
table = torch.randn(B * G, 256, device=dev)
gamma = torch.ones(256, device=dev)
db = torch.zeros(256, device=dev)
dg = torch.zeros(256, device=dev)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


KB = 1024
res = {}
res["gather_rows (dagg[dst] -> per edge)"] = (timeit(lambda: ag.gather_rows(table, G, plan.dst, B, E)), rows * KB + rows * KB)
res["segment_sum_rows by dst (sorted)"] = (timeit(lambda: ag.segment_sum_rows(x, E, B, B, G, plan.dst_ptr(), None)), rows * KB + B * G * KB)
perm, ptr = plan.src_sorted()
res["segment_sum_rows by src (permuted)"] = (timeit(lambda: ag.segment_sum_rows(x, E, B, B, M, ptr, perm)), rows * KB + B * M * KB)
res["relu_backward + bias grad"] = (timeit(lambda: ag.relu_backward(x.clone(), h, db)) - timeit(lambda: x.clone()), 3 * rows * KB)
res["layernorm_backward"] = (timeit(lambda: ag.layernorm_backward(x, h, gamma, dg, db)), 3 * rows * KB)
out = {k: {"ms": round(ms, 3), "GB": round(by / 1e9, 3), "GB_per_s": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 3)}
       for k, (ms, by) in res.items()}
print(json.dumps(out, indent=1))
