"""Training step of the reference training script's model (train/run.py:486-521: 1024-wide forecaster, NormalizedMSELoss, AdamW)
on the 1 degree grid, batch 1, on the wide path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graph_weather_amd as gw
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons

dev = torch.device("cuda:0")
W = 1024
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll, edge_dim=W, hidden_dim_processor_edge=W, node_dim=W, hidden_dim_processor_node=W, hidden_dim_decoder=W)
deterministic_fill_(m, 0)
m = m.to(dev).train()
crit = gw.NormalizedMSELoss([1.0] * 78, ll, normalize=False).to(dev)
opt = gw.AdamW(m.parameters(), lr=1e-5)
x = torch.randn(1, len(ll), 102, device=dev)
y = torch.randn(1, len(ll), 78, device=dev)


def step():
    opt.zero_grad()
    loss = crit(m(x), y)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
n = 3
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print("wide 1024 training step, 1 degree, B=1: %.1f ms, loss %.4f, peak memory %.1f GB, %d parameters" %
      (dt * 1e3, float(loss.detach()), torch.cuda.max_memory_allocated() / 2**30, sum(p.numel() for p in m.parameters())))
