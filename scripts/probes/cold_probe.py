"""Cold forwards only (every parameter version bumped before each forward), for rocprofv3 --kernel-trace --stats: where the
per-weight-version work of a forward goes (packing, edge / mesh embeddings, their layer-1 products)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from graph_weather_amd.optim import _bump_versions

dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]
model, lat_lons = bench.build_model(cfg, dev)
model = model.to(dev).eval()
torch.manual_seed(42)
feats = torch.randn(cfg["batch"], len(lat_lons), 102, device=dev)
with torch.no_grad():
    for _ in range(2):
        model(feats)
    ts = []
    for _ in range(6):
        _bump_versions(model.parameters())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(feats)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
print("cold forwards ms:", [round(t, 2) for t in ts])
