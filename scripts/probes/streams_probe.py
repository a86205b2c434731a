"""Samples per launch of the mesh stack: the fused forward splits the batch into `gp.streams` chunks, one HIP stream each
(layers.fused_forward).  Median ms per step of 3 x 20 steps for every split of the batch, eager and replayed as one HIP graph.
CONFIG = a bench.py configuration name (default c3x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import graph_weather_amd as gw
from graph_weather_amd.utils import seeded_features

dev = torch.device("cuda:0")
name = os.environ.get("CONFIG", "c3x3")
cfg = bench.CONFIGS[name]
model, lat_lons = bench.build_model(cfg, dev)
model = model.to(dev).eval()
bench.set_precision(model, cfg["precision"])
feats = seeded_features(cfg["batch"], len(lat_lons), 102, seed=42).to(dev)
gp = model.processor.graph_processor


def timed(fn, steps=20, repeats=3):
    ts = []
    with torch.no_grad():
        for _ in range(3):
            fn()
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0) / steps)
    return sorted(ts)[len(ts) // 2]


for n in [int(s) for s in os.environ.get("STREAMS", "0,1,2,4,8,16").split(",")]:
    if n > cfg["batch"]:
        continue
    gp.streams = n
    eager = timed(lambda: model(feats))
    fg = gw.ForwardGraph(model)
    graph = timed(lambda: fg(feats))
    B = cfg["batch"]
    what = "auto" if n == 0 else "%d samples per launch" % (B // n)
    print(f"[{name}] streams {n} ({what}): eager {eager:.3f} ms ({B / eager * 1e3:.1f} /s), graph {graph:.3f} ms "
          f"({B / graph * 1e3:.1f} /s)", flush=True)
    del fg
