"""Where the cold step's extra time sits: the forward right after a weight update of (a) every parameter, (b) the decoder's only,
(c) encoder + processor only, against the warm forward (1 degree, batch 2; eager, HIP graph off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from graph_weather_amd.optim import _bump_versions

PRECS = sys.argv[1:] if len(sys.argv) > 1 else ["fp32", "bf16x3"]
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]
model, lat_lons = bench.build_model(cfg, dev)
model = model.to(dev).eval()
model.auto_graph = False
torch.manual_seed(42)
feats = torch.randn(cfg["batch"], len(lat_lons), 102, device=dev)


def run(params, n=6):
    ts = []
    with torch.no_grad():
        for _ in range(n):
            if params is not None:
                _bump_versions(params)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(feats)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts[len(ts) // 2]


dec = list(model.decoder.parameters())
rest = list(model.encoder.parameters()) + list(model.processor.parameters())
for PREC in PRECS:
    bench.set_precision(model, PREC)
    with torch.no_grad():
        for _ in range(3):
            model(feats)
    print(PREC, "warm %.2f ms | all cold %.2f | decoder cold %.2f | encoder + processor cold %.2f"
          % (run(None), run(list(model.parameters())), run(dec), run(rest)), flush=True)
