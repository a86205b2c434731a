"""Where a training step's wall time goes (host clocks around synchronised phases), for the two loops of bench.py: the
``--mode train`` loop (train_bench) and the ``extra.train`` loop (run_train_extra)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import graph_weather_amd as gw
from graph_weather_amd import sharding as sh

dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]
model, lat_lons = bench.build_model(cfg, dev)
model = model.to(dev).train()
if os.environ.get("PRECISION"):
    bench.set_precision(model, os.environ["PRECISION"])
crit = gw.NormalizedMSELoss([1.0] * 78, lat_lons, normalize=False)
flat = sh.FlatGradients(model.parameters())
if os.environ.get("ATTACH", "0") == "1":
    flat.attach(sh.ShardContext(0, 0, 1, None))
opt = gw.AdamW(model.parameters(), lr=1e-4, flat=flat)
torch.manual_seed(42)
feats = torch.randn(cfg["batch"], len(lat_lons), 102, device=dev)
target = torch.randn(cfg["batch"], len(lat_lons), 78, device=dev)

def sync():
    torch.cuda.synchronize()
    return time.perf_counter()

for it in range(6):
    t0 = sync()
    flat.zero_()
    y = model(feats)
    loss = crit(y, target)
    t1 = sync()
    loss.backward()
    t2 = sync()
    opt.step()
    t3 = sync()
    print(f"step {it}: forward+loss {1e3 * (t1 - t0):.1f} ms  backward {1e3 * (t2 - t1):.1f} ms  optimizer {1e3 * (t3 - t2):.1f} ms")
t0 = sync()
for it in range(5):
    flat.zero_()
    loss = crit(model(feats), target)
    loss.backward()
    opt.step()
t1 = sync()
print(f"unsynchronised loop: {1e3 * (t1 - t0) / 5:.1f} ms per step")
