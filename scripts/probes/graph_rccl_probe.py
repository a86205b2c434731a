"""Does the automatic HIP graph of the eval forward (graphed.AutoGraph) capture while an RCCL process group - and its watchdog
thread - is alive in the process?  (bench.py --gpus N runs exactly that on every rank.)  World size 1 on one GPU: the
communicator, a collective before and after, the forward replayed in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import graph_weather_amd as gw
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
ll = regular_lat_lons(5.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval()
x = seeded_features(2, len(ll), 102, seed=1).to(dev)
with torch.no_grad():
    ref = m._forward_eager(x)
    ys = [m(x) for _ in range(6)]
    auto = m.__dict__["_auto"]
    print("auto graph enabled", auto.enabled, "captures", None if auto._fg is None else auto._fg.captures, "pinned", None if auto._fg is None else auto._fg.pinned)
    print("max |replay - eager|", max((y - ref).abs().max().item() for y in ys))
    time.sleep(3.0)  # let the watchdog thread run beside an idle graph
    dist.all_reduce(t); dist.barrier()
    y = m(x); torch.cuda.synchronize()
    print("after collectives:", (y - ref).abs().max().item(), "captures", auto._fg.captures if auto._fg else None)
dist.destroy_process_group()
print("ok")
