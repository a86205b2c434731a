"""Does a HIP graph of the whole forward (all launches of one model(features) call, side streams included) run faster than the
eager launches?  c2 (fp32, batch 2) and c3 (bf16, batch 16): HIP-event time of 20 eager forwards against 20 graph replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import graph_weather_amd as gw

dev = torch.device("cuda:0")
for name in ("c2", "c3"):
    cfg = bench.CONFIGS[name]
    model, lat_lons = bench.build_model(cfg, dev)
    model = model.to(dev).eval()
    if cfg["precision"] == "bf16":
        gw.layers.set_compute_dtype(model, torch.bfloat16)
    torch.manual_seed(42)
    feats = torch.randn(cfg["batch"], len(lat_lons), 102, device=dev)
    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    with torch.no_grad():
        eager = timed(lambda: model(feats))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                model(feats)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(feats)
        torch.cuda.synchronize()
        ref = model(feats)
        g.replay()
        torch.cuda.synchronize()
        print(name, "graph vs eager max abs diff", (out - ref).abs().max().item())
        graph = timed(lambda: g.replay())
    print(f"{name}: eager {eager:.3f} ms, graph replay {graph:.3f} ms")
    del model, feats, g, out
    torch.cuda.empty_cache()
