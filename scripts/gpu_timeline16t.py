"""Phase clocks of the team-pipelined bf16 edge kernel (gw_edge16t.hip) on the 1-degree decoder / processor, batch 16.
Per workgroup and team (A = waves 0-3: middle layer + Hbuf1 of the next tile; B = waves 4-7: output layer + LayerNorm):
wait at alpha | half 1 | wait at beta | segment sums | rest of half 2, stamped on each workgroup's 4th pipeline step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_weather_amd as gw
from graph_weather_amd import _lib
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
WHICH = sys.argv[2] if len(sys.argv) > 2 else "decoder"  # "decoder" or "processor" (last launch of the stack wins)
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
x = seeded_features(B, len(ll)).to(dev)
cap = 256
buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
L = _lib.lib()
with torch.no_grad():
    y = m(x)
    xe = m.encoder.encode(x)
    _, lp = m.encoder._plans(x.device)
    el = m.encoder.latent_edge_embedding(lp)
    if WHICH == "processor":
        torch.cuda.synchronize()
        L.gw_debug_timestamps(buf.data_ptr(), cap, 3)
    xp, _ = m.processor.graph_processor.run_plan(xe, lp, el, True, B, False)
    torch.cuda.synchronize()
    if WHICH == "processor":
        L.gw_debug_timestamps(None, 0, -1)
    else:
        L.gw_debug_timestamps(buf.data_ptr(), cap, 3)
        yd = m.decoder.decode(xp, B, residual=x.reshape(B * len(ll), 102))
        torch.cuda.synchronize()
        L.gw_debug_timestamps(None, 0, -1)
rec = buf.cpu().numpy().reshape(cap, 16)
rec = rec[rec[:, 0] != 0]
names = ["wait at alpha", "half 1", "wait at beta", "segment sums", "rest of half 2"]
print(WHICH, "batch", B, "workgroups", rec.shape[0], "(s_memtime ticks = shader cycles)")
for team, off, what in (("A", 0, "half 1 = middle layer (+ issue of gather pass 0); rest of half 2 = gather / DMA wait of the next tile"),
                        ("B", 8, "half 1 = LayerNorm + residual + staging; rest of half 2 = output layer + residual request")):
    r = rec[:, off:off + 6]
    d = np.diff(r, axis=1)
    print("team", team, "-", what)
    for i, n in enumerate(names):
        print(f"  {n:18s} median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    print(f"  step total         median {np.median(r[:, 5] - r[:, 0]):8.0f}")
