"""Phase clocks of the team-pipelined bf16 edge kernel (gw_edge16t.hip) on the 1-degree decoder / processor, batch 16.
Per workgroup and team (A = waves 0-3: middle layer + Hbuf1 of the next tile; B = waves 4-7: output layer + LayerNorm),
stamped on each workgroup's 4th pipeline step (gw_debug_timestamps kind 3; 32 x u64 per workgroup)."""
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
buf = torch.zeros(cap * 32, dtype=torch.int64, device=dev)
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
rec = buf.cpu().numpy().reshape(cap, 32)
rec = rec[rec[:, 0] != 0]
NA = ["wait at alpha", "mid group 0", "mid group 1", "mid group 2", "mid group 3", "issue gather pass 0", "wait at beta", "segment sums",
      "gather: pass 0 done", "gather: pass 1 done", "dst ids / DMA wait"]
NB = ["wait at alpha", "LN: mean / rstd", "LN: tile 0 (needs residual)", "LN: tile 1 (+ residual request)", "LN: tile 2", "LN: tile 3",
      "wait at beta", "segment sums", "out group 0", "out group 1", "out group 2", "out group 3", "residual request"]
print(WHICH, "batch", B, "workgroups", rec.shape[0], "env", {k: v for k, v in os.environ.items() if k.startswith("GW_")}, "(ticks = shader cycles)")
for team, off, names in (("A", 0, NA), ("B", 16, NB)):
    r = rec[:, off:off + len(names) + 1].astype(np.int64)
    if WHICH == "processor" and team == "A":  # (DMA form: no gather stamps 9 - their slots stay 0)
        r = r.copy(); r[:, 9] = r[:, 8]
    d = np.diff(r, axis=1)
    print("team", team)
    for i, n in enumerate(names):
        print(f"  {n:32s} median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    print(f"  step total                       median {np.median(r[:, len(names)] - r[:, 0]):8.0f}")
