"""Dump per-workgroup phase timestamps of the edge kernels of one 1-degree forward (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_weather_amd as gw
from graph_weather_amd import _lib
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

out = sys.argv[1]
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda:0"
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval()
x = seeded_features(2, len(ll)).to(dev)
with torch.no_grad():
    for _ in range(2): y = m(x)
cap = 30000
buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
L = _lib.lib()
# stamp only the decoder block: enable around the decoder call
xp = None
with torch.no_grad():
    xe = m.encoder.encode(x)
    _, lp = m.encoder._plans(x.device)
    el = m.encoder.latent_edge_embedding(lp)
    xp, _ = m.processor.graph_processor.run_plan(xe, lp, el, True, 2, False)
    torch.cuda.synchronize()
    L.gw_debug_timestamps(buf.data_ptr(), cap, kind)
    yd = m.decoder.decode(xp, 2, residual=x.reshape(2 * len(ll), 102))
    torch.cuda.synchronize()
    L.gw_debug_timestamps(None, 0, -1)
rec = buf.cpu().numpy().reshape(cap, 16)
rec = rec[rec[:, 0] != 0]
np.save(out, rec)
print("records", rec.shape)
