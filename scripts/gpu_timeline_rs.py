"""Phase clocks of the row-split node update (csrc/gw_noders.hip, tuning builds): per workgroup (wave 0), medians over the launch.
usage: GW_TUNING=1 python scripts/gpu_timeline_rs.py B PRECISION     PRECISION = fp32 | bf16x3 (the processor's node updates, batch B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_weather_amd as gw
from graph_weather_amd import _lib
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval()
if PREC != "fp32":
    m.set_compute_dtype(PREC)
x = seeded_features(B, len(ll)).to(dev)
cap = 4096
buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
L = _lib.lib()
with torch.no_grad():
    y = m(x)
    xe = m.encoder.encode(x)
    _, lp = m.encoder._plans(x.device)
    el = m.encoder.latent_edge_embedding(lp)
    torch.cuda.synchronize(); L.gw_debug_timestamps(buf.data_ptr(), cap, 4 if PREC != "fp32" else 2)
    xp, _ = m.processor.graph_processor.run_plan(xe, lp, el, True, B, False)
    torch.cuda.synchronize(); L.gw_debug_timestamps(None, 0, -1)
rec = buf.cpu().numpy().reshape(cap, 16)
rec = rec[rec[:, 0] != 0].astype(np.int64)
if rec.shape[0] == 0:
    raise SystemExit("no stamps: the phase clocks are compiled into tuning builds only (GW_TUNING=1)")
print("row-split node update", PREC, "batch", B, "workgroups", rec.shape[0], "env", {k: v for k, v in os.environ.items() if k.startswith("GW_")},
      "(ticks = shader cycles; the last launch that wrote each record)")
LAB = ["start", "layer 1 done (operand loads + raw passes)", "middle layer done", "output layer done", "LayerNorm + residual + stores",
       "post products done", "end"]
prev = 0.0
for i in range(1, 7):
    d = rec[:, i] - rec[:, 0]
    t = np.median(d)
    print(f"  [{i}] {LAB[i]:44s} at {t:8.0f}  (+{t - prev:7.0f})  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    prev = t
print(f"  middle + output passes (8 chunks): waiting for own DMA pieces {np.median(rec[:, 11]):.0f}, workgroup barrier {np.median(rec[:, 12]):.0f} cycles (medians, wave 0)")
st = rec[:, 0] - rec[:, 0].min()
print(f"  workgroup start skew: median {np.median(st):.0f}, p90 {np.percentile(st, 90):.0f}, max {st.max():.0f} ticks; launch span {rec[:, 6].max() - rec[:, 0].min()} ticks")
