"""Phase clocks of the split-operand kernels (csrc/gw_split.hip, tuning builds): per workgroup (wave 0), medians over the launch.
usage: GW_TUNING=1 python scripts/gpu_timeline_x3.py B WHICH     WHICH = decoder | processor | encoder | node | dechead | nodeenc"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_weather_amd as gw
from graph_weather_amd import _lib
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
WHICH = sys.argv[2] if len(sys.argv) > 2 else "decoder"
KIND = {"decoder": 1, "processor": 1, "encoder": 1, "node": 4, "dechead": 6, "nodeenc": 5}[WHICH]
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval(); m.set_compute_dtype("bf16x3")
x = seeded_features(B, len(ll)).to(dev)
cap = 20000
buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
L = _lib.lib()
on = lambda: (torch.cuda.synchronize(), L.gw_debug_timestamps(buf.data_ptr(), cap, KIND))
off = lambda: (torch.cuda.synchronize(), L.gw_debug_timestamps(None, 0, -1))
with torch.no_grad():
    y = m(x)
    if WHICH in ("encoder", "nodeenc"):
        on()
    xe = m.encoder.encode(x)
    if WHICH in ("encoder", "nodeenc"):
        off()
    _, lp = m.encoder._plans(x.device)
    el = m.encoder.latent_edge_embedding(lp)
    if WHICH in ("processor", "node"):
        on()
    xp, _ = m.processor.graph_processor.run_plan(xe, lp, el, True, B, False)
    if WHICH in ("processor", "node"):
        off()
    if WHICH in ("decoder", "dechead"):
        on()
    yd = m.decoder.decode(xp, B, residual=x.reshape(B * len(ll), 102))
    off()
rec = buf.cpu().numpy().reshape(cap, 16)
rec = rec[rec[:, 0] != 0].astype(np.int64)
if rec.shape[0] == 0:
    raise SystemExit("no stamps: the phase clocks are compiled into tuning builds only (GW_TUNING=1)")
print(WHICH, "batch", B, "workgroups", rec.shape[0], "env", {k: v for k, v in os.environ.items() if k.startswith("GW_")}, "(ticks = shader cycles)")
LAB = ["start", "layer 1 done (gather-adds / raw passes)", "middle layer(s) done", "output layer done", "LayerNorm + residual + stores",
       "head / post products done", "end (segment sums)"]
prev = 0.0
for i in range(1, 7):
    d = rec[:, i] - rec[:, 0]
    t = np.median(d)
    print(f"  [{i}] {LAB[i]:48s} at {t:8.0f}  (+{t - prev:7.0f})  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    prev = t
print(f"  middle + output passes: waiting for own DMA pieces {np.median(rec[:, 11]):.0f}, LDS drain + chunk barrier {np.median(rec[:, 12]):.0f} cycles (medians per workgroup, wave 0)")
span = rec[:, 6].max() - rec[:, 0].min()
print(f"  launch span {span} ticks; sum of workgroup durations / span = {float((rec[:, 6] - rec[:, 0]).sum()) / span:.1f} workgroups in flight on average")
