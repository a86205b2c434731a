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
if rec.shape[0] == 0:
    raise SystemExit("no stamps: the phase clocks are compiled into tuning builds only - rebuild with GW_TUNING=1 "
                     "(GW_TUNING=1 python -c 'import __graft_entry__ as g; g.build()')")
print(WHICH, "batch", B, "workgroups", rec.shape[0], "env", {k: v for k, v in os.environ.items() if k.startswith("GW_")}, "(ticks = shader cycles)")
# stamps are GW_TS(i) of csrc/gw_edge16t.hip, written on each workgroup's 4th pipeline step; offsets from the team's stamp 0
LA = {0: "step start", 1: "after alpha", 2: "middle layer done", 6: "gather of the next tile issued", 7: "after beta", 9: "gather pass 0 stored",
      8: "segment sums done (lock-step forms)", 10: "gather pass 1 stored", 11: "step end (dst slots published / DMA landed)"}
LB = {0: "step start", 1: "after alpha", 2: "LayerNorm statistics + normalised bf16 operands", 3: "segment sums (MFMA) + stores / LN tile 0", 4: "LN tile 1", 5: "LN tile 2",
      6: "LN tile 3", 7: "after beta", 8: "segment sums done (lock-step forms)", 9: "output layer done", 13: "step end"}
for team, off, lab in (("A", 0, LA), ("B", 16, LB)):
    r = rec[:, off:off + 16].astype(np.int64)
    print("team", team)
    order = sorted((i for i in lab if np.median(r[:, i]) != 0), key=lambda i: np.median(r[:, i] - r[:, 0]))
    prev = 0.0
    for i in order:
        t = np.median(r[:, i] - r[:, 0])
        print(f"  [{i:2d}] {lab[i]:56s} at {t:8.0f}  (+{t - prev:7.0f})  p90 {np.percentile(r[:, i] - r[:, 0], 90):8.0f}")
        prev = t
