"""Experiment (round 6): the batch-2 forward as TWO independent batch-1 pipelines on two HIP streams, the second one started
`delay` ms after the first, against the product's forward (encoder and decoder on the whole batch, only the mesh stack per
sample).  Idea: offset pipelines put one sample's chip-filling launches (node encoder, decoder edge update) beside the other
sample's mesh-sized launches, whose last rounds leave workgroup slots idle.  usage: python scripts/probes/pipeline_offset_probe.py [PRECISION]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graph_weather_amd as gw
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
ll = regular_lat_lons(1.0)
m = gw.GraphWeatherForecaster(ll); deterministic_fill_(m, 0); m = m.to(dev).eval()
if PREC != "fp32":
    m.set_compute_dtype(PREC)
m.auto_graph = False
x = seeded_features(2, len(ll), 102, seed=42).to(dev)
xa, xb = x[0:1].contiguous(), x[1:2].contiguous()
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
clock_hz = 100e6  # torch.cuda._sleep counts device clock ticks of a fixed-rate counter on ROCm builds; calibrated below


def calibrate():
    global clock_hz
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(100_000_000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    clock_hz = 100_000_000 / dt


def two_pipelines(delay_ms):
    s0.wait_stream(main); s1.wait_stream(main)
    with torch.cuda.stream(s0):
        ya = m(xa)
    with torch.cuda.stream(s1):
        if delay_ms > 0:
            torch.cuda._sleep(int(delay_ms * 1e-3 * clock_hz))
        yb = m(xb)
    main.wait_stream(s0); main.wait_stream(s1)
    return ya, yb


def timed(fn, steps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


with torch.no_grad():
    calibrate()
    ref = m(x)
    ya, yb = two_pipelines(0.0)
    torch.cuda.synchronize()
    print("max |two pipelines - batch forward|", max((ya - ref[0:1]).abs().max().item(), (yb - ref[1:2]).abs().max().item()))
    print(f"{PREC}: product forward (batch 2)      {timed(lambda: m(x)):.3f} ms per step   (sleep clock {clock_hz / 1e6:.1f} MHz)")
    print(f"{PREC}: one batch-1 forward            {timed(lambda: m(xa)):.3f} ms")
    for d in (0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
        print(f"{PREC}: two batch-1 pipelines, delay {d:4.2f} ms: {timed(lambda: two_pipelines(d)):.3f} ms per step")
