"""Timing probe of the bf16 chain kernel (csrc/gw_bf16.hip) on the mesh-sized node update with post products (94 112 rows) and on
a grid-sized node update (1 036 800 rows): HIP-event time per launch.  With a tuning build, GW_CHAIN16_TUNE selects timing
experiments (1: no weight DMA, 2: no MFMAs, 4: no chunk barrier - results are wrong then)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graph_weather_amd as gw
from graph_weather_amd import ops
from graph_weather_amd.ops import Operand
from graph_weather_amd.utils import deterministic_fill_

dev = "cuda:0"
blk = gw.layers.build_graph_processor_block(256, 256, 256, 256, 2, 2, "LayerNorm")
deterministic_fill_(blk, seed=0)
blk = blk.to(dev)
gw.layers.set_compute_dtype(blk, torch.bfloat16)
pm_n = blk.node_model.node_mlp.packed()
pm_e = blk.edge_model.edge_mlp.packed()
for rows, per in ((16 * 5882, 5882), (16 * 64800, 64800)):
    x = torch.randn(rows, 256, device=dev)
    agg = torch.randn(rows, 256, device=dev)
    z = torch.empty(rows, 256, device=dev)
    def run():
        return ops.node_update_forward(pm_n, rows, per, Operand(x, per, 256), Operand(x, per, 256), Operand(agg, per, 256),
                                       post_w=[pm_e.w1[0], pm_e.w1[1]], zero_rows=z, post_half=True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"rows {rows}: {ev[0].elapsed_time(ev[1]) / 10:.4f} ms per launch  (GW_CHAIN16_TUNE={os.environ.get('GW_CHAIN16_TUNE', '0')})")
