"""What a graph_weather user changes to run the forecaster hot path on an MI355X.

    python examples/switch_from_reference.py [--grid 5] [--train-steps 2] [--rollout 3]

Everything below is the reference's own usage (README.md:40-75 of openclimatefix/graph_weather, train/run.py:506-521) with
the import line changed; the checkpoint round trip shows that `state_dict` keys are the reference's."""
import argparse
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

# from graph_weather import GraphWeatherForecaster                      # reference
# from graph_weather.models.losses import NormalizedMSELoss             # reference
from graph_weather_amd import AdamW, GraphWeatherForecaster, NormalizedMSELoss, rollout  # MI355X-native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=float, default=5.0)
    ap.add_argument("--train-steps", type=int, default=2)
    ap.add_argument("--rollout", type=int, default=3)
    args = ap.parse_args()
    step = args.grid
    lat_lons = [(float(lat), float(lon)) for lat in torch.arange(-90, 90, step) for lon in torch.arange(0, 360, step)]
    device = torch.device("cuda:0")

    model = GraphWeatherForecaster(lat_lons)                 # same constructor arguments as the reference
    # a reference checkpoint loads as is: identical state_dict keys (here: a round trip through torch.save)
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    model.load_state_dict(torch.load(buf), strict=True)
    model = model.to(device)

    features = torch.randn((2, len(lat_lons), 78 + 24), device=device)
    target = torch.randn((2, len(lat_lons), 78), device=device)
    criterion = NormalizedMSELoss(lat_lons=lat_lons, feature_variance=torch.rand(78) + 0.5, normalize=True)
    optimizer = AdamW(model.parameters(), lr=1e-3)          # or torch.optim.AdamW: gradients are ordinary .grad tensors

    for i in range(args.train_steps):                        # train/run.py:509-521
        optimizer.zero_grad()
        loss = criterion(model(features), target)
        loss.backward()
        optimizer.step()
        print(f"train step {i}: loss {loss.item():.5f}")

    model.eval()
    with torch.no_grad():                                    # inference kernels (no activations kept)
        out = model(features)
    print("forecast", tuple(out.shape), "finite:", bool(torch.isfinite(out).all()))
    states = rollout(model, features, steps=args.rollout)    # autoregressive use: output + aux channels -> next input
    print("rollout steps:", len(states), "last mean |x|:", float(states[-1].abs().mean()))

    # Optional, beyond the reference: the matrix products on the bf16 matrix cores WITHOUT leaving the fp32 results' neighbourhood -
    # every product as three bf16 MFMAs on hi / lo operand pairs ("bf16x3").  Same tensors, same state_dict; ~2e-5 of the
    # forecast's delta scale against fp32, about twice the forecasts/s; also trains (mixed precision, fp32 master weights).
    model.set_compute_dtype("bf16x3")
    with torch.no_grad():
        out3 = model(features)
    delta = (out - features[..., :78]).abs().max().item()
    print("bf16x3 vs fp32: max |diff| / max |delta| =", float((out3 - out).abs().max().item() / max(delta, 1e-30)))


if __name__ == "__main__":
    main()
