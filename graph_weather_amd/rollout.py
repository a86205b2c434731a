"""Autoregressive rollout: the production use of forecasts/s - the model's output plus the
next step's auxiliary channels become the next input (the data layout of ``features[..., :feature_dim]`` /
``features[..., feature_dim:]`` in forecast.py:215-247)."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


@torch.no_grad()
def rollout(model, features: torch.Tensor, steps: int, aux_fn: Optional[Callable[[int], torch.Tensor]] = None,
            keep: bool = True) -> List[torch.Tensor]:
    """``steps`` forecasts from ``features`` [B, G, feature_dim + aux_dim].  ``aux_fn(t)`` returns the auxiliary channels
    [B, G, aux_dim] of step ``t`` (default: those of the initial state are reused).  Returns the forecasts (all of them, or
    only the last one with ``keep=False``).  One input buffer is reused for every step: the prognostic channels are
    overwritten in place, nothing is concatenated."""
    fdim = int(model.output_dim)
    x = features.clone()
    outs: List[torch.Tensor] = []
    for t in range(steps):
        y = model(x)
        if keep or t == steps - 1:
            outs.append(y)
        if t + 1 < steps:
            x[..., :fdim].copy_(y)
            if aux_fn is not None and x.shape[-1] > fdim:
                x[..., fdim:].copy_(aux_fn(t + 1))
    return outs
