"""``NormalizedMSELoss`` - reference ``graph_weather/models/losses.py:9-94`` (cos-latitude weighted MSE)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class NormalizedMSELoss(torch.nn.Module):
    """losses.py:12-44 (constructor) / :46-94 (forward).  The four shape prints and the two host-syncing NaN
    asserts of the reference forward are dropped; the value is identical."""

    def __init__(self, feature_variance: list, lat_lons: list, device="cpu", normalize: bool = False):
        super().__init__()
        self.feature_variance = torch.as_tensor(feature_variance, dtype=torch.float32).clone()
        assert not torch.isnan(self.feature_variance).any()
        unique_lats = sorted(set(lat for lat, _ in lat_lons))
        self.weights = torch.tensor([np.cos(lat * np.pi / 180.0) for lat in unique_lats], dtype=torch.float)
        self.normalize = normalize
        assert not torch.isnan(self.weights).any()

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self.feature_variance = self.feature_variance.to(pred.device)
        self.weights = self.weights.to(pred.device)
        inv_var = None
        if self.normalize:
            fv = self.feature_variance
            if fv.numel() == pred.shape[-1]:
                inv_var = (1.0 / fv.reshape(-1)).contiguous()
            else:  # anything that broadcasts against [B, G, C] (losses.py:69; the reference's own test passes [B, G, C])
                inv_var = (1.0 / torch.broadcast_to(fv, pred.shape)).contiguous()
        if torch.is_grad_enabled() and pred.requires_grad:
            from .autograd import NormalizedMSEFunction

            return NormalizedMSEFunction.apply(pred.contiguous(), target.contiguous(), self.weights.contiguous(), inv_var)
        return ops.normalized_mse_forward(pred.contiguous(), target.contiguous(), self.weights.contiguous(), inv_var)
