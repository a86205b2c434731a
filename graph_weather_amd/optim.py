"""``AdamW`` on the fused HIP kernel ``gw_adamw_step`` (same update rule and defaults as ``torch.optim.AdamW``, which
the reference's training loops use: train/run.py:506)."""
from __future__ import annotations

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("graph_weather_amd.AdamW: parameters must be contiguous fp32 tensors on a HIP device")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad.contiguous()
                _lib.check(L.gw_adamw_step(p.numel(), p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                           group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"],
                                           torch.cuda.current_stream(p.device).cuda_stream), "gw_adamw_step")
                p._version  # noqa: B018  (in-place update through the raw pointer: bump the version below)
                p.add_(0)   # version bump so that packed-weight caches notice the new values
        return loss
