"""``AdamW`` on the fused HIP kernel ``gw_adamw_step`` (same update rule and defaults as ``torch.optim.AdamW``, which
the reference's training loops use: train/run.py:506)."""
from __future__ import annotations

import torch

from . import _lib


def _bump_versions(params) -> None:
    """The kernel updates parameters through raw pointers; the packed-weight / embedding caches of the modules key on
    ``Tensor._version``, so it is advanced here - without launching anything."""
    ps = tuple(params)
    if not ps:
        return
    try:
        torch._C._autograd._unsafe_set_version_counter(ps, tuple(p._version + 1 for p in ps))
    except (TypeError, AttributeError):
        # older torch: (Tensor, int) signature, or no such hook at all - fall back to an in-place no-op per tensor
        for p in ps:
            try:
                torch._C._autograd._unsafe_set_version_counter(p, p._version + 1)
            except (TypeError, AttributeError):
                with torch.no_grad():
                    p.add_(0)


class AdamW(torch.optim.Optimizer):
    """``flat``: a ``sharding.FlatGradients`` that owns the parameters and gradients as views of two flat buffers - the whole
    model is then updated by ONE kernel launch per step (one per parameter tensor otherwise: 215 for the forecaster)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = flat
        if flat is not None:
            if flat.param is None:
                raise ValueError("graph_weather_amd.AdamW: flat must own the parameters (FlatGradients(flatten_params=True))")
            mine = {id(p) for g in self.param_groups for p in g["params"] if p.requires_grad}
            if len(self.param_groups) != 1 or mine != {id(p) for p in flat.params}:
                raise ValueError("graph_weather_amd.AdamW: flat must cover exactly the parameters of the single parameter group")
            self._flat_state = {"step": 0, "exp_avg": torch.zeros_like(flat.param), "exp_avg_sq": torch.zeros_like(flat.param)}

    # The flat moments and the bias-correction step live outside torch.optim.Optimizer.state (they are not per parameter):
    # carry them through state_dict() / load_state_dict() so that a resumed run does not silently restart them from zero.
    def state_dict(self):
        sd = super().state_dict()
        if self.flat is not None:
            st = self._flat_state
            sd["gw_flat_state"] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fs = state_dict.pop("gw_flat_state", None)
        if fs is not None and self.flat is None:
            # the flat moments / step cannot be dropped silently: a per-parameter AdamW would restart them from zero
            raise ValueError("graph_weather_amd.AdamW: state_dict carries 'gw_flat_state' (saved from an optimizer built with "
                             "flat=...); build this optimizer with flat= as well to resume it")
        super().load_state_dict(state_dict)
        if self.flat is not None:
            if fs is None:
                raise ValueError("graph_weather_amd.AdamW: state_dict has no 'gw_flat_state' (saved from an optimizer without flat=...)")
            st = self._flat_state
            if fs["exp_avg"].numel() != st["exp_avg"].numel():
                raise ValueError("graph_weather_amd.AdamW: flat optimizer state has %d elements, this model needs %d"
                                 % (fs["exp_avg"].numel(), st["exp_avg"].numel()))
            st["step"] = int(fs["step"])
            st["exp_avg"].copy_(fs["exp_avg"])
            st["exp_avg_sq"].copy_(fs["exp_avg_sq"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        if self.flat is not None:
            f, st, group = self.flat, self._flat_state, self.param_groups[0]
            if not f.views_intact():
                raise RuntimeError("graph_weather_amd.AdamW: a parameter or gradient is no longer a view of the flat buffers "
                                   "(use flat.zero_() instead of zero_grad(set_to_none=True); build FlatGradients after model.to())")
            st["step"] += 1
            b1, b2 = group["betas"]
            with torch.cuda.device(f.param.device):
                _lib.check(L.gw_adamw_step(f.numel, f.param.data_ptr(), f.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                           st["exp_avg_sq"].data_ptr(), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                           st["step"], torch.cuda.current_stream(f.param.device).cuda_stream), "gw_adamw_step")
            _bump_versions(f.params)
            return loss
        for group in self.param_groups:
            b1, b2 = group["betas"]
            done = []
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
                with torch.cuda.device(p.device):
                    _lib.check(L.gw_adamw_step(p.numel(), p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                               group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"],
                                               torch.cuda.current_stream(p.device).cuda_stream), "gw_adamw_step")
                done.append(p)
            _bump_versions(done)
        return loss
