"""Fused Adam on the HIP kernel, with ``torch.optim.Adam`` semantics.

Replaces ``torch.optim.Adam(lr=1e-3, weight_decay=1e-3)`` instantiated by ``VitalSystem.configure_optimizers``
(reference vital/vital/system.py:82-115, vital/vital/config/task/optim/adam.yaml:1-4): L2 weight decay folded into the
gradient, bias-corrected first/second moments, parameters whose ``.grad`` is None are skipped (the six unused
``deep_supervision_heads`` tensors never move, as in the reference).

Parameters that live back-to-back in one flat buffer (``UNet.flat_params``) with back-to-back gradients are updated
by ONE kernel launch per run.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **_unused):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference config")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)

    @staticmethod
    def _runs(params: List[torch.Tensor]):
        """Group parameters into maximal runs contiguous in memory for both data and grad."""
        runs, cur = [], []
        for p in params:
            if p.grad is None:
                continue
            ok = p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
            if not ok:
                raise RuntimeError("FusedAdam needs contiguous float32 parameters and gradients")
            if cur:
                q = cur[-1]
                if (q.data_ptr() + q.numel() * 4 == p.data_ptr()
                        and q.grad.data_ptr() + q.numel() * 4 == p.grad.data_ptr()):
                    cur.append(p)
                    continue
                runs.append(cur)
            cur = [p]
        if cur:
            runs.append(cur)
        return runs

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for run in self._runs(group["params"]):
                first = run[0]
                n = sum(p.numel() for p in run)
                key = (first.data_ptr(), n)
                st = self.state[first]
                if st.get("key") != key:
                    # (re)allocate the moments for this run; carry over per-parameter state if the run changed shape
                    st.clear()
                    st["key"] = key
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros(n, dtype=torch.float32, device=first.device)
                    st["exp_avg_sq"] = torch.zeros(n, dtype=torch.float32, device=first.device)
                st["step"] += 1
                pflat = torch.as_strided(first.data, (n,), (1,))
                gflat = torch.as_strided(first.grad, (n,), (1,))
                ops.adam_step(pflat, gflat, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"], grad_scale)
        return loss
