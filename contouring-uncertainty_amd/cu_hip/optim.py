"""Fused Adam on the HIP kernel, with ``torch.optim.Adam`` semantics AND ``torch.optim.Adam`` state layout.

Replaces ``torch.optim.Adam(lr=1e-3, weight_decay=1e-3)`` instantiated by ``VitalSystem.configure_optimizers``
(reference vital/vital/system.py:82-115, vital/vital/config/task/optim/adam.yaml:1-4): L2 weight decay folded into the
gradient, bias-corrected first/second moments, parameters whose ``.grad`` is None are skipped (the six unused
``deep_supervision_heads`` tensors never move, as in the reference).

Parameters that live back-to-back in one flat buffer (``UNet.flat_params``) with back-to-back gradients are updated
by ONE kernel launch per run.  The optimizer state is nevertheless PER PARAMETER (``state[p] = {"step", "exp_avg",
"exp_avg_sq"}`` like ``torch.optim.Adam``): the moments of a run live in one flat buffer and every parameter's entry is
a view into it, so ``state_dict()`` / ``load_state_dict()`` interchange with ``torch.optim.Adam`` (a Lightning resume,
an ``optimizer_states`` entry of a reference checkpoint).  After a ``load_state_dict`` -- or a re-flatten / ``.to()`` --
the loaded per-parameter tensors are no longer views of one buffer: the flat moments are then REBUILT from them (never
silently zeroed).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import ops


def _step_of(st) -> int:
    s = st.get("step", 0)
    return int(s.item()) if torch.is_tensor(s) else int(s)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 capturable: bool = False, **_unused):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference config")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        # capturable: the step count used for the bias correction lives on the device, so that a hipGraph-captured
        # training step replays with the right correction (cu_hip.graph.CapturedStep); every parameter that takes part
        # must then have been stepped equally often (true for the DSNT tasks: all used parameters get a gradient)
        self.capturable = capturable
        self._steps_dev = None
        # flat moment buffers per run, keyed by the run's first parameter (NOT part of state_dict: the per-parameter
        # views in self.state are)
        self._flat: Dict[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.grad_scale = 1.0          # multiplies every gradient inside the kernel (1 / world after a SUM all-reduce)
        # per parameter group: the launches of the last step (runs, flat moments, step count).  The next step replays them after
        # checking that nothing moved -- same parameter list, same addresses, gradients laid out as before -- instead of
        # re-deriving runs, step splits and moment views from 138 parameters (1.3 ms of host time per step; round 4)
        self._plans: Dict[int, dict] = {}

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
                # back to back in memory AND inside one storage each (separately allocated tensors can be neighbours by
                # accident; a strided view cannot span two storages)
                if (q.data_ptr() + q.numel() * 4 == p.data_ptr()
                        and q.grad.data_ptr() + q.numel() * 4 == p.grad.data_ptr()
                        and q.untyped_storage().data_ptr() == p.untyped_storage().data_ptr()
                        and q.grad.untyped_storage().data_ptr() == p.grad.untyped_storage().data_ptr()):
                    cur.append(p)
                    continue
                runs.append(cur)
            cur = [p]
        if cur:
            runs.append(cur)
        return runs

    def _split_by_step(self, run: List[torch.Tensor]):
        """A fused launch shares one bias correction: split a memory run where the per-parameter step counts differ
        (parameters that had no gradient in some earlier step)."""
        out, cur, cur_step = [], [], None
        for p in run:
            s = _step_of(self.state[p]) if p in self.state else 0
            if cur and s != cur_step:
                out.append(cur)
                cur = []
            cur.append(p)
            cur_step = s
        if cur:
            out.append(cur)
        return out

    def _moments(self, run: List[torch.Tensor]):
        """(flat exp_avg, flat exp_avg_sq, step counts) of a run with every parameter's state entries views into them (the
        ``step`` entries are 0-dim views of ONE host tensor per run: one increment per launch instead of one per parameter)."""
        first = run[0]
        n = sum(p.numel() for p in run)
        flat = self._flat.get(first)
        ok = flat is not None and flat[0].numel() == n and flat[0].device == first.device and flat[2].numel() == len(run)
        if ok:
            off = 0
            base_m, base_v, base_s = flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr()
            for i, p in enumerate(run):
                st = self.state.get(p)
                if (st is None or "exp_avg" not in st or st["exp_avg"].data_ptr() != base_m + 4 * off
                        or st["exp_avg_sq"].data_ptr() != base_v + 4 * off
                        or not torch.is_tensor(st.get("step")) or st["step"].data_ptr() != base_s + 4 * i):
                    ok = False
                    break
                off += p.numel()
        if ok:
            return flat
        # (re)build: keep whatever per-parameter state exists (fresh start: zeros)
        m = torch.zeros(n, dtype=torch.float32, device=first.device)
        v = torch.zeros(n, dtype=torch.float32, device=first.device)
        steps = torch.zeros(len(run), dtype=torch.float32)
        off = 0
        for i, p in enumerate(run):
            st = self.state[p]
            k = p.numel()
            if "exp_avg" in st:
                m[off:off + k].copy_(st["exp_avg"].detach().reshape(-1).to(m))
                v[off:off + k].copy_(st["exp_avg_sq"].detach().reshape(-1).to(v))
            steps[i] = float(_step_of(st))
            st["step"] = steps[i]
            st["exp_avg"] = m[off:off + k].view(p.shape)
            st["exp_avg_sq"] = v[off:off + k].view(p.shape)
            off += k
        self._flat[first] = (m, v, steps)
        return m, v, steps

    def _replay(self, plan, group, b1, b2, grad_scale) -> bool:
        """Launch the planned runs if the plan still describes the group exactly; False = take the full path."""
        if plan["params"] is not group["params"] or plan["ids"] != list(map(id, group["params"])):
            return False
        for p in plan["idle"]:
            if p.grad is not None:
                return False
        for first, n, m, v, step, offs, steps in plan["launches"]:
            g0 = first.grad
            if g0 is None:
                return False
            pbase, gbase = first.data_ptr(), g0.data_ptr()
            gstore = g0.untyped_storage().data_ptr()
            for p, off in offs:
                g = p.grad
                if (g is None or p.data_ptr() != pbase + 4 * off or g.data_ptr() != gbase + 4 * off
                        or g.untyped_storage().data_ptr() != gstore or not g.is_contiguous()):
                    return False
        launches = []
        for first, n, m, v, step, offs, steps in plan["launches"]:
            step += 1
            steps += 1
            pflat = torch.as_strided(first.data, (n,), (1,))
            gflat = torch.as_strided(first.grad, (n,), (1,))
            if self.capturable:
                ops.adam_step_dev(pflat, gflat, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._steps_dev,
                                  grad_scale)
            else:
                ops.adam_step(pflat, gflat, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"], step, grad_scale)
            launches.append((first, n, m, v, step, offs, steps))
        plan["launches"] = launches
        return True

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = None):
        loss = None
        ops.pending_wait()                  # gradients produced on a side stream (the skew head's backward)
        if grad_scale is None:
            grad_scale = self.grad_scale
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            plan = self._plans.get(gi)
            if plan is not None and self._replay(plan, group, b1, b2, grad_scale):
                continue
            self._plans.pop(gi, None)
            built = []          # (first, n, m, v, step after this call, [(p, offset in floats)], step counts) per launch: next step's plan
            for mem_run in self._runs(group["params"]):
                for run in self._split_by_step(mem_run):
                    first = run[0]
                    n = sum(p.numel() for p in run)
                    m, v, steps = self._moments(run)
                    step = _step_of(self.state[first]) + 1
                    steps += 1
                    pflat = torch.as_strided(first.data, (n,), (1,))
                    gflat = torch.as_strided(first.grad, (n,), (1,))
                    if self.capturable:
                        if self._steps_dev is None:
                            self._steps_dev = torch.full((1,), step - 1, dtype=torch.int32, device=first.device)
                        ops.adam_step_dev(pflat, gflat, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                          self._steps_dev, grad_scale)
                    else:
                        ops.adam_step(pflat, gflat, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                                      grad_scale)
                    offs, off = [], 0
                    for p in run:
                        offs.append((p, off))
                        off += p.numel()
                    built.append((first, n, m, v, step, offs, steps))
            self._plans[gi] = {"params": group["params"], "ids": list(map(id, group["params"])), "launches": built,
                               "idle": [p for p in group["params"] if p.grad is None]}
        if self.capturable and self._steps_dev is not None:
            ops.step_advance(self._steps_dev)
        return loss

    def load_state_dict(self, state_dict):
        """``torch.optim.Adam`` layout in; the device step counter and the flat moment buffers of an optimizer that has
        already stepped belong to the OLD state: both are dropped and rebuilt from what was loaded at the next
        ``step`` (``_moments`` re-packs the per-parameter tensors, the counter restarts from the loaded ``step``)."""
        super().load_state_dict(state_dict)
        self._steps_dev = None
        self._flat.clear()
        self._plans.clear()

    def hyper_key(self):
        """the scalars a captured step bakes into its launches (cu_hip.graph.CapturedStep re-captures when they change)"""
        return tuple((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in self.param_groups) + (self.grad_scale,)

    def note_replayed_steps(self, k: int):
        """A captured step was replayed k times: bring the per-parameter host counters (state_dict) up to date."""
        for st in self.state.values():
            if "step" in st:
                st["step"] += k
        self._plans.clear()
