"""One training step as a hipGraph (torch.cuda.CUDAGraph on ROCm): ~500 kernel launches replayed by one host call.

What makes the DSNT step capturable: no host synchronisation inside it (losses stay device
tensors), static per-layer workspaces (cu_hip.engine: one persistent read-and-clear weight-gradient accumulator, no
per-layer memset), the optimiser's step count on the device (FusedAdam(capturable=True) -> cu_adam_step_dev), the
operand preparation re-run inside the graph (its decision "parameters changed" is taken at capture time, when they
just did).  With N > 1 ranks the gradient all-reduces are part of the graph (RCCL collectives are capturable).

    step = CapturedStep(task, optimizer, batch)        # warms up on a side stream, captures one step
    for _ in range(k): step.replay()                   # batch tensors may be refilled in place between replays
    step.logs["loss"]                                   # device tensors of the last replayed step
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


class CapturedStep:
    def __init__(self, task, optimizer, batch: Dict[str, torch.Tensor], after_backward: Optional[Callable] = None,
                 warmup: int = 3):
        assert getattr(optimizer, "capturable", False), "build the optimizer with capturable=True (device step counter)"
        self.task, self.optimizer, self.batch = task, optimizer, batch
        self.after_backward = after_backward
        self.logs: Dict[str, torch.Tensor] = {}
        self.replays = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):
                self._step(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._capture()
        self.warmup_steps = warmup

    def _capture(self):
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logs = self._step(0)
        # the capture ran the optimizer's host code (its per-parameter step counters moved) but no kernel
        self.optimizer.note_replayed_steps(-1)
        # lr, betas, eps, weight decay and the gradient scale are scalar arguments of the captured launches:
        # an LR scheduler or a param_group edit after the capture must not be ignored by the replays
        self._hyper = self.optimizer.hyper_key()

    def _step(self, i):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.task.training_step(self.batch, i)
        out["loss"].backward()
        if self.after_backward is not None:
            self.after_backward()
        self.optimizer.step()
        return out

    def replay(self):
        if self.optimizer.hyper_key() != self._hyper:       # hyper-parameters changed since the capture: capture again
            torch.cuda.synchronize()
            self.optimizer.note_replayed_steps(self.replays)
            self.replays = 0
            self._capture()
        self.graph.replay()
        self.replays += 1

    def finish(self):
        """host-side bookkeeping after the last replay (optimizer state_dict step counters)"""
        self.optimizer.note_replayed_steps(self.replays)
        self.replays = 0
