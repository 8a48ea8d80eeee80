"""Data-parallel gradient exchange: bucketed all-reduce over the flat gradient buffers, overlapped with backward.

The reference is single-device; this is the new N-GPU path of BASELINE.json (one process per GPU,
``torch.distributed`` backend "nccl" = RCCL over xGMI).  Minibatches shard over ranks, InstanceNorm is per sample and
the loss is a mean over N*K, so the only exchange is one sum of the gradients per step:

* ``UNet`` / ``ConfidenceNet`` write their gradients into ONE flat float32 buffer per backward; the kernel schedule
  calls ``grad_ready_hook(prefix)`` when a layer's slice is final (decoder first, i.e. from the END of the buffer).
* ``BucketedAllReduce`` turns those notifications into a few large asynchronous all-reduces of contiguous slices,
  launched while the remaining backward kernels run (RCCL uses its own stream; ``async_op=True`` orders the collective
  after everything queued on the compute stream so far).
* the division by the world size is folded into the fused Adam (``grad_scale``).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large buckets (default 32 MiB) keep each link busy; the
168.7 MB fp32 gradient of the 8-stage net is 6 collectives.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
import torch.distributed as dist


def sum_over_ranks(buf: torch.Tensor, group=None, algo: str = "allreduce", async_op: bool = False):
    """In-place SUM of ``buf`` over the ranks through torch.distributed.  ``algo`` "rs_ag": reduce-scatter + all-gather of
    equal shards (what the native cu_comm_* path does by default); backends without a tensor reduce-scatter (gloo, the
    CPU tests) get the same two phases from ``reduce`` to the shard's owner + ``all_gather``."""
    world = dist.get_world_size(group)
    per = buf.numel() // world if algo == "rs_ag" else 0
    if per == 0:
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    head, rank = buf[:per * world], dist.get_rank(group)
    shards = list(head.view(world, per).unbind(0))
    try:
        mine = torch.empty_like(shards[rank])
        dist.reduce_scatter_tensor(mine, head, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(head, mine, group=group)
    except (RuntimeError, NotImplementedError):
        for r, sh in enumerate(shards):                   # phase 1: shard r summed on rank r
            dist.reduce(sh, dst=dist.get_global_rank(group, r) if group is not None else r, op=dist.ReduceOp.SUM, group=group)
        mine = shards[rank].clone()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)      # phase 2: everybody gets every shard
        for sh, g_ in zip(shards, gathered):
            sh.copy_(g_)
    if buf.numel() > per * world:
        dist.all_reduce(buf[per * world:], op=dist.ReduceOp.SUM, group=group)
    return None


class BucketedAllReduce:
    def __init__(self, total: int, bucket_elems: int = 8 * 1024 * 1024, group=None, native=None):
        self.total = total
        self.bucket = bucket_elems
        self.group = group
        self.native = native            # cu_hip.comm.NativeComm: the cu_comm_* C ABI instead of torch.distributed
        # torch.distributed path: "allreduce" (asynchronous, overlapped) or "rs_ag" (CONTOUR_COMM_ALGO; the native path's
        # default algorithm, here for rehearsals and tests -- its torch form is synchronous)
        self.algo = os.environ.get("CONTOUR_COMM_ALGO", "allreduce") if native is None else native.algo
        self.flat: Optional[torch.Tensor] = None
        self._done: List[Tuple[int, int]] = []
        self._frontier = total          # everything in [frontier, total) has been handed to a collective
        self._works = []
        self.launched: List[Tuple[int, int]] = []

    def begin(self, flat: torch.Tensor):
        assert flat.numel() == self.total
        self.flat = flat
        self._done = []
        self._frontier = self.total
        self._works = []
        self.launched = []

    def _contiguous_tail(self) -> int:
        """lowest offset lo such that [lo, frontier) is completely done."""
        lo = self._frontier
        changed = True
        while changed:
            changed = False
            for a, b in self._done:
                if b == lo and a < lo:
                    lo = a
                    changed = True
        return lo

    def _launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        self._done = [(a, b) for a, b in self._done if not (a >= lo and b <= hi)]
        self.launched.append((lo, hi))
        self._frontier = lo
        if self.native is not None:
            self.native.allreduce_async(self.flat[lo:hi])
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if self.algo == "rs_ag":
                sum_over_ranks(self.flat[lo:hi], self.group, "rs_ag")
            else:
                self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True))

    def ready(self, lo: int, hi: int):
        """Mark [lo, hi) final; launch a collective when a bucket's worth is contiguous with the frontier."""
        self._done.append((lo, hi))
        tail = self._contiguous_tail()
        if self._frontier - tail >= self.bucket:
            self._launch(tail, self._frontier)

    def drain(self):
        """Wait for the collectives launched so far and forget the step (an aborted backward: nothing new is launched)."""
        for w in self._works:
            w.wait()
        self._works = []
        if self.native is not None:
            self.native.wait()
        self.flat = None

    def finish(self):
        """Flush what is left (in at most two collectives) and make the current stream wait for all of them."""
        tail = self._contiguous_tail()
        self._launch(tail, self._frontier)
        if self._frontier > 0:              # slices that never reported (e.g. frozen layers): reduce them too
            self._launch(0, self._frontier)
        for w in self._works:
            w.wait()
        self._works = []
        if self.native is not None:
            self.native.wait()


def prefix_ranges(names: Sequence[str], sizes: Sequence[int]) -> Dict[str, Tuple[int, int]]:
    """Map every layer prefix the kernel schedule reports (e.g. 'upsamples.3.conv_block.conv1',
    'upsamples.3.transp_conv', 'output_block') to its contiguous [lo, hi) slice of the flat buffer."""
    out: Dict[str, Tuple[int, int]] = {}
    off = 0
    for n, s in zip(names, sizes):
        parts = n.split(".")
        for cut in range(1, len(parts)):
            p = ".".join(parts[:cut])
            lo, hi = out.get(p, (off, off))
            out[p] = (min(lo, off), max(hi, off + s))
        off += s
    return out


class GradSync:
    """Wires a DSNT task's modules (``model`` and optionally ``skew_block``) to bucketed all-reduces.

    The modules' autograd nodes write every backward's gradients into ONE fresh flat buffer and return views of it;
    when ``p.grad`` is ``None`` beforehand (``zero_grad(set_to_none=True)``, no gradient accumulation) autograd keeps
    those views as ``p.grad``, so reducing the flat buffer IS reducing ``p.grad`` and the reduction can start while the
    backward still runs (overlapped mode).  Whenever that does not hold -- ``zero_grad(set_to_none=False)``, Lightning's
    ``accumulate_grad_batches > 1``, a hook that replaced a gradient -- autograd ADDS the views into older ``p.grad``
    tensors: an in-flight in-place collective would race with that read and the sum would never reach the optimizer.
    ``begin`` detects it (a used parameter already has a gradient) and switches this backward to deferred mode: nothing
    is launched during the backward and ``finish`` packs the accumulated ``p.grad`` into the flat buffer, reduces it and
    writes the sums back.  ``finish`` verifies the aliasing in overlapped mode and refuses to continue if it broke.

    Gradient accumulation over micro-batches: set ``overlap = False`` (the Lightning glue does when
    ``accumulate_grad_batches > 1``) and call ``finish`` once, after the last micro-batch's backward -- an overlapped
    first micro-batch would otherwise be reduced twice."""

    def __init__(self, task, bucket_elems: int = 8 * 1024 * 1024, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.model = task.model
        self.skew = getattr(task, "skew_block", None)
        params = dict(self.model.named_parameters())
        names = list(self.model._used_names)
        sizes = [params[n].numel() for n in names]
        self._used = [params[n] for n in names]
        self.ranges = prefix_ranges(names, sizes)
        self.native = None
        if self.world > 1 and os.environ.get("CONTOUR_COMM", "torch") == "native":
            from .comm import NativeComm
            self.native = NativeComm.create(group=group)
        self.bar = BucketedAllReduce(sum(sizes), bucket_elems, group, self.native)
        self.dry = False                  # True: hooks run, no collective is launched (bench.py's exposed-communication probe)
        self.overlap = True               # False: never reduce during a backward (gradient accumulation)
        self.overlapped = False           # mode of the backward in flight
        self.deferred_steps = 0           # statistics (tests): backwards that could not overlap
        if self.world > 1:      # a single rank has nothing to exchange: the engine then un-prepares all gradients at once
            self.model.engine.grad_ready_hook = self._ready
            self.model.flat_grad_hook = self._begin

    def broadcast_parameters(self):
        """Rank 0's weights everywhere (same start as a single-GPU run)."""
        if self.world == 1:
            return
        flat, _ = self.model.flat_params()
        dist.broadcast(flat, 0, group=self.group)
        if self.skew is not None:
            sflat, _ = self.skew.flat_params()
            dist.broadcast(sflat, 0, group=self.group)

    def _begin(self, flat: torch.Tensor):
        if self.dry:
            return
        if self.overlapped:
            raise RuntimeError("GradSync: a second backward started before finish() consumed the first one; set "
                               "overlap = False when accumulating gradients over micro-batches")
        self.overlapped = self.overlap and all(p.grad is None for p in self._used)
        if self.overlapped:
            self.bar.begin(flat)
        else:
            self.deferred_steps += 1

    def _ready(self, prefix: str):
        if not self.overlapped:
            return
        lo, hi = self.ranges[prefix]
        self.bar.ready(lo, hi)

    def close(self):
        """Release the RCCL communicator of the native path (also released when the object is collected)."""
        if self.native is not None:
            self.native.close()
            self.native = None
            self.bar.native = None

    def abort(self):
        """The backward in flight died (``UNetEngine.backward`` calls this before re-raising): wait for the collectives
        already launched on the dead step's buffer and leave overlapped mode, so the next backward can begin."""
        if self.overlapped:
            try:
                self.bar.drain()
            finally:
                self.overlapped = False

    def _allreduce_now(self, buf: torch.Tensor):
        if self.native is not None:
            self.native.allreduce_async(buf)
            self.native.wait()
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    @staticmethod
    def _aliased(plist, flat) -> bool:
        """every p.grad is the view of ``flat`` at the parameter's offset"""
        if flat is None:
            return False
        ptr = flat.data_ptr()
        for p in plist:
            g = p.grad
            if g is None or g.data_ptr() != ptr or not g.is_contiguous() or g.dtype != flat.dtype:
                return False
            ptr += p.numel() * flat.element_size()
        return ptr == flat.data_ptr() + flat.numel() * flat.element_size()

    def _reduce_packed(self, plist, like: torch.Tensor):
        """pack p.grad (missing gradients count as zero) -> one all-reduce -> write the sums back"""
        plist = [p for p in plist]
        total = sum(p.numel() for p in plist)
        buf = torch.zeros(total, dtype=torch.float32, device=like.device)
        off = 0
        for p in plist:
            if p.grad is not None:
                buf[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        self._allreduce_now(buf)
        off = 0
        for p in plist:
            if p.grad is None:
                p.grad = buf[off:off + p.numel()].view(p.shape)
            else:
                p.grad.copy_(buf[off:off + p.numel()].view(p.shape))
            off += p.numel()

    def finish(self):
        """Call after ``loss.backward()`` (of the LAST micro-batch when accumulating) and before the optimizer step."""
        if self.world == 1 or self.dry:
            return
        ops.pending_wait()                  # the skew head's gradients may come from its own stream
        if self.skew is not None:
            splist = [p for _, p in self.skew.named_parameters()]
            if any(p.grad is not None for p in splist):
                if self._aliased(splist, self.skew.last_flat_grad):
                    self._allreduce_now(self.skew.last_flat_grad)
                else:
                    self._reduce_packed(splist, splist[0])
        if self.overlapped:
            self.bar.finish()
            self.overlapped = False
            if not self._aliased(self._used, self.model.last_flat_grad):
                raise RuntimeError(
                    "GradSync: the gradients were all-reduced in the flat buffer of this backward, but p.grad no longer "
                    "aliases it (a hook or a second backward replaced the gradients): the reduced values would not reach "
                    "the optimizer.  Call finish() right after the backward it belongs to.")
        else:
            self._reduce_packed(self._used, self._used[0])

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world
