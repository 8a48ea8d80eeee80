"""Host side of the cu_comm_* C ABI: one RCCL communicator per process, collectives on a HIP stream of their own.

``NativeComm.create()`` exchanges the 128-byte unique id through the already initialised ``torch.distributed`` group (any
backend: only the id travels that way) and builds the communicator; ``allreduce_async`` orders a bucket behind the
kernel stream with an event (the "bucket ready" signal), runs the RCCL all-reduce on the communication stream, and
``wait`` makes the kernel stream wait for everything issued.  Selected by ``CONTOUR_COMM=native`` in
``cu_hip.ddp.GradSync``; the default is ``torch.distributed``'s "nccl" backend -- the same RCCL, reached through torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L


import os
import weakref


class NativeComm:
    """``algo``: "rs_ag" (default; CONTOUR_COMM_ALGO) = in-place reduce-scatter + all-gather of every bucket -- on xGMI's
    point-to-point links (7 x ~153 GB/s per GPU) the two halves of a ring all-reduce as two collectives, each moving
    (N-1)/N of the bucket over all links -- or "allreduce" (one ncclAllReduce per bucket)."""

    def __init__(self, handle: int, rank: int, world: int, device: torch.device, algo: Optional[str] = None):
        self._h, self.rank, self.world, self.device = handle, rank, world, device
        self.stream = torch.cuda.Stream(device=device)
        self.algo = algo or os.environ.get("CONTOUR_COMM_ALGO", "rs_ag")
        assert self.algo in ("rs_ag", "allreduce"), self.algo
        # never leak the RCCL communicator; the finalizer drains what is still queued first, as close() does (a
        # GradSync dropped, or a process exiting, with collectives in flight must not tear the communicator down under them)
        self._fin = weakref.finalize(self, NativeComm._destroy, handle, self.stream, device)

    @staticmethod
    def _destroy(handle, stream=None, device=None):
        try:
            if stream is not None:
                stream.synchronize()
            if device is not None:
                torch.cuda.synchronize(device)
        except Exception:      # noqa: BLE001 -- interpreter shutdown / a dead context: still release the communicator
            pass
        try:
            L.load().cu_comm_destroy(handle)
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    @classmethod
    def create(cls, rank: Optional[int] = None, world: Optional[int] = None, group=None) -> "NativeComm":
        import torch.distributed as dist
        lib = L.load()
        L.require_gpu()
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:                                    # (world given without rank crashed)
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if not 0 <= rank < world:
            raise L.ContourHipError(f"NativeComm.create: rank {rank} outside world {world}")
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(lib.cu_comm_unique_id(C.cast(uid, C.c_void_p)), "cu_comm_unique_id")
        if world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        out = C.c_void_p()
        # cu_comm_init binds the communicator to the CURRENT device: make that explicit and remember it
        device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(device):
            L.check(lib.cu_comm_init(rank, world, C.cast(uid, C.c_void_p), C.byref(out)), "cu_comm_init")
        return cls(out.value, rank, world, device)

    def allreduce_async(self, buf: torch.Tensor):
        """in-place f32 SUM over the ranks, asynchronous w.r.t. the current (kernel) stream"""
        assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.is_cuda
        if buf.device != self.device:
            raise L.ContourHipError(f"bucket on {buf.device}, communicator bound to {self.device}")
        ready = torch.cuda.Event()
        ready.record()                                   # everything queued on the kernel stream so far
        self.stream.wait_event(ready)
        lib, st = L.load(), self.stream.cuda_stream
        n, w = buf.numel(), self.world
        per = n // w if self.algo == "rs_ag" else 0
        if per > 0:
            # in place: rank r's shard of the sum lands in buf[r*per : (r+1)*per], then every rank gathers all shards
            mine = buf.data_ptr() + self.rank * per * 4
            L.check(lib.cu_comm_reduce_scatter_bucket(self._h, buf.data_ptr(), mine, per, st), "cu_comm_reduce_scatter_bucket")
            L.check(lib.cu_comm_allgather_bucket(self._h, mine, buf.data_ptr(), per, st), "cu_comm_allgather_bucket")
        tail = n - per * w
        if tail:                                         # the few elements a bucket does not divide into (or "allreduce")
            L.check(lib.cu_comm_allreduce_bucket(self._h, buf.data_ptr() + per * w * 4, tail, st), "cu_comm_allreduce_bucket")
        buf.record_stream(self.stream)

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)

    def close(self):
        if self._h:
            self.wait()
            torch.cuda.synchronize()
            self._fin()            # cu_comm_destroy, once
            self._h = None
