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


class NativeComm:
    def __init__(self, handle: int, rank: int, world: int, device: torch.device):
        self._h, self.rank, self.world, self.device = handle, rank, world, device
        self.stream = torch.cuda.Stream(device=device)

    @classmethod
    def create(cls, rank: Optional[int] = None, world: Optional[int] = None, group=None) -> "NativeComm":
        import torch.distributed as dist
        lib = L.load()
        L.require_gpu()
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(lib.cu_comm_unique_id(C.cast(uid, C.c_void_p)), "cu_comm_unique_id")
        if world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        out = C.c_void_p()
        L.check(lib.cu_comm_init(rank, world, C.cast(uid, C.c_void_p), C.byref(out)), "cu_comm_init")
        return cls(out.value, rank, world, torch.device("cuda", torch.cuda.current_device()))

    def allreduce_async(self, buf: torch.Tensor):
        """in-place f32 SUM over the ranks, asynchronous w.r.t. the current (kernel) stream"""
        assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.is_cuda
        ready = torch.cuda.Event()
        ready.record()                                   # everything queued on the kernel stream so far
        self.stream.wait_event(ready)
        L.check(L.load().cu_comm_allreduce_bucket(self._h, buf.data_ptr(), buf.numel(), self.stream.cuda_stream),
                "cu_comm_allreduce_bucket")
        buf.record_stream(self.stream)

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)

    def close(self):
        if self._h:
            self.wait()
            torch.cuda.synchronize()
            L.load().cu_comm_destroy(self._h)
            self._h = None
