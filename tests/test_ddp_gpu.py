"""N > 1 on real kernels: two ranks share the single GPU of the test box (gloo carries the collective; on a multi-GPU
node the same code runs over RCCL).  The 2-rank step on a split batch must reproduce the 1-rank step on the whole
batch: sum-all-reduce / world == gradient of the global mean."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(kind="dsnt-skew"):
    for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from test_model_gpu import make_task
    from oracle import unet as OU
    task = make_task(kind, 6, 64, "f32")
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
    task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
    return task.to("cuda:0")


def _step(task, img, contour, sync):
    opt = task.configure_optimizers()["optimizer"]
    out = task.training_step({"img": img.cuda(), "contour": contour.cuda()}, 0)
    out["loss"].backward()
    sync.finish()
    opt.step(grad_scale=sync.grad_scale)
    torch.cuda.synchronize()
    flat, _ = task.model.flat_params()
    sflat, _ = task.skew_block.flat_params()
    return float(out["loss"]), flat.detach().cpu().clone(), sflat.detach().cpu().clone()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        task = _build()
        from cu_hip.ddp import GradSync
        from oracle.step import synthetic_batch
        sync = GradSync(task, bucket_elems=1 << 20)
        sync.broadcast_parameters()
        img, contour = synthetic_batch(4, 64, 21, seed=77)
        half = slice(2 * rank, 2 * rank + 2)
        loss, flat, sflat = _step(task, img[half], contour[half], sync)
        ret[rank] = (loss, flat, sflat, len(sync.bar.launched))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_rank_step():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    task = _build()
    from cu_hip.ddp import GradSync
    from oracle.step import synthetic_batch
    img, contour = synthetic_batch(4, 64, 21, seed=77)
    loss, flat, sflat = _step(task, img, contour, GradSync(task))
    l0, f0, s0, nb0 = ret[0]
    l1, f1, s1, nb1 = ret[1]
    assert nb0 >= 3 and nb0 == nb1, "expected several overlapped buckets"
    assert torch.equal(f0, f1) and torch.equal(s0, s1), "ranks diverged"
    assert abs(0.5 * (l0 + l1) - loss) < 1e-5 * abs(loss)
    # Adam moves every weight by ~lr: compare the moves (noise-level gradients excepted, see test_model_gpu)
    diff = (f0 - flat).abs()
    assert float((diff > 5e-5).float().mean()) < 0.02 and float(diff.max()) <= 2.2e-3
    diff = (s0 - sflat).abs()
    assert float((diff > 5e-5).float().mean()) < 0.02 and float(diff.max()) <= 2.2e-3


def test_native_comm_single_rank():
    """cu_comm_* (RCCL bound by dlopen inside libcontour_hip.so): a one-rank communicator on the test box -- unique id,
    init, all-reduce / reduce-scatter / all-gather of a bucket on the communication stream ordered behind the kernel
    stream, destroy.  (More ranks need more GPUs: RCCL refuses two ranks on one device.)"""
    import ctypes as C
    from cu_hip import lib as L
    from cu_hip.comm import NativeComm
    comm = NativeComm.create(rank=0, world=1)
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * 0.5
    ref = x.clone()
    y = x * 2                         # queued on the kernel stream before the bucket is handed over
    comm.allreduce_async(y)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(y, ref * 2)    # SUM over one rank
    h = L.load()
    out = torch.empty_like(x)
    assert h.cu_comm_reduce_scatter_bucket(comm._h, x.data_ptr(), out.data_ptr(), x.numel(), comm.stream.cuda_stream) == 0
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    out.zero_()
    assert h.cu_comm_allgather_bucket(comm._h, x.data_ptr(), out.data_ptr(), x.numel(), comm.stream.cuda_stream) == 0
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert h.cu_comm_allreduce_bucket(None, None, 0, None) == -22
    comm.close()
