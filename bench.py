#!/usr/bin/env python
"""Headline benchmark: train images/sec of task=dsnt-skew, 256x256x1 input, K=21, bf16 compute, fused Adam, on N GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step of the hot path (U-Net forward, skew head, DSNT head + skew-normal NLL, hand-written
backward, RCCL gradient all-reduce for N > 1, fused Adam) on one synthetic minibatch already resident in HBM.
Default: the per-GPU batch is fixed (64, BASELINE.json configs[2] / configs[3] = 512 over 8 GPUs), "scaling": "weak";
--scaling strong fixes the GLOBAL batch (--batch) and splits it over the GPUs.  The eager step runs its weight gradients on
a second HIP stream; up to 16 images per GPU (one GPU) the step is replayed as one hipGraph instead (--graph).

Prints ONE JSON line on rank 0 with the driver's contract plus:
  roofline     : the dominant kernel family (MFMA implicit-GEMM convolution), algorithmic FLOPs / measured launch time
                 (HIP events on the launch stream, in a separate profiled pass after the timed region, with the step on
                 ONE stream so that an event pair times one launch) vs 2.5 PFLOP/s; traffic = HBM bytes per launch from
                 the committed PMC passes (profiles/r02_pmc_hbm_traffic.json).
  cpu_baseline : the CPU oracle (PyTorch-CPU restatement of the reference step, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_DENSE = 2.5e15     # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
PEAK_HBM = 8.0e12


def build_task(size: int, dtype: str, task_name: str):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    n_stages = 8 if size >= 256 else 6
    model_cfg = {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * n_stages,
                 "strides": [[1, 1]] + [[2, 2]] * (n_stages - 1), "patch_size": [256, 256], "drop_block": False,
                 "deep_supervision": False, "compute_dtype": dtype}
    cls = DSNTSkew if task_name == "dsnt-skew" else DSNTAleatoric
    torch.manual_seed(0)
    task = cls(model=model_cfg, optim={"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}, choices={},
               data_params=DataParameters((1, size, size), (21, 2), [0, 1]), psm_path="camus-cont_psm_11_no_std.npy",
               seq_psm_path="camus-cont_sequence_psm_11_no_std.npy", t_a=25, t_e=1, covar=True)
    return task, n_stages


def conv_flops_per_image(n_stages: int, size: int):
    """Algorithmic conv FLOPs of one training step per image: fwd + dgrad + wgrad,
    no dgrad for the first conv (counted from the layer list, cu_hip/flops.py)."""
    from cu_hip.flops import train_step_flops_per_image
    return train_step_flops_per_image([1] + [2] * (n_stages - 1), size)


def cpu_baseline(size: int, n_stages: int, task_name: str, batch: int, steps: int):
    """The oracle's training step (op-for-op the reference on PyTorch-CPU) on the host cores.  The ONLY place where
    bench.py touches oracle/ (as the thing timed beside the product, never as part of it)."""
    from contour_uncertainty.data.synthetic import synthetic_batch
    from oracle.step import OracleTask
    from oracle.unet import UNetSpec
    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() would report the whole host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("BENCH_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    spec = UNetSpec(strides=tuple([1] + [2] * (n_stages - 1)))
    ot = OracleTask(spec, task=task_name, seed=0)
    img, contour = synthetic_batch(batch, size, 21, seed=1234)
    ot.train_step(img, contour)                      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        ot.train_step(img, contour)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{steps} steps of batch {batch} at {size}x{size}, fp32, oracle/step.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU minibatch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--task", default="dsnt-skew", choices=["dsnt-skew", "dsnt-al", "dsnt-al2"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU; strong: --batch images in total, split over the GPUs")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step as one hipGraph (cu_hip.graph.CapturedStep); auto = on for a single GPU with at "
                         "most 16 images per GPU, where the eager step is bound by its ~320 host launches (from 32 images "
                         "on the eager step is faster: its weight-gradient stream runs beside the main one, which a "
                         "replayed graph serialises -- profiles/r02_small_batch_graph.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")     # "gloo" lets a 1-GPU box rehearse the N-rank flow
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"launched {world} ranks for --gpus {args.gpus}"

    from cu_hip import ops
    from cu_hip.ddp import GradSync
    from contour_uncertainty.data.synthetic import synthetic_batch   # SURVEY 8d synthetic inputs

    task_name = "dsnt-al" if args.task == "dsnt-al2" else args.task
    task, n_stages = build_task(args.size, args.dtype, task_name)
    task = task.to(dev)
    sync = GradSync(task)
    sync.broadcast_parameters()
    if args.scaling == "strong":
        assert args.batch % world == 0, "strong scaling: --batch (global) must divide by the number of GPUs"
        per_gpu = args.batch // world
    else:
        per_gpu = args.batch
    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1 and per_gpu <= 16)
    if use_graph:
        task.hparams.optim = dict(task.hparams.optim, capturable=True)
    opt = task.configure_optimizers()["optimizer"]
    opt.grad_scale = sync.grad_scale
    img, contour = synthetic_batch(per_gpu, args.size, 21, seed=1234 + rank)
    batch = {"img": img.to(dev), "contour": contour.to(dev)}

    def eager_step(i):
        opt.zero_grad(set_to_none=True)
        out = task.training_step(batch, i)
        out["loss"].backward()
        sync.finish()
        opt.step()
        return out

    captured = None
    if use_graph:
        from cu_hip.graph import CapturedStep
        try:
            captured = CapturedStep(task, opt, batch, after_backward=sync.finish)
        except Exception as e:      # noqa: BLE001 -- report and measure the eager step instead
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr,
                  flush=True)
            captured = None

    def step(i):
        if captured is not None:
            captured.replay()
            return captured.logs
        return eager_step(i)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = step(i)
    fence()
    if rank == 0:
        print(f"[bench] warm-up done, timing {args.steps} steps ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    fence()
    dt = time.perf_counter() - t0
    loss = float(out["loss"].detach())
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    result = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = per_gpu * world * args.steps / dt
        flops_step_img, _ = conv_flops_per_image(n_stages, args.size)
        result = {
            "metric": f"train images/sec, {args.task} {args.size}x{args.size} {args.dtype}",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"task={args.task} {args.size}x{args.size}x1, K=21, {n_stages}-stage unet2, "
                                   f"batch {per_gpu}/GPU, Adam(lr=1e-3, wd=1e-3)",
                       "per_gpu_batch": per_gpu, "global_batch": per_gpu * world,
                       "launch": "hipGraph replay" if captured is not None else "eager",
                       "parallelism": f"dp{world}", "final_loss": round(loss, 4),
                       "mfma_roofline_frac_whole_step": round(value / world * flops_step_img / PEAK_BF16_DENSE, 4)},
        }

    # ---- roofline of the dominant kernel family: separate profiled pass (events around every launch).  Every rank
    #      runs the two extra steps (they contain collectives); only rank 0 records.
    if not args.no_roofline:
        if rank == 0:
            ops.PROFILE.clear()
            ops.PROFILE_ON[0] = True
        engines = [m.engine for m in task.modules() if hasattr(getattr(m, "engine", None), "side_wgrad")]
        side = [e.side_wgrad for e in engines]
        for e in engines:
            e.side_wgrad = False    # one stream: an event pair around a launch then times that launch alone (with the
                                    # weight-gradient stream running beside it, it would time both streams' kernels)
        for i in range(2):
            eager_step(i)           # per-launch events need the eager launches (a graph replay is one opaque launch)
        torch.cuda.synchronize()
        ops.PROFILE_ON[0] = False
        for e, s_ in zip(engines, side):
            e.side_wgrad = s_
    if rank == 0 and not args.no_roofline:
        fam = {}
        for name, flops, e0, e1, *_ in ops.PROFILE:
            ms_k = e0.elapsed_time(e1)
            f = fam.setdefault(name, [0.0, 0.0, 0])
            f[0] += flops
            f[1] += ms_k
            f[2] += 1
        total_ms = sum(v[1] for v in fam.values())
        dom = max(fam.items(), key=lambda kv: kv[1][1])
        name, (fl, ms_k, cnt) = dom
        achieved = fl / (ms_k * 1e-3) / 1e12
        traffic = None
        pmc = ROOT / "profiles" / "r02_pmc_hbm_traffic.json"     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this bench
        if pmc.exists() and per_gpu == 64 and args.size == 256 and args.dtype == "bf16":
            famrec = json.loads(pmc.read_text())["families"].get(name)
            if famrec:
                traffic = round(famrec["bytes_per_launch"])
        symbols = {"igemm_conv": "igemm_conv_kernel<*> + igemm_conv_dma_kernel<*> + igemm_conv_dma_ring_kernel + pconv_kernel<*> + pconv2_kernel + tconv_kernel<*> (all instantiations; + ksplit_finish_kernel)",
                   "igemm_wgrad": "igemm_wgrad_kernel<*> + igemm_wgrad_dma_kernel<*> (all instantiations)"}
        result["roofline"] = {"bound": "mfma", "kernel": name, "kernel_symbols": symbols.get(name, name),
                              "achieved": round(achieved, 2),
                              "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s",
                              "frac": round(achieved * 1e12 / PEAK_BF16_DENSE, 4), "traffic": traffic,
                              "traffic_source": "profiles/r02_pmc_hbm_traffic.json (HBM bytes per launch, PMC)" if traffic else None,
                              "launches_per_step": cnt // 2, "avg_launch_ms": round(ms_k / cnt, 4),
                              "flops_per_launch": fl / cnt,
                              "family_ms_per_step": {k: round(v[1] / 2, 3) for k, v in fam.items()},
                              "family_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in fam.items()
                                                if v[1] > 0 and v[0] > 0},
                              "profiled_ms_per_step": round(total_ms / 2, 3)}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        print("[bench] timing the CPU oracle baseline ...", file=sys.stderr, flush=True)
        # bounded sample of the same workload: ~15 s of host work (24 steps of batch 4 at 256x256)
        result["cpu_baseline"] = cpu_baseline(args.size, n_stages, task_name, 4 if args.size >= 256 else 8,
                                              24 if args.size >= 256 else 60)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
