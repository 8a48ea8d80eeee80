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
                 ONE stream so that an event pair times one launch) vs 2.5 PFLOP/s, from ALGORITHMIC FLOPs; traffic = HBM
                 bytes per launch from the newest committed PMC passes (profiles/r0N_pmc_hbm_traffic.json; a constant).
  parity       : f32 mu / Sigma / alpha vs the reference's own outputs and bf16-vs-f32 NLL after 20 steps (smoke size).
  multi_gpu    : N > 1: world / devices, parameter checksums of all ranks, exposed communication time.
  cpu_baseline : the CPU oracle (PyTorch-CPU restatement of the reference step, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import contextlib
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


def build_task(size: int, dtype: str, task_name: str, backbone: str = "unet2"):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    n_stages = 8 if size >= 256 else 6
    model_cfg = {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * n_stages,
                 "strides": [[1, 1]] + [[2, 2]] * (n_stages - 1), "patch_size": [256, 256], "drop_block": False,
                 "deep_supervision": False, "compute_dtype": dtype}
    if backbone == "vital":     # `task/model=unet`: the `vital` BatchNorm U-Net (north_star's second backbone; secondary number)
        model_cfg = {"_target_": "contour_uncertainty.models.vital.unet.UNet", "init_channels": 32, "use_batchnorm": True,
                     "bilinear": False, "dropout": 0.0, "compute_dtype": dtype, "drop_block": False}
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


def cpu_baseline(size: int, n_stages: int, task_name: str, runs=((4, 5), (32, 2))):
    """The oracle's training step (op-for-op the reference on PyTorch-CPU) on the host cores.  The ONLY place where
    bench.py touches oracle/ (as the thing timed beside the product, never as part of it).
    ``runs`` = (batch, timed steps) pairs: SURVEY.md 8d asks for N = 4 and N = 32; one warm-up step each, bounded to ~25 s of
    host work in total (5 steps of batch 4 + 2 steps of batch 32 at 256x256)."""
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
    by_batch, samples = {}, []
    for batch, steps in runs:
        ot = OracleTask(spec, task=task_name, seed=0)
        img, contour = synthetic_batch(batch, size, 21, seed=1234)
        ot.train_step(img, contour)                      # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            ot.train_step(img, contour)
        dt = time.perf_counter() - t0
        by_batch[str(batch)] = round(batch * steps / dt, 3)
        samples.append(f"{steps} steps of batch {batch}")
        del ot
    best = max(by_batch.values())
    return {"value": best, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "images_per_s_by_batch": by_batch,
            "sample": " + ".join(samples) + f" at {size}x{size}, fp32, oracle/step.py (1 warm-up step each); value = the faster batch"}


def parity_report(dev, steps: int = 20):
    """Parity checks printed WITH the number, at the smoke size (6 stages, 64x64, batch 2,
    the seeded weights and inputs the committed golden vectors were made with -- product code only, no oracle):
      f32 : max relative error of mu / Sigma / alpha of ``predict_on_batch`` (f32 parity mode) vs the REFERENCE's own outputs
            (tests/golden/train_step.npz, written by oracle/make_golden.py from the imported reference); north_star bound 1e-4.
      bf16: contour NLL (the `loss` log) of the bf16 production mode vs the f32 parity mode after `steps` identical Adam
            steps on one fixed batch, same initial weights."""
    import numpy as np
    from contour_uncertainty.data.synthetic import synthetic_batch
    from contour_uncertainty.data.synthetic.weights import seeded_confidence_state, seeded_unet_state
    gold = ROOT / "tests" / "golden" / "train_step.npz"
    if not gold.exists():
        return None
    g = np.load(gold)
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    batch = {"img": img.to(dev), "contour": contour.to(dev)}
    out = {}
    finals = {}
    for dtype in ("f32", "bf16"):
        task, _ = build_task(64, dtype, "dsnt-skew")
        gen = torch.Generator().manual_seed(0)
        task.model.load_state_dict(seeded_unet_state(task.model, gen), strict=True)
        task.skew_block.load_state_dict(seeded_confidence_state(task.skew_block, gen), strict=True)
        task = task.to(dev)
        if dtype == "f32":
            mu, sigma, alpha = task.predict_on_batch(batch["img"], task.model)[:3]
            for name, got, key in (("mu", mu, "dsnt-skew_mu0"), ("sigma", sigma, "dsnt-skew_sigma0"),
                                   ("alpha", alpha, "dsnt-skew_alpha0_predict")):
                ref = torch.from_numpy(g[key])
                out[f"f32_{name}_max_rel_vs_reference"] = float(f"{float((got.cpu() - ref).abs().max() / ref.abs().max()):.3e}")
        opt = task.configure_optimizers()["optimizer"]
        for i in range(steps):
            opt.zero_grad(set_to_none=True)
            o = task.training_step(batch, i)
            o["loss"].backward()
            opt.step()
        with torch.no_grad():
            finals[dtype] = float(task._shared_step(batch, 0)["loss"])
    out["reference_bound"] = 1e-4
    out[f"nll_f32_after_{steps}_steps"] = round(finals["f32"], 5)
    out[f"nll_bf16_after_{steps}_steps"] = round(finals["bf16"], 5)
    out["nll_bf16_minus_f32"] = round(finals["bf16"] - finals["f32"], 5)
    out["sample"] = "dsnt-skew, 6-stage unet2, 64x64, batch 2, seed 0 weights / seed 1234 inputs (tests/golden/train_step.npz)"
    return out


def run_c5(args, world: int, rank: int, dev):
    """BASELINE config c5: uncertainty-propagation inference, `--samples` Monte-Carlo contours per frame, frames sharded over the
    ranks (no collective inside the path: SURVEY.md 8e).  A step = one pass over `--frames` frames per GPU: predicted (mu, Sigma,
    alpha) already on the device -> (frames, samples, 21, 2) sampled contours.  `value` = frames/s of the skew-normal PSM sampler
    (the dsnt-skew task's); the Gaussian sampler and the samples -> masks -> entropy map pipeline are reported beside it.
    Algorithmic bytes: the sampled contours written once (samples x 21 x 2 x 4 B per frame) + the per-frame inputs."""
    import numpy as np
    from cu_hip import ops
    from contour_uncertainty.sampler.posterior_shape_model.psm import PosteriorShapeModelSampler
    from contour_uncertainty.sampler.posterior_shape_model.psm_skew import SkewPosteriorShapeModelSampler
    G = ROOT / "tests" / "golden"
    psm_path = G / "camus-cont_psm_11_no_std.npz"
    psm = dict(np.load(psm_path))
    F, NS = args.frames, args.samples
    g = torch.Generator().manual_seed(100 + rank)
    idx = torch.randint(0, psm["X_val"].shape[0], (F,), generator=g)
    mu = torch.stack([torch.tensor(psm["X_val"][i] + psm["scaler_mean"]).float().reshape(21, 2) for i in idx.tolist()])
    a = torch.randn(F, 21, 2, 2, generator=g)
    cov = a @ a.transpose(-1, -2) * 6.0 + torch.eye(2) * 2.0
    alpha = torch.randn(F, 21, 2, generator=g) * 2.0
    mu, cov, alpha = mu.to(dev), cov.to(dev), alpha.to(dev)
    gs, sk = PosteriorShapeModelSampler(psm_path), SkewPosteriorShapeModelSampler(psm_path)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for i in range(max(args.warmup, 2)):
            fn(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn(i)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def pipeline(sampler, *extra, seed=0):
        c = sampler.sample_batch(mu, cov, *extra, n=NS, seed=seed)
        packed, _ = ops.contour_masks(c.reshape(F * NS, 21, 2), 256, 256, round_landmarks=True, as_bytes=False)
        return ops.mask_entropy(packed, F, 256)

    dt_skew = timed(lambda i: sk.sample_batch(mu, cov, alpha, n=NS, seed=i))
    dt_gauss = timed(lambda i: gs.sample_batch(mu, cov, n=NS, seed=i))
    dt_pipe = timed(lambda i: pipeline(sk, alpha, seed=i))
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    per_frame_bytes = NS * 21 * 2 * 4 + 21 * (2 + 4 + 2) * 4
    fps = lambda dt: F * world * args.steps / dt          # noqa: E731
    achieved = per_frame_bytes * F * args.steps / dt_skew / 1e9
    result = {
        "metric": f"MC contour sampling frames/sec, {NS} samples/frame, skew-normal PSM sampler (BASELINE config c5)",
        "value": round(fps(dt_skew), 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2),
        "ms_per_step": round(dt_skew / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f64 per-frame PSM algebra)", "data": "synthetic",
        "config": {"workload": f"c5: {F} frames/GPU x {NS} samples, K=21, 256x256 grid, PSM camus-cont_psm_11_no_std",
                   "parallelism": f"frames sharded over {world} rank(s), no collective",
                   "gauss_psm_frames_per_s": round(fps(dt_gauss), 1), "gauss_ms_per_step": round(dt_gauss / args.steps * 1e3, 3),
                   "skew_psm_plus_masks_plus_entropy_frames_per_s": round(fps(dt_pipe), 1),
                   "pipeline_ms_per_step": round(dt_pipe / args.steps * 1e3, 3)},
        "roofline": {"bound": "hbm", "kernel": "psm_skew_kernel", "achieved": round(achieved, 2), "peak": PEAK_HBM / 1e9,
                     "unit": "GB/s", "frac": round(achieved * 1e9 / PEAK_HBM, 5), "traffic": None,
                     "note": "algorithmic bytes = sampled contours written once + per-frame inputs; the kernel evaluates a skew-normal "
                             "pdf x conditional Gaussian on up to 256^2 grid cells per drawn point (exp / erf VALU work), so it sits far "
                             "below the HBM roof by construction: transcendental-throughput-bound, not bandwidth-bound"},
    }
    if not args.no_cpu_baseline:
        from oracle import sampler as S
        torch.set_num_threads(16)
        osk = S.SkewPSMSamplerOracle(psm)
        e3, u = torch.randn(1, 4, 21, 3), torch.rand(1, 4, 21)
        t0 = time.perf_counter()
        osk(mu[:1].cpu(), cov[:1].cpu(), alpha[:1].cpu(), 4, e3, u)
        dtc = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(1 / (dtc / 4 * NS), 5), "unit": "frames/s", "cores": torch.get_num_threads(),
                                  "kind": "port", "sample": f"1 frame x 4 samples of the skew sampler oracle (oracle/sampler.py), "
                                                            f"scaled to {NS} samples per frame"}
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY.md 8d: 100)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed warm-up steps (SURVEY.md 8d: 20)")
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
    ap.add_argument("--backbone", default="unet2", choices=["unet2", "vital"],
                    help="unet2 = the headline network; vital = `task/model=unet` (BatchNorm U-Net, dsnt-al only): a secondary "
                         "number, no roofline / CPU baseline / parity legs")
    ap.add_argument("--workload", default="train", choices=["train", "c5"],
                    help="train = the headline training step; c5 = BASELINE config c5: Monte-Carlo contour sampling, frames/s at "
                         "--samples samples per frame, frames sharded over the GPUs (secondary metric)")
    ap.add_argument("--frames", type=int, default=64, help="c5: frames per GPU and pass")
    ap.add_argument("--samples", type=int, default=1024, help="c5: contour samples per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-comm-probe", action="store_true",
                    help="N > 1: skip the extra timed pass without gradient exchange (exposed_comm_ms)")
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

    if args.workload == "c5":
        return run_c5(args, world, rank, dev)

    from cu_hip import ops
    from cu_hip.ddp import GradSync
    from contour_uncertainty.data.synthetic import synthetic_batch   # SURVEY 8d synthetic inputs

    task_name = "dsnt-al" if args.task == "dsnt-al2" else args.task
    if args.backbone == "vital":
        assert task_name == "dsnt-al", "the vital U-Net has no bottleneck output: --task dsnt-al"
        args.no_roofline = args.no_cpu_baseline = args.no_parity = True
    task, n_stages = build_task(args.size, args.dtype, task_name, args.backbone)
    task = task.to(dev)
    if args.backbone == "vital":
        assert world == 1, "the BatchNorm U-Net has no bucketed exchange: one GPU"

        class _NoSync:      # single rank: nothing to exchange
            grad_scale, native = 1.0, None

            def broadcast_parameters(self): pass
            def finish(self): pass
        sync = _NoSync()
    else:
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

    # experiment knob (profiles/r04_stream_priority.txt): run the step's main stream at another HIP stream priority
    main_ctx = contextlib.nullcontext()
    if os.environ.get("CONTOUR_MAIN_PRIORITY"):
        main_ctx = torch.cuda.stream(torch.cuda.Stream(dev, priority=int(os.environ["CONTOUR_MAIN_PRIORITY"])))
    with main_ctx:
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

    # ---- N > 1: make the run diagnosable: the process group really has N ranks on N devices, every
    #      rank holds the same parameters after the timed steps, and how much of the step the gradient exchange costs
    multi = None
    if world > 1:
        assert dist.get_world_size() == args.gpus and dist.get_backend() == backend
        flat, _ = task.model.flat_params()
        cs = torch.stack([flat.double().sum(), flat.double().abs().sum()])
        gathered = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        same = all(bool(torch.equal(gathered[0], x)) for x in gathered)
        devs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([torch.cuda.current_device()], device=dev))
        multi = {"backend": backend, "world": dist.get_world_size(), "devices": [int(x) for x in devs],
                 "param_checksums_equal": same, "param_checksum": [float(x) for x in gathered[0]],
                 "comm": "native cu_comm_* (RCCL)" if sync.native is not None else f"torch.distributed {backend}"}
        if backend == "nccl" and not args.no_comm_probe:
            # the two exchange paths agree: one gradient-sized buffer, different on every rank, summed by
            # torch.distributed's all-reduce and by the C-ABI path (cu_comm_*: reduce-scatter + all-gather by default).  The two
            # algorithms add in different orders, so the sums are compared to f32 rounding, and a failure is REPORTED, never
            # raised: the timed numbers above stand on their own
            try:
                from cu_hip.comm import NativeComm
                nc = sync.native if sync.native is not None else NativeComm.create()
                gen = torch.Generator(device=dev).manual_seed(7 + rank)
                buf = torch.randn(1 << 22, device=dev, generator=gen)
                a, b = buf.clone(), buf.clone()
                dist.all_reduce(a, op=dist.ReduceOp.SUM)
                nc.allreduce_async(b)
                nc.wait()
                torch.cuda.synchronize()
                rel = float((a - b).abs().max() / a.abs().max())
                multi["comm_crosscheck"] = {"native_vs_torch_max_rel_diff": rel, "ok": rel < 1e-5, "elements": buf.numel(),
                                            "native_algo": nc.algo}
                if sync.native is None:
                    nc.close()
            except Exception as e:      # noqa: BLE001
                multi["comm_crosscheck"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_comm_probe and captured is None:
            # the same K steps with the collectives replaced by nothing (the ranks then drift apart: the result is
            # discarded; this is the last thing the model is used for): step time - this = exposed communication
            sync.dry = True
            for i in range(2):
                step(i)
            fence()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            fence()
            dt_dry = time.perf_counter() - t1
            sync.dry = False
            t = torch.tensor([dt_dry], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            multi["ms_per_step_without_exchange"] = round(float(t) / args.steps * 1e3, 3)
            multi["exposed_comm_ms"] = round(dt / args.steps * 1e3 - float(t) / args.steps * 1e3, 3)

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
            "config": {"workload": f"task={args.task} {args.size}x{args.size}x1, K=21, "
                                   + (f"{n_stages}-stage unet2, " if args.backbone == "unet2" else "vital U-Net (BatchNorm), ") +
                                   f"batch {per_gpu}/GPU, Adam(lr=1e-3, wd=1e-3)",
                       "per_gpu_batch": per_gpu, "global_batch": per_gpu * world,
                       "launch": "hipGraph replay" if captured is not None else "eager",
                       "parallelism": f"dp{world}", "final_loss": round(loss, 4),
                       "mfma_roofline_frac_whole_step": round(value / world * flops_step_img / PEAK_BF16_DENSE, 4)},
        }
        if args.backbone != "unet2":
            result["metric"] += " (vital U-Net backbone: secondary)"
            result["config"].pop("mfma_roofline_frac_whole_step")
        if multi is not None:
            result["multi_gpu"] = multi

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
        for name, flops, e0, e1, _note, _bytes, xflops in ops.PROFILE:
            ms_k = e0.elapsed_time(e1)
            f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
            f[0] += flops              # ALGORITHMIC FLOPs (cu_hip/ops.py: existing (tap, parity) pairs, true class count)
            f[1] += ms_k
            f[2] += 1
            f[3] += xflops             # executed (padding and absent taps included): reported beside, never in `frac`
        total_ms = sum(v[1] for v in fam.values())
        dom = max(fam.items(), key=lambda kv: kv[1][1])
        name, (fl, ms_k, cnt, xfl) = dom
        achieved = fl / (ms_k * 1e-3) / 1e12
        traffic, pmc_used = None, None
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of this very command (FETCH_SIZE x 2 + WRITE_SIZE,
        # MI355X_MICROARCH.md HBM section), committed under profiles/: a COMMITTED constant of the newest such file, not a
        # measurement of this run (traffic_source says which)
        for cand in ("r04_pmc_hbm_traffic.json", "r03b_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json"):
            pmc = ROOT / "profiles" / cand
            if pmc.exists() and per_gpu == 64 and args.size == 256 and args.dtype == "bf16":
                famrec = json.loads(pmc.read_text())["families"].get(name)
                if famrec:
                    traffic, pmc_used = round(famrec["bytes_per_launch"]), cand
                    break
        symbols = {"igemm_conv": "igemm_conv_kernel<*> + igemm_conv_dma_kernel<*> + igemm_conv_dma_ring_kernel + pconv_kernel<*> + pconv2_kernel + tconv_kernel<*> (all instantiations; + ksplit_finish_kernel)",
                   "igemm_wgrad": "igemm_wgrad_kernel<*> + igemm_wgrad_dma_kernel<*> (all instantiations)"}
        result["roofline"] = {"bound": "mfma", "kernel": name, "kernel_symbols": symbols.get(name, name),
                              "achieved": round(achieved, 2),
                              "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s",
                              "frac": round(achieved * 1e12 / PEAK_BF16_DENSE, 4), "traffic": traffic,
                              "traffic_source": f"committed: profiles/{pmc_used} (HBM bytes per launch from separate --pmc passes "
                                                f"of this command; not measured in this run)" if traffic else None,
                              "flops": "algorithmic (existing taps, true class count)",
                              "executed_tflops": round(xfl / (ms_k * 1e-3) / 1e12, 2),
                              "launches_per_step": cnt // 2, "avg_launch_ms": round(ms_k / cnt, 4),
                              "flops_per_launch": fl / cnt,
                              "family_ms_per_step": {k: round(v[1] / 2, 3) for k, v in fam.items()},
                              "family_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in fam.items()
                                                if v[1] > 0 and v[0] > 0},
                              "mfma_ms_per_step": round(sum(v[1] for v in fam.values() if v[0] > 0) / 2, 3),
                              "mfma_algorithmic_tflop_per_step": round(sum(v[0] for v in fam.values()) / 2 / 1e12, 4),
                              "profiled_ms_per_step": round(total_ms / 2, 3)}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_parity:
        print("[bench] parity checks at the smoke size ...", file=sys.stderr, flush=True)
        result["parity"] = parity_report(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        print("[bench] timing the CPU oracle baseline ...", file=sys.stderr, flush=True)
        # bounded sample of the same workload at N = 4 and N = 32: ~25 s of host work at 256x256
        result["cpu_baseline"] = cpu_baseline(args.size, n_stages, task_name,
                                              ((4, 5), (32, 2)) if args.size >= 256 else ((4, 20), (32, 5)))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
