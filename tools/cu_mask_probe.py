"""Does confining the weight-gradient stream to a subset of the CUs help the step?  hipExtStreamCreateWithCUMask through ctypes,
wrapped as a torch ExternalStream and handed to the engine as its side stream.

    python tools/cu_mask_probe.py > profiles/r04_cu_mask.txt
"""
import ctypes
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def run(label, bits, wgs):
    os.environ["CONTOUR_WGRAD_WGS"] = str(wgs)
    task, _ = build_task(256, "bf16", "dsnt-skew")
    task = task.to(dev)
    opt = task.configure_optimizers()["optimizer"]
    img, contour = synthetic_batch(64, 256, 21, seed=1234)
    b = {"img": img.to(dev), "contour": contour.to(dev)}
    eng = task.model.engine
    if bits is not None:
        eng._side = masked_stream(bits)

    def step(i):
        opt.zero_grad(set_to_none=True)
        out = task.training_step(b, i)
        out["loss"].backward()
        opt.step()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(100):
            step(i)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 10)
    print(f"{label:58s} wgrad workgroups {wgs:3d}: " + " / ".join(f"{r:.3f}" for r in res) + " ms per step", flush=True)


ALL = (1 << 256) - 1
every2 = sum(1 << i for i in range(0, 256, 2))
cases = [("no mask", None, 192), ("no mask", None, 128),
         ("side stream: CUs 0-127", (1 << 128) - 1, 128), ("side stream: CUs 0-191", (1 << 192) - 1, 192),
         ("side stream: every second CU", every2, 128), ("side stream: CUs 0-95", (1 << 96) - 1, 96),
         ("no mask", None, 192)]
sel = [int(a) for a in sys.argv[1:]] or range(len(cases))      # one case per process: a masked stream left alive skews the next case
for k in sel:
    run(*cases[k])
