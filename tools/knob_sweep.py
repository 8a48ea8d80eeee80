"""One-knob-at-a-time A/B of the whole step on the tuning build (interleaved, two rounds): which library / engine defaults
still hold after a round's changes.  python tools/knob_sweep.py > gpurun_out/knob_sweep.txt"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
LIB = str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so")
CONFIGS = [("baseline", {}),
           ("CONTOUR_WGRAD_WGS=160", {"CONTOUR_WGRAD_WGS": "160"}), ("CONTOUR_WGRAD_WGS=224", {"CONTOUR_WGRAD_WGS": "224"}),
           ("CU_TCONV_VAR=0", {"CU_TCONV_VAR": "0"}), ("CU_CONV_NORING", {"CU_CONV_NORING": "1"}),
           ("CU_PCONV_CAP=1", {"CU_PCONV_CAP": "1"}), ("CU_PCONV_CAP=3", {"CU_PCONV_CAP": "3"}), ("CU_PCONV_RING=3", {"CU_PCONV_RING": "3"}),
           ("CU_PCONV_NO256", {"CU_PCONV_NO256": "1"}), ("CU_CONV_KSPLIT_WGS=256", {"CU_CONV_KSPLIT_WGS": "256"}),
           ("CU_WGRAD_SMALLPX=1024", {"CU_WGRAD_SMALLPX": "1024"}), ("CU_WGRAD_SMALLPX=4096", {"CU_WGRAD_SMALLPX": "4096"}),
           ("CU_CONV_DMA_MINC=128", {"CU_CONV_DMA_MINC": "128"}), ("CU_WGRAD_PCE=6", {"CU_WGRAD_PCE": "6"})]
res = {name: [] for name, _ in CONFIGS}
for rnd in range(2):
    for name, env in CONFIGS:
        e = dict(os.environ, CONTOUR_HIP_LIB=LIB, **env)
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "25", "--warmup", "5", "--no-cpu-baseline",
                              "--no-parity", "--no-roofline"], env=e, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[name].append(json.loads(lines[-1])["ms_per_step"] if lines else float("nan"))
        print(f"{name:28s} round {rnd}: {res[name][-1]:.3f} ms", flush=True)
base = sum(res["baseline"]) / len(res["baseline"])
print("# summary (mean ms, delta vs baseline)")
for name, v in res.items():
    m = sum(v) / len(v)
    print(f"{name:28s} {m:.3f}  {100 * (m - base) / base:+.2f} %")
