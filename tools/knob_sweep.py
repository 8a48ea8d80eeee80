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
           ("CU_CONV_DMA_MINC=128", {"CU_CONV_DMA_MINC": "128"}), ("CU_WGRAD_PCE=6", {"CU_WGRAD_PCE": "6"}),
           # round 4's switches, each against its default
           ("CONTOUR_PARAM_PARTS=0", {"CONTOUR_PARAM_PARTS": "0"}), ("CONTOUR_PREP_OVERLAP=0", {"CONTOUR_PREP_OVERLAP": "0"}),
           ("CONTOUR_C1_BWD_MAIN=0", {"CONTOUR_C1_BWD_MAIN": "0"}), ("CU_NORM_SMALL_RES_MASK=3", {"CU_NORM_SMALL_RES_MASK": "3"}),
           ("CU_NORM_SMALL_RES_MASK=1", {"CU_NORM_SMALL_RES_MASK": "1"}), ("CU_CONV_MF16=1", {"CU_CONV_MF16": "1"}),
           ("CU_WGRAD_TWO=1", {"CU_WGRAD_TWO": "1"}), ("CU_CONV_KSPLIT_TARGET=512", {"CU_CONV_KSPLIT_TARGET": "512"}),
           ("CONTOUR_WGRAD_LAG=1", {"CONTOUR_WGRAD_LAG": "1"}), ("CONTOUR_WGRAD_WGS=256", {"CONTOUR_WGRAD_WGS": "256"})]
ROUNDS = int(os.environ.get("KNOB_ROUNDS", "3"))
if os.environ.get("KNOB_SET") == "2":      # the feature switches and the remaining library knobs
    names = ["CONTOUR_FUSED_HEAD=0", "CONTOUR_FIRST_NO_Z=0", "CONTOUR_FIRST_FUSED=0", "CONTOUR_FUSED_NORM_BWD=0", "CONTOUR_FUSED_STATS=0",
             "CONTOUR_SMALL_NORM=0", "CONTOUR_SMALL_NORM_PX=16", "CONTOUR_SKEW_SIDE=0", "CU_CONV_RING2_WIDE=0", "CU_CONV_NBMAX=2",
             "CU_HF_WGS=256", "CU_HF_WGS=1024", "CU_WGRAD_NW=4", "CU_WGRAD_PC=0", "CU_WGRAD_N128=1", "CU_WGRAD_DPCE=6",
             "CU_CONV_NO_NORMFIN8=1", "CU_CONV_NORING2=1", "CU_CONV_NO_NARROW=1", "CU_PCONV_S2FWD=1", "CU_WGRAD_NODEINT=1",
             "CU_WGRAD_NOGEMMTN=1", "CU_CONV_NO_DMA_STATS=1", "CU_CONV_S2_RAGGED_OFF=1", "CU_CONV_KSPLIT_MAX=4", "CU_WGRAD_PCF=70"]
    CONFIGS = [("baseline", {})] + [(n, dict([n.split("=")])) for n in names]
    res = {name: [] for name, _ in CONFIGS}
res = {name: [] for name, _ in CONFIGS}
for rnd in range(ROUNDS):
    for name, env in CONFIGS:
        e = dict(os.environ, CONTOUR_HIP_LIB=LIB, **env)
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "100", "--warmup", "20", "--no-cpu-baseline",
                              "--no-parity", "--no-roofline"], env=e, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[name].append(json.loads(lines[-1])["ms_per_step"] if lines else float("nan"))
        print(f"{name:28s} round {rnd}: {res[name][-1]:.3f} ms", flush=True)
base = sum(res["baseline"]) / len(res["baseline"])
print("# summary (mean ms, delta vs baseline)")
for name, v in res.items():
    m = sum(v) / len(v)
    print(f"{name:28s} {m:.3f}  {100 * (m - base) / base:+.2f} %")
