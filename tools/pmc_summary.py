"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/rNN_pmc_hbm_traffic.json.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <steps incl. warm-up> <out.json>

FETCH_SIZE is in KiB and, on gfx950, reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section):
bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is in KiB: bytes = WRITE_SIZE * 1024.
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

FAMILIES = [("head_fwd", "dsnt_head"), ("head_bwd", "dsnt_head"), ("c1_bwd", "conv_c1"), ("igemm_conv", "igemm_conv"), ("pconv_kernel", "igemm_conv"), ("pconv2_kernel", "igemm_conv"), ("tconv_kernel", "igemm_conv"), ("ksplit_finish", "igemm_conv"),
            ("igemm_wgrad", "igemm_wgrad"), ("gemm_tn_kernel", "igemm_wgrad"), ("parts_", "weight_prep"), ("fwd_resident", "instnorm_apply"), ("stats_", "instnorm_stats"),
            ("apply_kernel", "instnorm_apply"), ("bwd_", "instnorm_bwd"), ("act_bwd", "instnorm_bwd"),
            ("weight_prep", "weight_prep"), ("grad_unprep", "weight_prep"), ("adam", "adam"), ("conv_c1", "conv_c1"),
            ("head_fwd", "dsnt_head"), ("head_bwd", "dsnt_head"), ("c1_bwd", "conv_c1"), ("c1_moments", "conv_c1"), ("c1_stats", "conv_c1"),
            ("dsnt", "dsnt_head"), ("nll", "dsnt_head")]


def family(name: str) -> str:
    for key, fam in FAMILIES:
        if key in name and not (key == "apply_kernel" and "bwd" in name):
            return fam
    return "other"


def collect(d: Path, counter: str):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in d.rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                fam = family(row["Kernel_Name"])
                tot[fam] += float(row["Counter_Value"])
                cnt[fam] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, steps, out = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4])
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    fams = {}
    for fam in sorted(set(ft) | set(wt)):
        launches = max(fc.get(fam, 0), wc.get(fam, 0)) / steps
        fb, wb = ft.get(fam, 0.0) * 1024 * 2 / steps, wt.get(fam, 0.0) * 1024 / steps
        fams[fam] = {"launches_per_step": launches, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
                     "bytes_per_launch": (fb + wb) / launches if launches else 0.0}
    total = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in fams.values())
    out.write_text(json.dumps({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py (batch 64, bf16), "
                  "summarised by tools/pmc_summary.py",
        "correction": "FETCH_SIZE * 1024 * 2 (gfx950 reports half the bytes of wide coalesced reads, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE * 1024",
        "note": "memory-side (fabric) requests: Infinity-Cache hits are counted; per-step figures include the roofline "
                "pass's share of warm-up launches (divided by the number of steps run)",
        "total_bytes_per_step": total, "families": fams}, indent=1))
    print(f"total {total / 1e9:.1f} GB/step;", {k: round((v["fetch_bytes_per_step"] + v["write_bytes_per_step"]) / 1e9, 2) for k, v in fams.items()})


if __name__ == "__main__":
    main()
