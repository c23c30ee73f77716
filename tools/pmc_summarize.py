#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of a bench.py run into the traffic record bench.py reads (profiles/*pmc_traffic*.json).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/WRITE_SIZE -o run -- python3 bench.py ... (same command)
    python3 tools/pmc_summarize.py gpurun_out/pmc profiles/r02_x_pmc_traffic_vitl16_b166.json --workload vit_l16_224 --batch 166

(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; the program follows `--` directly.)
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are reported in KB, and on gfx950 FETCH_SIZE tallies a 128-byte read
request as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md, HBM section) — every kernel here reads 16 B per lane.
The record carries `src_hash` (bench.source_hash() of the tree it was taken from): bench.py ignores it once the kernels change."""
import argparse
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = {
    "gemm": re.compile(r"gemm[35]_kernel"),
    "attention": re.compile(r"attn_(fwd|bwd_dq|bwd_dkv|s3_fwd|g_bwd)"),
    "conv": re.compile(r"conv_(fwd|fwd_strip|wgrad)_kernel"),
}


def read_pass(d, counter):
    files = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}/{counter}")
    per = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                ent = per.setdefault(row["Kernel_Name"], [0.0, 0])
                ent[0] += float(row["Counter_Value"])
                ent[1] += 1
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("out_json")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--command", default="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
    a = ap.parse_args()
    import bench
    fetch, write = read_pass(a.pmc_dir, "FETCH_SIZE"), read_pass(a.pmc_dir, "WRITE_SIZE")
    kernels, fam = {}, {k: dict(fetch_kb=0.0, write_kb=0.0, launches=0) for k in FAMILIES}
    for name in sorted(set(fetch) | set(write)):
        f_sum, f_n = fetch.get(name, [0.0, 0])
        w_sum, w_n = write.get(name, [0.0, 0])
        n = max(f_n, w_n)
        if n == 0:
            continue
        short = re.sub(r"\(.*", "", name)[:120]
        kernels[short] = {"launches_sampled": n, "FETCH_SIZE_KB_avg_per_launch": f_sum / max(f_n, 1), "WRITE_SIZE_KB_avg_per_launch": w_sum / max(w_n, 1),
                          "hbm_bytes_per_launch_corrected": (2 * f_sum / max(f_n, 1) + w_sum / max(w_n, 1)) * 1024}
        for k, rx in FAMILIES.items():
            if rx.search(name):
                # per-launch averages of the two passes are taken separately (the passes may sample a different number of launches)
                fam[k]["fetch_kb"] += f_sum
                fam[k]["write_kb"] += w_sum
                fam[k]["launches"] += n
                fam[k].setdefault("f_n", 0)
                fam[k].setdefault("w_n", 0)
                fam[k]["f_n"] += f_n
                fam[k]["w_n"] += w_n
    families = {}
    for k, v in fam.items():
        if not v["launches"]:
            continue
        fa, wa = v["fetch_kb"] / max(v.get("f_n", 0), 1), v["write_kb"] / max(v.get("w_n", 0), 1)
        families[k] = {"launches_sampled": v["launches"], "FETCH_SIZE_KB_avg_per_launch": fa, "WRITE_SIZE_KB_avg_per_launch": wa,
                       "hbm_bytes_per_launch_corrected": (2 * fa + wa) * 1024, "hbm_bytes_per_launch_uncorrected": (fa + wa) * 1024}
    rec = {"workload": a.workload, "dtype": a.dtype, "per_gpu_batch": a.batch, "src_hash": bench.source_hash(), "command": a.command,
           "note": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KB counters; gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md HBM "
                   "section); 'families' average every launch of a kernel family in the profiled run",
           "families": families, "kernels": kernels}
    with open(a.out_json, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"]) for k, v in families.items()}))


if __name__ == "__main__":
    main()
