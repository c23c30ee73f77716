#!/usr/bin/env python3
"""Summaries of a tools/collect_profiles.sh run for profiles/:  python tools/prof_summarize.py gpurun_out/prof/<tag> profiles/r02_<tag>
writes  <prefix>_kernel_stats_<workload>.txt, <prefix>_pmc_traffic_<workload>.json (bench.py reads it), <prefix>_pmc_sq_<workload>.txt and
copies the bench lines."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, prefix = sys.argv[1], sys.argv[2]
SHORT = {"vit_l16_224": ("vitl16", 665), "vit_b16_224": ("vitb16", 1330), "vit_l16_adaptive196": ("vitl16_adaptive196", 665), "mae_vit_l16_224": ("mae_vitl16", 1002), "unetr_enc_512x512x128": ("unetr_enc", 2),
         "unetr_512x512x128": ("unetr", 2)}
FAM = {"gemm": re.compile(r"gemm[35]_kernel"), "attention": re.compile(r"attn_(fwd|bwd_dq|bwd_dkv|s3_fwd|g_bwd|delta)"),
       "conv": re.compile(r"conv_(fwd|fwd_strip|wgrad)_kernel")}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def stats(wl):
    fs = glob.glob(os.path.join(src, f"stats_{wl}", "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        return
    rows = list(csv.DictReader(open(fs[0], newline="")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "mfma_probe" not in r["Name"])
    out = [f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps 4 --warmup 2 --no-cpu-baseline   (6 steps in the trace; "
           f"mfma_probe launches of the clock probe excluded from the percentages)",
           f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}"]
    fam_ns = {k: 0.0 for k in FAM}
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if "mfma_probe" in r["Name"]:
            continue
        ns = float(r["TotalDurationNs"])
        for k, rx in FAM.items():
            if rx.search(r["Name"]):
                fam_ns[k] += ns
        if ns / tot < 0.0015:
            continue
        out.append(f"{short(r['Name']):112s} {int(r['Calls']):7d} {ns / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} {100 * ns / tot:6.2f}")
    out.append(f"GPU busy {tot / 6e6:.2f} ms per step; " + ", ".join(f"{k} family {100 * v / tot:.1f} %" for k, v in fam_ns.items()))
    open(f"{prefix}_kernel_stats_{SHORT[wl][0]}.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


def counters(d):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            e = per.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0.0, 0, 0.0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
            e[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return per


def sq(wl):
    a, b = counters(os.path.join(src, f"sq_{wl}", "wave")), counters(os.path.join(src, f"sq_{wl}", "mfma"))
    if not a or not b:
        return
    out = [f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --workload {wl} --steps 2 --warmup 1 --no-cpu-baseline  (two passes: wave-state "
           f"counters, MFMA / LDS counters; per-launch averages over every launch of the kernel in the run; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles)"]
    names = sorted(set(a) & set(b), key=lambda n: -sum(v[2] for v in b[n].values()) )
    for n in names:
        if not any(rx.search(n) for rx in FAM.values()):
            continue
        ca, cb = a[n], b[n]
        g = lambda c, k: c[k][0] / c[k][1] if k in c and c[k][1] else float("nan")
        dur = cb["GRBM_GUI_ACTIVE"][2] / cb["GRBM_GUI_ACTIVE"][1] if "GRBM_GUI_ACTIVE" in cb else float("nan")      # ns per launch
        share = sum(v[2] for v in cb.values()) / max(1, len(cb))
        if dur < 20000:
            continue
        clk = g(cb, "GRBM_GUI_ACTIVE") / 8 / dur                           # GHz
        cyc = dur * clk                                                    # kernel cycles
        wc = g(ca, "SQ_WAVE_CYCLES")
        out.append(f"\n{short(n)}   ({cb['GRBM_GUI_ACTIVE'][1]} launches, {dur / 1e3:.1f} us each)")
        for k in sorted(ca):
            out.append(f"   {k:28s} {g(ca, k):.4g}")
        for k in sorted(cb):
            out.append(f"   {k:28s} {g(cb, k):.4g}")
        out.append(f"   derived: clock {clk:.2f} GHz | MFMA pipe busy {100 * g(cb, 'SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc:.1f} % | LDS array busy "
                   f"{100 * g(cb, 'SQ_LDS_IDX_ACTIVE') / 256 / cyc:.1f} % (bank-conflict cycles {100 * g(cb, 'SQ_LDS_BANK_CONFLICT') / max(g(cb, 'SQ_LDS_IDX_ACTIVE'), 1):.1f} % of those) | "
                   f"VALU instructions per MFMA {g(cb, 'SQ_INSTS_VALU') / max(g(cb, 'SQ_INSTS_MFMA'), 1):.1f}")
        out.append(f"            wave time: parked at s_waitcnt / s_barrier {100 * g(ca, 'SQ_WAIT_ANY') / wc:.0f} %, issue-stalled {100 * g(ca, 'SQ_WAIT_INST_ANY') / wc:.0f} % "
                   f"(LDS {100 * g(ca, 'SQ_WAIT_INST_LDS') / wc:.0f} %), issuing {100 * g(ca, 'SQ_ACTIVE_INST_ANY') / wc:.0f} %")
    open(f"{prefix}_pmc_sq_{SHORT[wl][0]}.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:30]))


for wl in SHORT:
    stats(wl)
    if os.path.isdir(os.path.join(src, f"pmc_{wl}")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), os.path.join(src, f"pmc_{wl}"),
                        f"{prefix}_pmc_traffic_{SHORT[wl][0]}_b{SHORT[wl][1]}.json", "--workload", wl, "--batch", str(SHORT[wl][1])], check=False)
    if os.path.isdir(os.path.join(src, f"sq_{wl}")):
        sq(wl)
    bj = os.path.join(src, f"bench_{wl}.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        open(f"{prefix}_bench_{SHORT[wl][0]}_b{SHORT[wl][1]}.json", "w").write(open(bj).read().strip().splitlines()[-1] + "\n")
