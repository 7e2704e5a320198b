#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh into the committed evidence under profiles/.

Usage: python tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>
Writes profiles/<tag>_kernel_stats.md, <tag>_bench_*.json and copies the raw CSVs to profiles/<tag>_raw/
(the PMC passes are summarised by tools/make_roofline_inputs.py).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KT = [("scalar_mul", "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 3 --cpu-sample 0` "
                     "(strict scalar-mul, 2^20 per launch; inputs come from k_ed_mul_base)"),
      ("ristretto", "`... --workload ristretto --units 4194304 --steps 5 --warmup 1` (fused decompress -> scalar-mul -> compress on the "
                    "windowed core, 2^22 per call = one launch over the 256 MB table ring; inputs come from k_ed_mul_base + k_ris_compress)"),
      ("msm_2p21", "`... --workload msm --units 2097152 --steps 5 --warmup 1` (bucket method, 2^21 pairs per call: the per-GPU shard of BASELINE configs[4])"),
      ("msm_2p24", "`... --workload msm --units 16777216 --steps 3 --warmup 1` (2^24 pairs on one GPU)"),
      ("fe_mul", "`... --workload fe_mul --units 16777216 --steps 20 --warmup 30` (2^24 elements, 2.0 GB per launch; the long warm-up "
       "steps over the board's power transient)")]


DOMINANT = {"scalar_mul": "k_ed_scalar_mul", "ristretto": "k_ris_roundtrip_mul_fast", "fe_mul": "k_fe_mul"}
WARMUP = {"scalar_mul": 3, "ristretto": 1, "fe_mul": 30}
CALLS = {"msm_2p21": 6, "msm_2p24": 4}


def kernel_table(path):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg (us) | min (us) | max (us) | % |", "|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return out, rows


def counters(path, kernel):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith(kernel):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    raw = os.path.join(prof, tag + "_raw")
    shutil.rmtree(raw, ignore_errors=True)
    os.makedirs(raw)
    for f in glob.glob(os.path.join(src, "*.csv")):
        if f.endswith(("_kernel_stats.csv", "_counter_collection.csv")) or (f.endswith("_kernel_trace.csv") and "msm" not in f):
            shutil.copy(f, raw)
    for f in glob.glob(os.path.join(src, "bench_*.json")):
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(prof, tag + "_" + os.path.basename(f)))

    # ---- kernel trace
    md = ["# Round %s: rocprofv3 kernel-trace summaries, MI355X (gfx950)" % tag[1:].lstrip("0"), "",
          "Raw CSVs: profiles/%s_raw/ (collected by tools/profile_round.sh)." % tag, ""]
    sm_avg_ms = None
    for name, title in KT:
        path = os.path.join(src, "kt_%s_kernel_stats.csv" % name)
        if not os.path.exists(path):
            continue
        table, rows = kernel_table(path)
        md += ["## " + title, ""] + table + [""]
        trace = os.path.join(src, "kt_%s_kernel_trace.csv" % name)
        if os.path.exists(trace) and name in DOMINANT:
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace))
                    if r["Kernel_Name"].startswith(DOMINANT[name])]
            timed = durs[WARMUP[name]:]
            md += ["%s per dispatch, in order (ms): %s.  Warm-up launches first: %d; the "
                   "%d timed launches average %.3f ms, which is what bench.py reports as `kernel_avg_ms`." % (
                       DOMINANT[name], ", ".join("%.2f" % d for d in durs), WARMUP[name], len(timed), sum(timed) / len(timed)), ""]
        if name in CALLS:
            tot = sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("k_ed_mul_base", "k_base_table_build")))
            md += ["All kernels of one MSM call (%d calls in the trace): %.3f ms of kernel time per call." % (CALLS[name], tot / CALLS[name] / 1e6), ""]
    ops = os.path.join(src, "ops.txt")
    if os.path.exists(ops):
        md += ["## Secondary kernels, HIP events on the launch stream (`tools/bench_ops.py`)", "", "```"] + \
              [l.rstrip() for l in open(ops) if l.startswith("{")] + ["```", ""]
    hp = os.path.join(src, "host_path.txt")
    if os.path.exists(hp):
        md += ["## Host-pointer (PCIe-inclusive) strict scalar-mul, pageable numpy buffers (`tools/host_path.py`; "
               "chunks auto = overlapped 2^18-element chunks, 1 = one piece)", "", "```"] + \
              [l.rstrip() for l in open(hp) if l.startswith("{")] + ["```", ""]
    sp = os.path.join(src, "step_probe.txt")
    if os.path.exists(sp):
        md += ["## Unified-step probe (`tools/step_probe.py`)", "", "```"] + [l.rstrip() for l in open(sp)] + ["```", ""]
    open(os.path.join(prof, tag + "_kernel_stats.md"), "w").write("\n".join(md))

    print("wrote profiles/%s_kernel_stats.md" % tag)


if __name__ == "__main__":
    main()
