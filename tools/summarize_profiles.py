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
WARMUP = {"scalar_mul": 3, "ristretto": 1, "fe_mul": 30, "msm_2p21": 1, "msm_2p24": 1}
STEPS = {"scalar_mul": 5, "ristretto": 5, "fe_mul": 20, "msm_2p21": 5, "msm_2p24": 3}
BENCH_OF = {"scalar_mul": "bench_scalar_mul.json", "ristretto": "bench_ristretto_2p22.json", "fe_mul": "bench_fe_mul_2p24.json",
            "msm_2p21": "bench_msm_2p21.json", "msm_2p24": "bench_msm_2p24.json"}
# Kernels of a trace that are NOT part of a bench step: input generation (k_ed_mul_base and its table, k_ris_compress for the
# Ristretto encodings), the live instruction-rate measurement after the timed region (k_mad_chains), torch's own kernels.
# The runtime's fill / copy kernels are listed with the steps they belong to but carry no arithmetic.
STEP_KERNELS = {
    "scalar_mul": ("k_ed_scalar_mul", "k_sm_cost"),
    "ristretto": ("k_ris_roundtrip_mul",),
    "fe_mul": ("k_fe_mul",),
    "msm_2p21": ("k_msm", "k_scan", "k_ed_scalar_mul", "k_ed_add", "k_ed_fold"),
    "msm_2p24": ("k_msm", "k_scan", "k_ed_scalar_mul", "k_ed_add", "k_ed_fold"),
}


def kernel_table(path, name):
    rows = list(csv.DictReader(open(path)))
    calls = WARMUP[name] + STEPS[name]
    mine = [r for r in rows if r["Name"].startswith(STEP_KERNELS[name])]
    other = [r for r in rows if r not in mine]
    out = ["| kernel of a step | calls | per step | avg (us) | min (us) | max (us) | us per step |", "|---|---|---|---|---|---|---|"]
    total = 0.0
    for r in mine:
        per = float(r["TotalDurationNs"]) / calls / 1e3
        total += per
        out.append("| %s | %s | %.1f | %.1f | %.1f | %.1f | %.1f |" % (r["Name"][:70], r["Calls"], int(r["Calls"]) / calls, float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, per))
    out += ["", "Sum over the step's own kernels: **%.3f ms of kernel time per step** (%d steps in the trace: %d warm-up + %d timed)." % (
        total / 1e3, calls, WARMUP[name], STEPS[name]), "",
            "| not part of a step (input generation, the live rate measurement after the timed region, runtime fills / copies) | calls | avg (us) |", "|---|---|---|"]
    for r in other:
        out.append("| %s | %s | %.1f |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    return out, rows, total / 1e3


def step_spans(trace, name):
    """Wall-clock span of every step from the kernel trace (first kernel start to last kernel end of the step's own kernels):
    with two streams in flight (the MSM's normalisation beside its sort) the sum of kernel times exceeds it."""
    rows = sorted((r for r in csv.DictReader(open(trace)) if r["Kernel_Name"].startswith(STEP_KERNELS[name])), key=lambda r: int(r["Start_Timestamp"]))
    if not name.startswith("msm"):
        return None
    spans, first = [], None
    for r in rows:
        if first is None:
            first = int(r["Start_Timestamp"])
        if r["Kernel_Name"].startswith("k_ed_fold_ordered"):      # the last kernel of a zc_msm_sharded step
            spans.append((int(r["End_Timestamp"]) - first) / 1e6)
            first = None
    return spans


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
    shutil.rmtree(raw, ignore_errors=True)        # rewritten from scratch: keep hand-collected experiment records elsewhere (profiles/<tag>_experiments/)
    os.makedirs(raw)
    for f in glob.glob(os.path.join(src, "*.csv")):
        if f.endswith(("_kernel_stats.csv", "_counter_collection.csv")) or f.endswith("_kernel_trace.csv"):
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
        table, rows, per_step_ms = kernel_table(path, name)
        md += ["## " + title, ""] + table + [""]
        trace = os.path.join(src, "kt_%s_kernel_trace.csv" % name)
        bj = os.path.join(src, BENCH_OF[name])
        if os.path.exists(bj) and os.path.getsize(bj):
            b = json.load(open(bj))
            md += ["The same workload's bench line (%s, a separate run on the same box): `ms_per_step` %.3f, `kernel_avg_ms` %.3f." % (
                BENCH_OF[name], b["ms_per_step"], b["roofline"]["kernel_avg_ms"]), ""]
        if os.path.exists(trace) and name in DOMINANT:
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace))
                    if r["Kernel_Name"].startswith(DOMINANT[name])]
            timed = durs[WARMUP[name]:]
            md += ["%s per dispatch, in order (ms): %s.  Warm-up launches first: %d; the "
                   "%d timed launches average %.3f ms, which is what bench.py reports as `kernel_avg_ms`." % (
                       DOMINANT[name], ", ".join("%.2f" % d for d in durs), WARMUP[name], len(timed), sum(timed) / len(timed)), ""]
        if os.path.exists(trace) and name.startswith("msm"):
            spans = step_spans(trace, name)
            if spans:
                timed = spans[WARMUP[name]:]
                md += ["Span of each step in the trace, first kernel start to last kernel end (ms): %s; the %d timed steps average %.3f ms.  "
                       "Shards below 2^23 pairs run k_msm_prepare_affine on a second stream beside the sort, so there the sum of kernel times "
                       "above exceeds the span." % (", ".join("%.3f" % x for x in spans), len(timed), sum(timed) / len(timed)), ""]
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
