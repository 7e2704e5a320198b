#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh into the committed evidence under profiles/.

Usage: python tools/summarize_profiles.py gpurun_out/prof r01
Writes profiles/<tag>_kernel_stats.md, <tag>_pmc_summary.md, <tag>_hbm_traffic.json (read by bench.py for
`roofline.traffic`), <tag>_bench_*.json and copies the raw CSVs to profiles/<tag>_raw/.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KT = [("scalar_mul", "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --cpu-sample 0` "
                     "(strict scalar-mul, 2^20 per launch; k_ed_scalar_mul calls = 1 warmup + 5 timed; inputs come from k_ed_mul_base)"),
      ("ristretto", "`... --workload ristretto` (fused decompress -> scalar-mul -> compress on the windowed core, 2^20; "
                    "inputs come from k_ed_mul_base + k_ris_compress)"),
      ("msm", "`... --workload msm` (bucket method, 2^20 pairs per launch)"),
      ("fe_mul", "`... --workload fe_mul --units 16777216 --steps 20 --warmup 30` (2^24 elements, 2.0 GB per launch; the long warm-up "
       "steps over the board's power transient: from idle the launches run 0.36 ms, rise to 0.50 ms around the tenth and "
       "settle at 0.39-0.42 ms)")]


DOMINANT = {"scalar_mul": "k_ed_scalar_mul", "ristretto": "k_ris_roundtrip_mul_fast", "fe_mul": "k_fe_mul"}
WARMUP = {"scalar_mul": 1, "ristretto": 1, "fe_mul": 30}


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
        if name == "scalar_mul":
            sm_avg_ms = [float(r["AverageNs"]) / 1e6 for r in rows if r["Name"] == "k_ed_scalar_mul"][0]
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

    # ---- PMC
    pm = ["# Round %s: rocprofv3 --pmc passes (one counter set per run, no trace domains)" % tag[1:].lstrip("0"), "",
          "Raw CSVs: profiles/%s_raw/." % tag, "", "| workload | kernel | counter | dispatches | mean per dispatch |",
          "|---|---|---|---|---|"]
    c = {}
    for f, wl, kern in [("pmc_fetch", "scalar_mul", "k_ed_scalar_mul"), ("pmc_write", "scalar_mul", "k_ed_scalar_mul"),
                        ("pmc_sq", "scalar_mul", "k_ed_scalar_mul"), ("pmc_fetch_fe", "fe_mul_2^24", "k_fe_mul"),
                        ("pmc_write_fe", "fe_mul_2^24", "k_fe_mul")]:
        path = os.path.join(src, f + "_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for name, (cnt, mean) in sorted(counters(path, kern).items()):
            pm.append("| %s | %s | %s | %d | %.1f |" % (wl, kern, name, cnt, mean))
            c[(wl, name)] = mean
    pm += ["", "FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half "
           "of a wide coalesced read stream, so read bytes = 2 x FETCH_SIZE x 1024.", ""]
    traffic = {"source": "profiles/%s_pmc_summary.md" % tag}
    n20, n24 = 1 << 20, 1 << 24
    if ("scalar_mul", "FETCH_SIZE") in c and ("scalar_mul", "WRITE_SIZE") in c:
        rd, wr = 2 * c[("scalar_mul", "FETCH_SIZE")] * 1024, c[("scalar_mul", "WRITE_SIZE")] * 1024
        traffic["scalar_mul"] = rd + wr
        pm.append("* k_ed_scalar_mul, 2^20 points: read %.1f MB (algorithmic 2^20 x 200 B = %.1f MB), written %.1f MB "
                  "(algorithmic 2^20 x 160 B = %.1f MB)." % (rd / 1e6, n20 * 200 / 1e6, wr / 1e6, n20 * 160 / 1e6))
    if ("fe_mul_2^24", "FETCH_SIZE") in c and ("fe_mul_2^24", "WRITE_SIZE") in c:
        rd, wr = 2 * c[("fe_mul_2^24", "FETCH_SIZE")] * 1024, c[("fe_mul_2^24", "WRITE_SIZE")] * 1024
        traffic["fe_mul_per_unit_bytes"] = (rd + wr) / n24
        pm.append("* k_fe_mul, 2^24 elements: read %.1f MB (algorithmic 2^24 x 80 B = %.1f MB), written %.1f MB "
                  "(algorithmic 2^24 x 40 B = %.1f MB)." % (rd / 1e6, n24 * 80 / 1e6, wr / 1e6, n24 * 40 / 1e6))
    if ("scalar_mul", "SQ_INSTS_VALU") in c:
        valu, waves = c[("scalar_mul", "SQ_INSTS_VALU")], c[("scalar_mul", "SQ_WAVES")]
        traffic["scalar_mul_valu_insts_per_launch"] = valu
        line = "* k_ed_scalar_mul: SQ_INSTS_VALU = %.3e wave-instructions per dispatch over SQ_WAVES = %d waves = %.0f per wave" % (
            valu, waves, valu / waves)
        if sm_avg_ms and ("scalar_mul", "GRBM_GUI_ACTIVE") in c:
            line += "; GRBM_GUI_ACTIVE / 8 XCDs / %.2f ms (kernel-trace average) = %.2f GHz effective clock" % (
                sm_avg_ms, c[("scalar_mul", "GRBM_GUI_ACTIVE")] / 8 / (sm_avg_ms * 1e-3) / 1e9)
        if ("scalar_mul", "SQ_BUSY_CYCLES") in c and ("scalar_mul", "SQ_WAVE_CYCLES") in c:
            line += "; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = %.1f resident waves per busy SQ cycle" % (
                c[("scalar_mul", "SQ_WAVE_CYCLES")] / c[("scalar_mul", "SQ_BUSY_CYCLES")])
        pm.append(line + ".")
    open(os.path.join(prof, tag + "_pmc_summary.md"), "w").write("\n".join(pm) + "\n")
    json.dump(traffic, open(os.path.join(prof, tag + "_hbm_traffic.json"), "w"), indent=1)
    print("wrote profiles/%s_{kernel_stats.md,pmc_summary.md,hbm_traffic.json}" % tag)


if __name__ == "__main__":
    main()
