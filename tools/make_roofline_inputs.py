#!/usr/bin/env python3
"""profiles/roofline_inputs.json + profiles/<tag>_pmc_summary.md from one round of measurements.

    python tools/make_roofline_inputs.py gpurun_out/pmc_<tag> gpurun_out/<dir>/occupancy.txt <tag>

Inputs: the rocprofv3 --pmc passes of tools/profile_pmc.sh (one counter group per pass, no trace
domains), the pinned-occupancy instruction-rate table of tools/ubench/occupancy, and the compiler's
instruction mix (tools/isa_mix.py, run here).  The JSON carries the sha256 of the kernel
sources (dusk_zerocaf_amd.build.sources_sha256) the counters were taken on; bench.py drops every PMC-derived
roofline field when the tree's sources differ.
Per-unit figures: counter total over every dispatch of the workload's kernels / (calls x units per call).
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), the gfx950 correction MI355X_MICROARCH.md
prescribes for wide coalesced reads (the guide calls other access widths uncalibrated: the 2x is an
upper bound for the narrow gathers of the windowed core and the MSM)."""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dusk_zerocaf_amd", "libzerocaf_hip.so")
CALLS = 3                                   # profile_pmc.sh: --warmup 1 --steps 2
WORKLOADS = {
    # name: (units per call, algorithmic bytes per unit, dispatch filter, ISA-mix kernel)
    "scalar_mul": (1 << 20, 360, lambda k: k.startswith(("k_ed_scalar_mul", "k_sm_cost")), "k_ed_scalar_mul_pw"),
    "ristretto": (1 << 22, 104, lambda k: k.startswith("k_ris_roundtrip_mul_fast"), "k_ris_roundtrip_mul_fast"),
    # the MSM's own kernels only: not the inputs (k_ed_mul_base, k_base_table_build), not the live rate measurement
    # (k_mad_chains), not the runtime's fill / copy kernels, which multiply nothing
    "msm": (1 << 21, 200, lambda k: k.startswith(("k_msm", "k_scan", "k_ed_scalar_mul", "k_ed_add", "k_ed_fold")), "k_msm_runs_affine"),
}
MSM_STEP_KERNELS = ("k_msm", "k_scan", "k_ed_scalar_mul", "k_ed_add", "k_ed_fold")


def totals(path, keep):
    """counter -> (sum over kept dispatches, number of kept dispatches); kernel -> counter -> sum"""
    tot, per_kernel, disp = {}, {}, set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not keep(k):
            continue
        v = float(r["Counter_Value"])
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + v
        short = re.sub(r"<.*", "", k.split("(")[0])[-60:]
        per_kernel.setdefault(short, {}).setdefault(r["Counter_Name"], 0.0)
        per_kernel[short][r["Counter_Name"]] += v
        disp.add(r["Dispatch_Id"])
    return tot, len(disp), per_kernel


def parse_occupancy(path):
    rows = {}
    for l in open(path):
        m = re.match(r"^(.{28})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m:
            rows.setdefault(m.group(1).strip(), {})[int(m.group(2))] = {
                "ms": float(m.group(3)), "cyc_per_slot_median_wave": float(m.group(4)), "spread": float(m.group(5)),
                "rounds": float(m.group(6)), "ghz": float(m.group(7)), "T_lane_ops_per_s": float(m.group(9))}
    return rows


def main():
    pmc_dir, occ_path, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    prof_dir = sys.argv[4] if len(sys.argv) > 4 else None      # gpurun_out/prof_<tag>: kernel traces (the MSM's k_msm_runs share)
    prof = os.path.join(ROOT, "profiles")
    raw = os.path.join(prof, tag + "_raw")
    os.makedirs(raw, exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    sys.path.insert(0, ROOT)
    from dusk_zerocaf_amd import build as zbuild
    text = isa_mix.compile_asm()
    mixes = {k: isa_mix.mix(text, k, inner=k.endswith("_pw")) for k in sorted({w[3] for w in WORKLOADS.values()} | {"k_ed_scalar_mul_fast"})}
    json.dump(mixes, open(os.path.join(prof, tag + "_isa_mix.json"), "w"), indent=1)

    occ = parse_occupancy(occ_path)
    shutil.copy(occ_path, os.path.join(prof, tag + "_ubench_occupancy.txt"))
    mad = occ["v_mad_u64_u32"]
    best = max(mad.values(), key=lambda r: r["T_lane_ops_per_s"])
    ubench = {
        "source": "profiles/%s_ubench_occupancy.txt (tools/ubench/occupancy.hip: waves per SIMD pinned with LDS, s_memtime per wave, HIP events per launch)" % tag,
        "v_mad_u64_u32_T_lane_ops_per_s": best["T_lane_ops_per_s"],
        "v_mad_u64_u32_best_at_waves_per_simd": [k for k, r in mad.items() if r is best][0],
        "v_mad_u64_u32_shader_clock_ghz_at_best": best["ghz"],
        "v_mad_u64_u32_cycles_per_wave_inst_per_simd_at_best": round(256 * 4 * 64 * best["ghz"] * 1e9 / (best["T_lane_ops_per_s"] * 1e12), 3),
        "v_mad_u64_u32_cycles_one_wave": mad[1]["cyc_per_slot_median_wave"],
        "v_mad_u64_u32_cycles_two_waves": mad[2]["cyc_per_slot_median_wave"],
        "v_mad_u64_u32_T_lane_ops_per_s_if_4_cycles_at_2p4_GHz": round(256 * 4 * 64 * 2.4e9 / 4 / 1e12, 2),
        "v_fma_f32_T_lane_ops_per_s": max(r["T_lane_ops_per_s"] for r in occ["v_fma_f32"].values()),
        "v_add_u32_T_lane_ops_per_s": max(r["T_lane_ops_per_s"] for r in occ["v_add_u32"].values()),
    }

    # the hash of the sources the profiled library was built from (zc_version(), recorded by profile_pmc.sh)
    ver = os.path.join(pmc_dir, "lib_version.txt")
    src_hash = open(ver).read().strip().rsplit("src:", 1)[-1] if os.path.exists(ver) else zbuild.sources_sha256()
    if src_hash != zbuild.sources_sha256():
        print("WARNING: the profile was taken on other kernel sources (%s...) than the tree's (%s...)" % (src_hash[:12], zbuild.sources_sha256()[:12]))
    out = {"source": "profiles/%s_pmc_summary.md" % tag, "kernel_sources_sha256": src_hash,
           "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
           "ubench": ubench, "kernels": {}}
    md = ["# Round %s: rocprofv3 --pmc passes (tools/profile_pmc.sh; one counter group per pass, no trace domains)" % tag[1:].lstrip("0"), "",
          "Raw CSVs: profiles/%s_raw/.  Kernel sources sha256 %s.  Every figure is the total over all dispatches of the workload's kernels in "
          "%d calls (1 warm-up + 2 timed), divided by calls x units." % (tag, out["kernel_sources_sha256"][:16], CALLS), ""]
    for wl, (units, alg, keep, mixk) in WORKLOADS.items():
        c, per_kernel = {}, {}
        for grp in ("sq1", "sq2", "fetch", "write"):
            path = os.path.join(pmc_dir, "%s_%s_counter_collection.csv" % (wl, grp))
            if not os.path.exists(path):
                continue
            shutil.copy(path, raw)
            t, nd, pk = totals(path, keep)
            c.update(t)
            c["_dispatches"] = nd
            for k, v in pk.items():
                per_kernel.setdefault(k, {}).update(v)
        if "SQ_INSTS_VALU" not in c:
            continue
        per = lambda name: c[name] / (CALLS * units) if name in c else None
        rd = 2 * c["FETCH_SIZE"] * 1024 / (CALLS * units) if "FETCH_SIZE" in c else None
        wr = c["WRITE_SIZE"] * 1024 / (CALLS * units) if "WRITE_SIZE" in c else None
        k = {"units_per_call": units, "calls": CALLS, "dispatches": c["_dispatches"],
             "valu_insts_per_unit": per("SQ_INSTS_VALU"), "multiplier_rate_share": mixes[mixk]["multiplier_rate_share"],
             "isa_mix_kernel": mixk,
             "hbm_read_bytes_per_unit": rd, "hbm_write_bytes_per_unit": wr,
             "hbm_bytes_per_unit": (rd + wr) if rd is not None and wr is not None else None,
             "algorithmic_bytes_per_unit": alg, "kernels_counted": sorted(per_kernel)}
        if wl == "msm" and prof_dir:
            # the bucket-sum kernel's share of a step, per traced size (the gather block of bench.py prices that kernel alone)
            k["runs"] = {}
            for units_tag, units_n in (("2p21", 1 << 21), ("2p24", 1 << 24)):
                st = os.path.join(prof_dir, "kt_msm_%s_kernel_stats.csv" % units_tag)
                if not os.path.exists(st):
                    continue
                rows = [r for r in csv.DictReader(open(st)) if r["Name"].startswith(MSM_STEP_KERNELS)]
                tot = sum(float(r["TotalDurationNs"]) for r in rows)
                runs = [r for r in rows if r["Name"].startswith("k_msm_runs_affine")]
                steps = [r for r in rows if r["Name"].startswith("k_msm_digits")]                 # one per step
                if runs and tot and steps:
                    nsteps = int(steps[0]["Calls"])
                    # window groups launch the kernel once per group: the figure is its time PER STEP, all launches together
                    k["runs"][str(units_n)] = {"time_share": round(float(runs[0]["TotalDurationNs"]) / tot, 4),
                                               "avg_ms": round(float(runs[0]["TotalDurationNs"]) / nsteps / 1e6, 4),
                                               "launches_per_step": round(int(runs[0]["Calls"]) / nsteps, 2),
                                               "source": "profiles/%s_raw/kt_msm_%s_kernel_stats.csv (k_msm_runs_affine per step, over the summed kernel time of a step)" % (tag, units_tag)}
        for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                     "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
            if name in c:
                k[name + "_per_call"] = c[name] / CALLS
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            k["wave_cycle_split"] = {n: round(c[n] / wc, 4) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if n in c}
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out["kernels"][wl] = k
        md += ["## %s (%d units per call, %d dispatches)" % (wl, units, c["_dispatches"]), "",
               "| quantity | per unit | note |", "|---|---|---|",
               "| SQ_INSTS_VALU (wave-instructions) | %.2f | x 64 lanes; multiplier-rate share of the hot loop %.3f (%s, tools/isa_mix.py) |" % (
                   k["valu_insts_per_unit"], k["multiplier_rate_share"], mixk)]
        if k["hbm_bytes_per_unit"] is not None:
            md.append("| HBM bytes: 2 x FETCH_SIZE + WRITE_SIZE | %.1f read + %.1f written = %.1f | algorithmic %d B: %.2fx |" % (
                rd, wr, rd + wr, alg, (rd + wr) / alg))
        if "wave_cycle_split" in k:
            md.append("| wave cycles: issuing / issue-stalled / parked | %s | SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES |" % (
                " / ".join("%.3f" % v for v in k["wave_cycle_split"].values())))
        if "lds_bank_conflict_frac" in k:
            md.append("| LDS bank-conflict cycles / LDS active cycles | %.4f | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE |" % k["lds_bank_conflict_frac"])
        if "SQ_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
            md.append("| resident waves per busy SQ cycle | %.2f | SQ_WAVE_CYCLES / SQ_BUSY_CYCLES |" % (c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]))
        md.append("")
        if len(per_kernel) > 1:
            md += ["| kernel | SQ_INSTS_VALU per call | 2 x FETCH_SIZE MB per call | WRITE_SIZE MB per call | LDS bank-conflict / LDS active cycles | wave cycles issuing / issue-stalled / parked |",
                   "|---|---|---|---|---|---|"]
            for kn, v in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
                conf = "%.3f (%.2e / %.2e)" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"] / CALLS, v["SQ_LDS_IDX_ACTIVE"] / CALLS) \
                    if v.get("SQ_LDS_IDX_ACTIVE") else "no LDS"
                wcy = v.get("SQ_WAVE_CYCLES")
                split = " / ".join("%.2f" % (v.get(n, 0) / wcy) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")) if wcy else ""
                md.append("| %s | %.3e | %.1f | %.1f | %s | %s |" % (kn, v.get("SQ_INSTS_VALU", 0) / CALLS, 2 * v.get("FETCH_SIZE", 0) / CALLS / 1024,
                                                                v.get("WRITE_SIZE", 0) / CALLS / 1024, conf, split))
            md.append("")
    md += ["## Instruction rates behind `roofline.peak` (profiles/%s_ubench_occupancy.txt)" % tag, "",
           "```", json.dumps(ubench, indent=1), "```", ""]
    open(os.path.join(prof, tag + "_pmc_summary.md"), "w").write("\n".join(md))
    json.dump(out, open(os.path.join(prof, "roofline_inputs.json"), "w"), indent=1)
    print("wrote profiles/roofline_inputs.json, profiles/%s_pmc_summary.md, profiles/%s_isa_mix.json" % (tag, tag))


if __name__ == "__main__":
    main()
