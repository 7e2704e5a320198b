#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code object inside libzerocaf_hip.so (no GPU needed):
VGPRs, spilled VGPRs/SGPRs, private (scratch) segment, LDS -- read from the code object's
metadata notes.  Usage: python tools/kernel_resources.py [substring ...] [--json out.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dusk_zerocaf_amd", "libzerocaf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = (".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size", ".sgpr_count")


def resources(lib=LIB):
    with tempfile.TemporaryDirectory() as d:
        co, fat = os.path.join(d, "zc.co"), os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "unused.so")], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    # one YAML map per kernel, keys in alphabetical order: .symbol precedes .vgpr_count, and
    # .wavefront_size closes the entry
    out, cur, name = {}, {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in KEYS:
            cur[k[1:]] = int(v)
        elif k == ".symbol":
            name = v.strip("'\"").removesuffix(".kd")
        elif k == ".wavefront_size" and name:
            out[name] = cur
            cur, name = {}, None
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = resources()
    names = sorted(n for n in res if not args or any(a in n for a in args))
    print("%-34s %5s %6s %6s %8s %7s" % ("kernel", "vgpr", "vspill", "sspill", "private", "lds"))
    for n in names:
        r = res[n]
        print("%-34s %5d %6d %6d %8d %7d" % (n, r.get("vgpr_count", -1), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0),
                                              r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0)))
    if "--json" in sys.argv:
        path = sys.argv[sys.argv.index("--json") + 1]
        json.dump({n: res[n] for n in names}, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
