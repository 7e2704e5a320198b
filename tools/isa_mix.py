#!/usr/bin/env python3
"""Instruction mix of the unified-step loop of k_ed_scalar_mul, from the compiler's own ISA.

Compiles the library source to gfx950 assembly (hipcc -S --offload-device-only, no GPU needed), finds
the largest backward-branch loop of the kernel and counts opcodes.  Writes the JSON named by --out (profiles/rNN_isa_mix.json),
which tools/make_roofline_inputs.py uses to split the PMC instruction count into the multiplier-rate class
(v_mad_u64_u32, v_mul_lo_u32, 64-bit shifts: ~5 cycles per wave-instruction per SIMD) and the rest.
Usage: python tools/isa_mix.py [kernel[:whole|:inner] ...] [--out profiles/rNN_isa_mix.json]
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dusk_zerocaf_amd", "csrc", "zerocaf_hip.hip")
SLOW = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64")


def compile_asm():
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "zc.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                        "-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
        return open(asm).read()


def mix(text, kernel, whole=False, inner=False):
    """Opcode counts of the kernel's largest inner loop (whole=True: of the whole kernel body, every
    instruction counted once -- for kernels whose time is spread over several loops of the same make-up;
    inner=True: the smallest loop that still holds >= 1000 v_mad_u64_u32 -- the step loop of a kernel
    whose outermost loop walks tiles)."""
    i0 = text.index("\n%s:" % kernel)
    body = text[i0:text.index(".Lfunc_end", i0)]
    lines = []
    for l in body.split("\n"):
        l = l.split(";")[0].strip()
        if l and not l.startswith((".p2align", ".section", ".type", ".globl", "#")):
            lines.append(l)
    labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", l)}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    # the step loop: the largest loop that is nested in the (slightly larger) per-block loop, if any
    loops.sort(reverse=True)
    size, a, b = loops[1] if len(loops) > 1 and loops[1][0] > 0.9 * loops[0][0] else loops[0]
    if inner:
        cands = [(sz, x, y) for sz, x, y in loops if sum(1 for l in lines[x:y] if l.startswith("v_mad_u64_u32")) >= 1000]
        size, a, b = min(cands)
    if whole:
        a, b = 0, len(lines)
    ops = collections.Counter(l.split()[0] for l in lines[a:b] if not l.endswith(":"))
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    slow = {k: v for k, v in ops.items() if k in SLOW}
    return {"kernel": kernel, "source": "hipcc -S --offload-device-only, " + ("whole kernel body" if whole else "largest inner loop"),
            "valu_per_step": valu, "multiplier_rate_class_per_step": sum(slow.values()), "multiplier_rate_class": slow,
            "multiplier_rate_share": round(sum(slow.values()) / valu, 4),
            "s_nop_per_step": ops.get("s_nop", 0),
            "other_valu": {k: v for k, v in ops.most_common() if k.startswith("v_") and k not in SLOW}}


def main():
    """python tools/isa_mix.py [kernel[:whole|:inner] ...] [--out profiles/rNN_isa_mix.json]"""
    args = sys.argv[1:]
    out_path = None
    if "--out" in args:
        i = args.index("--out")
        out_path = args[i + 1]
        del args[i:i + 2]
    kernels = args or ["k_ed_scalar_mul"]
    text = compile_asm()
    res = {}
    for k in kernels:
        name, _, mode = k.partition(":")
        res[name] = mix(text, name, whole=(mode == "whole"), inner=(mode == "inner"))
        print(json.dumps({x: res[name][x] for x in ("kernel", "valu_per_step", "multiplier_rate_class_per_step", "multiplier_rate_share", "multiplier_rate_class")}))
    if out_path:
        json.dump(res, open(out_path if os.path.isabs(out_path) else os.path.join(ROOT, out_path), "w"), indent=1)


if __name__ == "__main__":
    main()
