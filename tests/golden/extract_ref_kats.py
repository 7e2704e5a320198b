#!/usr/bin/env python3
"""
Extract the known-answer DATA (numbers only) held by the reference's own unit
tests and constant tables into tests/golden/ref_kats.json.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box); the JSON it writes is committed and is what the tests read.
It copies no code: only named numeric constants (limb arrays, byte arrays, hex
strings) with the file:line each came from.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")

NUM5 = r"\[\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,?\s*\]"


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def limb_consts(path, type_names):
    """NAME: FieldElement = FieldElement([a,b,c,d,e]) style constants."""
    text = open(os.path.join(REF, path)).read()
    out = {}
    pat = re.compile(
        r"(?:pub(?:\([a-z]+\))?\s+)?(?:static|const)\s+([A-Z0-9_]+)\s*:\s*(%s)\s*=\s*(?:%s)\(\s*%s\s*\)"
        % ("|".join(type_names), "|".join(type_names), NUM5),
        re.S,
    )
    for m in pat.finditer(text):
        out[m.group(1)] = {
            "limbs": [int(x) for x in m.groups()[2:7]],
            "cite": "%s:%d" % (path, line_of(text, m.start())),
        }
    return out


def point_consts(path):
    """NAME: EdwardsPoint/ProjectivePoint/AffinePoint = ... { X: FieldElement([..]), ... }"""
    text = open(os.path.join(REF, path)).read()
    out = {}
    pat = re.compile(
        r"(?:pub(?:\([a-z]+\))?\s+)?(?:static|const)\s+([A-Z0-9_]+)\s*:\s*(EdwardsPoint|ProjectivePoint|AffinePoint)\s*=\s*\2\s*\{(.*?)\};",
        re.S,
    )
    for m in pat.finditer(text):
        coords = {}
        for c in re.finditer(r"([XYZT])\s*:\s*FieldElement\(\s*%s\s*\)" % NUM5, m.group(3), re.S):
            coords[c.group(1)] = [int(x) for x in c.groups()[1:6]]
        out[m.group(1)] = {"kind": m.group(2), "coords": coords,
                           "cite": "%s:%d" % (path, line_of(text, m.start()))}
    return out


def byte_consts(path, type_name):
    text = open(os.path.join(REF, path)).read()
    out = {}
    pat = re.compile(
        r"(?:pub(?:\([a-z]+\))?\s+)?(?:static|const)\s+([A-Z0-9_]+)\s*:\s*(?:%s|\[u8;\s*32\])\s*=\s*(?:(?:%s)\()?\s*\[([\d\s,]+)\]" % (type_name, type_name),
        re.S,
    )
    for m in pat.finditer(text):
        vals = [int(x) for x in re.findall(r"\d+", m.group(2))]
        if len(vals) == 32:
            out[m.group(1)] = {"bytes": vals, "cite": "%s:%d" % (path, line_of(text, m.start()))}
    return out


def inline_byte_arrays(path):
    """32-element byte literals appearing inside tests (from_slice(&[...]), from_bytes(&[...]))."""
    text = open(os.path.join(REF, path)).read()
    out = []
    for m in re.finditer(r"&\[\s*((?:(?:0x[0-9a-fA-F]+|\d+)\s*,\s*){31}(?:0x[0-9a-fA-F]+|\d+)\s*,?\s*)\]", text, re.S):
        vals = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]
        out.append({"bytes": vals, "cite": "%s:%d" % (path, line_of(text, m.start()))})
    return out


def inline_limb_arrays(path, lo, hi):
    """FieldElement([..]) literals inside a line range of a test module."""
    text = open(os.path.join(REF, path)).read()
    out = []
    for m in re.finditer(r"(FieldElement|Scalar)\(\s*%s\s*\)" % NUM5, text, re.S):
        ln = line_of(text, m.start())
        if lo <= ln <= hi:
            out.append({"limbs": [int(x) for x in m.groups()[1:6]], "cite": "%s:%d" % (path, ln)})
    return out


def odd_multiples_table():
    path = "src/backend/u64/constants.rs"
    text = open(os.path.join(REF, path)).read()
    start = text.index("BASEPOINT_ODD_MULTIPLES_TABLE")
    body = text[start:]
    pts = []
    for m in re.finditer(
        r"X:\s*FieldElement\(%s\)\s*,\s*Y:\s*FieldElement\(%s\)\s*,\s*Z:\s*FieldElement\(%s\)\s*,\s*T:\s*FieldElement\(%s\)"
        % (NUM5, NUM5, NUM5, NUM5), body, re.S):
        g = [int(x) for x in m.groups()]
        pts.append([g[0:5], g[5:10], g[10:15], g[15:20]])
    return {"points": pts, "cite": "%s:%d" % (path, line_of(text, start))}


def four_coset_group():
    """FOUR_COSET_GROUP: [EdwardsPoint; 4] (what EdwardsPoint::coset4 adds, src/edwards.rs:603-610)."""
    path = "src/backend/u64/constants.rs"
    text = open(os.path.join(REF, path)).read()
    start = text.index("FOUR_COSET_GROUP")
    body = text[start:text.index("];", start)]
    pts = []
    for m in re.finditer(
        r"X:\s*FieldElement\(%s\)\s*,\s*Y:\s*FieldElement\(%s\)\s*,\s*Z:\s*FieldElement\(%s\)\s*,\s*T:\s*FieldElement\(%s\)"
        % (NUM5, NUM5, NUM5, NUM5), body, re.S):
        g = [int(x) for x in m.groups()]
        pts.append([g[0:5], g[5:10], g[10:15], g[15:20]])
    return {"points": pts, "cite": "%s:%d" % (path, line_of(text, start))}


def hex_strings(path, lo, hi):
    text = open(os.path.join(REF, path)).read()
    out = []
    for m in re.finditer(r'"([0-9a-f]{64})"', text):
        ln = line_of(text, m.start())
        if lo <= ln <= hi:
            out.append({"hex": m.group(1), "cite": "%s:%d" % (path, ln)})
    return out


def main():
    kats = {
        "_about": "Numeric known-answer data transcribed from dusk-zerocaf's own tests/constants; "
                  "generated by tests/golden/extract_ref_kats.py",
        "constants": limb_consts("src/backend/u64/constants.rs", ["FieldElement", "Scalar"]),
        "constants_points": point_consts("src/backend/u64/constants.rs"),
        "field": limb_consts("src/backend/u64/field.rs", ["FieldElement"]),
        "field_bytes": byte_consts("src/backend/u64/field.rs", "X"),
        "field_inline_bytes": inline_byte_arrays("src/backend/u64/field.rs"),
        "field_division": inline_limb_arrays("src/backend/u64/field.rs", 1243, 1260),
        "field_dalek": inline_limb_arrays("src/backend/u64/field.rs", 1379, 1422),
        "scalar": limb_consts("src/backend/u64/scalar.rs", ["Scalar"]),
        "edwards_points": point_consts("src/edwards.rs"),
        "edwards_compressed": byte_consts("src/edwards.rs", "CompressedEdwardsY"),
        "edwards_inline_bytes": inline_byte_arrays("src/edwards.rs"),
        "top_constants_bytes": byte_consts("src/constants.rs", "CompressedEdwardsY|CompressedRistretto"),
        "ristretto_small_multiples": hex_strings("src/ristretto.rs", 546, 566),
        "ristretto_inline_bytes": inline_byte_arrays("src/ristretto.rs"),
        "ristretto_elligator_hex": hex_strings("src/ristretto.rs", 700, 715),
        "ristretto_elligator_point": inline_limb_arrays("src/ristretto.rs", 683, 706),
        "odd_multiples_table": odd_multiples_table(),
        "four_coset_group": four_coset_group(),
    }
    # scalar constants that are `pub const fn`-style or inline in tests
    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
    for k, v in kats.items():
        if k != "_about":
            n = len(v["points"]) if isinstance(v, dict) and "points" in v else len(v)
            print("%-28s %d" % (k, n))


if __name__ == "__main__":
    main()
