#!/usr/bin/env python3
"""Per-op device-resident timings (HIP events on the launch stream) for the secondary
kernels of the path.  Prints one JSON object per (op, n).  GPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dusk_zerocaf_amd as z  # noqa: E402
from oracle import pymodel as pm  # noqa: E402


def main():
    ops = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fe_add", "fe_mul", "fe_square", "fe_invert"]
    sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 20, 1 << 24]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    eng = z.Engine([0])
    st = torch.cuda.current_stream()
    eng.set_stream(st.cuda_stream)
    rng = np.random.default_rng(1)
    for n in sizes:
        def fe():
            a = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
            a[:, 4] >>= np.uint64(8)
            return torch.from_numpy(a.view(np.int64)).cuda()
        a, b = fe(), fe()
        sa, sb = fe(), fe()
        sa[:, 4] >>= 3                                            # canonical scalars (< 2^249 < L): what k_sc_mul's one-pass product is for
        sb[:, 4] >>= 3
        pts = pts2 = enc = None
        for op in ops:
            if op.startswith(("ed_", "ris_")) and pts is None:
                base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
                pts = eng.ed_scalar_mul(torch.from_numpy(base.view(np.int64)).cuda(), a)
                pts2 = pts.clone()                               # a second array: two-input ops really move both operands
                enc = eng.ris_compress(pts)
                edc, _ = eng.ed_compress(pts)
            bytes_per = {"fe_add": 120, "fe_sub": 120, "fe_mul": 120, "fe_square": 80, "fe_neg": 80, "fe_invert": 80, "fe_div": 120,
                         "sc_mul": 120, "sc_mul_raw": 120, "sc_square": 80, "sc_square_raw": 80, "sc_neg": 80, "ed_add": 480, "ed_double": 320, "ed_compress": 192, "ed_decompress": 192,
                         "ris_compress": 192, "ris_decompress": 192, "ed_to_affine": 240, "fe_sqrt_ratio_i": 120, "fe_legendre": 41, "ed_neg": 320, "ed_eq": 321, "ris_eq": 321, "ed_is_valid": 161, "ed_mul_base": 200, "ris_mul_base_compress": 72, "ed_mul_base_wnaf5": 200}[op]
            fn = {"fe_add": lambda: eng.fe_add(a, b), "fe_sub": lambda: eng.fe_sub(a, b), "fe_mul": lambda: eng.fe_mul(a, b),
                  "fe_square": lambda: eng.fe_square(a), "fe_neg": lambda: eng.fe_neg(a), "fe_invert": lambda: eng.fe_invert(a), "fe_div": lambda: eng.fe_div(a, b),
                  "sc_mul": lambda: eng.sc_mul(sa, sb), "sc_square": lambda: eng.sc_square(sa), "sc_neg": lambda: eng.sc_neg(sa),
                  # raw 252-bit patterns as scalars: 7 of 8 at or above 2^249, the two-pass product
                  "sc_mul_raw": lambda: eng.sc_mul(a, b), "sc_square_raw": lambda: eng.sc_square(a), "fe_sqrt_ratio_i": lambda: eng.fe_sqrt_ratio_i(a, b), "fe_legendre": lambda: eng.fe_legendre_symbol(a),
                  "ed_add": lambda: eng.ed_add(pts, pts2), "ed_double": lambda: eng.ed_double(pts),
                  "ed_compress": lambda: eng.ed_compress(pts), "ed_decompress": lambda: eng.ed_decompress(edc),
                  "ris_compress": lambda: eng.ris_compress(pts), "ris_decompress": lambda: eng.ris_decompress(enc),
                  "ed_to_affine": lambda: eng.ed_to_affine(pts),
                  "ed_neg": lambda: eng.ed_neg(pts), "ed_eq": lambda: eng.ed_eq(pts, pts2), "ris_eq": lambda: eng.ris_eq(pts, pts2),
                  "ed_is_valid": lambda: eng.ed_is_valid(pts),
                  "ed_mul_base": lambda: eng.ed_mul_base(a), "ris_mul_base_compress": lambda: eng.ris_mul_base_compress(a),
                  "ed_mul_base_wnaf5": lambda: eng.ed_mul_base_wnaf(a, 5)}[op]
            fn()
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(json.dumps({"op": op, "n": n, "ms": round(ms, 4), "M_per_s": round(n / ms / 1e3, 1),
                              "alg_GBps": round(bytes_per * n / ms / 1e6, 1),
                              # at most 256 MB per launch: the arrays stay in the Infinity Cache between the repetitions,
                              # the figure is cache bandwidth and may exceed the 8 TB/s HBM roof
                              "cache_resident": bool(bytes_per * n <= 256 << 20)}), flush=True)


if __name__ == "__main__":
    main()
