#!/usr/bin/env python3
"""A/B on one box: the fused Ristretto round trip and the strict scalar-mul of another build of the
library (same C ABI, e.g. round 1's: build/variants/round1.so) against the current one.
Only entry points both builds export are called, through ctypes directly."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402


def timed(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2], 3), round(ts[0], 3)


def main():
    other = C.CDLL(sys.argv[1])
    eng = z.Engine()
    s = torch.cuda.current_stream().cuda_stream
    eng.set_stream(s)
    ctx = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert other.zc_ctx_create(dev, 1, C.byref(ctx)) == 0
    assert other.zc_ctx_set_stream(ctx, C.c_void_p(s), 1) == 0
    p = lambda t: C.c_void_p(t.data_ptr())
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    out = {}
    for rnd in range(2):
        for lg in (20, 22):
            n = 1 << lg
            P = eng.ed_mul_base(dv(rand_scalars_np(n, 11, 249)))
            K = dv(rand_scalars_np(n, 12, 252))
            enc = eng.ris_compress(P)
            o1, o2 = torch.empty_like(enc), torch.empty_like(enc)
            ok = torch.empty(n, dtype=torch.uint8, device="cuda")
            fc = lambda: eng.ris_roundtrip_mul(enc, K, out=o1)
            fo = lambda: other.zc_ris_roundtrip_mul(ctx, p(enc), p(K), p(o2), p(ok), C.c_size_t(n))
            timed(fc, 6, 2)                                   # past the board's power transient
            for tag, f in (("cur", fc), ("other", fo), ("other", fo), ("cur", fc)):      # ABBA: box drift cancels
                out.setdefault("%s_ris_2p%d_r%d" % (tag, lg, rnd), []).append(timed(f, 5, 1)[0])
            assert torch.equal(o1, o2)
            if lg == 20:
                q1, q2 = torch.empty_like(P), torch.empty_like(P)
                gc = lambda: eng.ed_scalar_mul(P, K, out=q1)
                go = lambda: other.zc_ed_scalar_mul(ctx, p(P), p(K), p(q2), C.c_size_t(n), 0)
                timed(gc, 6, 2)
                for tag, f in (("cur", gc), ("other", go), ("other", go), ("cur", gc)):
                    out.setdefault("%s_strict_2p20_r%d" % (tag, rnd), []).append(timed(f, 7, 1)[0])
                assert torch.equal(q1, q2)
            del P, K, enc, o1, o2
    print(json.dumps(out))


main()
