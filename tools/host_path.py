"""Time the host-pointer (PCIe-inclusive) path of zc_ed_scalar_mul against the chunk count.

Usage: python tools/host_path.py [log2_n]
One JSON line per chunk count: best-of-7 wall time of the C-ABI call with numpy (pageable)
buffers, output buffer reused ("warm") and freshly allocated ("fresh"), device-resident time beside it.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402


def best(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << lg
    eng = z.Engine()
    k = rand_scalars_np(n, 7, 252)
    P = eng.ed_mul_base(rand_scalars_np(n, 8, 249))
    dev = torch.device("cuda:0")
    tP, tk = torch.from_numpy(P.view(np.int64)).to(dev), torch.from_numpy(k.view(np.int64)).to(dev)
    tQ = torch.empty_like(tP)

    def dev_sm():
        eng._call("zc_ed_scalar_mul", tP.data_ptr(), tk.data_ptr(), tQ.data_ptr(), n, 0)
        eng.synchronize()

    t_dev = best(dev_sm)
    ref = tQ.cpu().numpy().view(np.uint64)
    out = np.empty_like(P)

    for chunks in ("auto", "1", "2", "3", "4", "6", "8", "12", "16", "24", "32"):
        if chunks == "auto":
            os.environ.pop("ZC_HOST_CHUNKS", None)
        else:
            os.environ["ZC_HOST_CHUNKS"] = chunks
        ce = z.Engine()                                            # knobs are read when a context is created

        def warm():
            ce._call("zc_ed_scalar_mul", P.ctypes.data, k.ctypes.data, out.ctypes.data, n, 0)

        def fresh():
            o = np.empty_like(P)
            ce._call("zc_ed_scalar_mul", P.ctypes.data, k.ctypes.data, o.ctypes.data, n, 0)

        tw, tf = best(warm), best(fresh)
        ce.close()
        assert np.array_equal(out, ref)
        print(json.dumps({"op": "ed_scalar_mul", "n": n, "chunks": chunks, "warm_ms": round(tw * 1e3, 2),
                          "fresh_ms": round(tf * 1e3, 2), "device_ms": round(t_dev * 1e3, 2),
                          "warm_units_per_s": round(n / tw)}))

    # Two device slots on the same GPU: the slots' uploads are issued by one worker thread each, so
    # they run side by side; pinned (zc_host_register) buffers copy asynchronously in any case.
    # A copy-bound op (fe_mul: 120 B over PCIe per 100 ns of kernel) shows the issue order best.
    os.environ.pop("ZC_HOST_CHUNKS", None)
    a, b = rand_scalars_np(n, 9, 252), rand_scalars_np(n, 10, 252)
    o = np.empty_like(a)
    for slots in (1, 2):
        e2 = z.Engine([0] * slots)
        for pinned in (False, True):
            bufs = (P, k, out, a, b, o)
            if pinned:
                for x in bufs:
                    z.Engine.host_register(x)
            t_sm = best(lambda: e2._call("zc_ed_scalar_mul", P.ctypes.data, k.ctypes.data, out.ctypes.data, n, 0))
            assert np.array_equal(out, ref)
            t_fe = best(lambda: e2._call("zc_fe_mul", a.ctypes.data, b.ctypes.data, o.ctypes.data, n))
            if pinned:
                for x in bufs:
                    z.Engine.host_unregister(x)
            print(json.dumps({"slots": slots, "host_memory": "pinned (zc_host_register)" if pinned else "pageable", "n": n,
                              "ed_scalar_mul_ms": round(t_sm * 1e3, 2), "fe_mul_ms": round(t_fe * 1e3, 2),
                              "fe_mul_GBps_over_pcie": round(120 * n / t_fe / 1e9, 1)}))
        e2.close()


if __name__ == "__main__":
    main()
