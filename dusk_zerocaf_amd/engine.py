"""Batch engine: the host-side mirror of zerocaf's field / scalar / edwards / ristretto
operator surface over the C ABI (include/zerocaf_hip.h).

Arrays use the reference's in-memory layout: FieldElement / Scalar = (n, 5) uint64 limbs
(radix 2^52), EdwardsPoint = (n, 20) uint64 (X|Y|Z|T), encodings = (n, 32) uint8.
Inputs may be numpy arrays (host memory: staged over PCIe by the library) or torch CUDA
tensors (device memory: used in place on the engine's stream, asynchronous).  Outputs are
of the same kind as the first input.  All arithmetic runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

STRICT = 0          # double_and_add (Mul<Scalar>)
LTR_BIN = 1         # ltr_bin_mul
BINARY_NAF = 2      # binary_naf_mul
FAST = 16           # same group element, not limb-exact (windowed, dedicated doubling)


def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "device")


class Engine:
    """One zc_ctx.  `devices=None` = one slot on torch's current device when torch is loaded and sees a GPU; otherwise
    zc_ctx_create(NULL, 0): the calling thread's current HIP device (whatever hipSetDevice chose), read back from the
    context.  The context reads the library's tuning knobs (ZC_* environment variables)
    once, here.  `lib`: another build of the same ABI (tests: the ZC_TEST_HOOKS build)."""

    def __init__(self, devices=None, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.ctx = C.c_void_p()
        self._pinned_stream = False      # set_stream() was called: keep that stream
        self._last_torch_stream = {}     # device slot -> handle of the torch stream last bound to it
        if not devices:
            import sys
            torch = sys.modules.get("torch")
            devices = [torch.cuda.current_device()] if torch is not None and torch.cuda.is_available() else []
        arr = (C.c_int * len(devices))(*devices) if devices else None
        rc = self.lib.zc_ctx_create(arr, len(devices), C.byref(self.ctx))
        _lib.check(rc, "zc_ctx_create", self.lib)
        # slot i of the context = HIP device _devices[i], as the context itself reports it: always known, so the
        # ownership check of _follow_torch_stream always runs
        self._devices = [self.lib.zc_ctx_device(self.ctx, i) for i in range(self.lib.zc_ctx_device_count(self.ctx))]
        if any(d < 0 for d in self._devices) or (devices and self._devices != list(devices)):
            got = self._devices
            self.close()
            raise _lib.ZerocafHipError("zc_ctx_create: the context reports devices %s, asked for %s" % (got, list(devices)))

    def close(self):
        if self.ctx:
            self.lib.zc_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        """Launch on the caller's HIP stream (handle 0 = the HIP null stream, which is what
        torch.cuda.current_stream().cuda_stream is by default)."""
        _lib.check(self.lib.zc_ctx_set_stream(self.ctx, C.c_void_p(stream_handle), 1), "zc_ctx_set_stream", self.lib)
        self._pinned_stream = True

    def use_own_stream(self):
        """Back to the default: host batches run on the context's own stream; calls on torch CUDA
        tensors follow torch's current stream of that device (see _follow_torch_stream)."""
        for slot in range(len(self._devices)):
            _lib.check(self.lib.zc_ctx_set_stream_dev(self.ctx, slot, None, 0), "zc_ctx_set_stream_dev", self.lib)
        self._pinned_stream = False
        self._last_torch_stream = {}

    def _follow_torch_stream(self, t):
        """Device tensors are produced and consumed on torch's streams and their memory belongs to
        torch's caching allocator, so unless the caller pinned a stream with set_stream() every
        call on torch tensors is enqueued on torch.cuda.current_stream(device): ordered after the
        work that produced the inputs, and outputs allocated with torch.empty are safe to use from
        that stream.  (The library orders a stream switch with an event, no host sync.)
        The stream is bound to the context slot that OWNS the tensor's device (a multi-device engine keeps
        one stream per slot); a tensor on a device outside the context is refused here, before any launch."""
        if self._pinned_stream:
            return
        import torch
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        if dev not in self._devices:
            raise _lib.ZerocafHipError("tensor on cuda:%d, but this engine's context owns devices %s" % (dev, self._devices))
        slot = self._devices.index(dev)
        h = torch.cuda.current_stream(t.device).cuda_stream
        if self._last_torch_stream.get(slot) != h:
            _lib.check(self.lib.zc_ctx_set_stream_dev(self.ctx, slot, C.c_void_p(h), 1), "zc_ctx_set_stream_dev", self.lib)
            self._last_torch_stream[slot] = h

    def synchronize(self):
        _lib.check(self.lib.zc_ctx_synchronize(self.ctx), "zc_ctx_synchronize", self.lib)

    # ------------------------------------------------------------------ helpers
    def _prep(self, x, width, dtype):
        if _is_torch(x):
            assert x.is_contiguous() and x.shape[-1] == width, (x.shape, width)
            assert x.element_size() == np.dtype(dtype).itemsize
            self._follow_torch_stream(x)
            return x, x.data_ptr(), x.shape[0]
        a = np.ascontiguousarray(x, dtype=dtype)
        assert a.ndim == 2 and a.shape[1] == width, (a.shape, width)
        return a, a.ctypes.data, a.shape[0]

    @staticmethod
    def _alloc(like, n, width, dtype):
        if _is_torch(like):
            import torch
            tdt = torch.uint8 if np.dtype(dtype) == np.uint8 else like.dtype if like.element_size() == 8 else torch.int64
            shape = (n, width) if width else (n,)
            t = torch.empty(shape, dtype=tdt, device=like.device)
            return t, t.data_ptr()
        shape = (n, width) if width else (n,)
        a = np.empty(shape, dtype=dtype)
        return a, a.ctypes.data

    def _call(self, name, *args):
        _lib.check(getattr(self.lib, name)(self.ctx, *args), name, self.lib)

    def _bin(self, name, a, b, w):
        a, pa, n = self._prep(a, w, np.uint64)
        b, pb, nb = self._prep(b, w, np.uint64)
        assert n == nb
        out, po = self._alloc(a, n, w, np.uint64)
        self._call(name, pa, pb, po, n)
        return out

    def _un(self, name, a, w):
        a, pa, n = self._prep(a, w, np.uint64)
        out, po = self._alloc(a, n, w, np.uint64)
        self._call(name, pa, po, n)
        return out

    # ------------------------------------------------------------------ FieldElement (field.rs)
    def fe_add(self, a, b): return self._bin("zc_fe_add", a, b, 5)
    def fe_sub(self, a, b): return self._bin("zc_fe_sub", a, b, 5)
    def fe_mul(self, a, b): return self._bin("zc_fe_mul", a, b, 5)
    def fe_neg(self, a): return self._un("zc_fe_neg", a, 5)
    def fe_square(self, a): return self._un("zc_fe_square", a, 5)

    def fe_invert(self, a):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 5, np.uint64)
        ok, pk = self._alloc(a, n, 0, np.uint8)
        self._call("zc_fe_invert", pa, po, pk, n)
        return out, ok

    def fe_div(self, a, b):
        a, pa, n = self._prep(a, 5, np.uint64)
        b, pb, _ = self._prep(b, 5, np.uint64)
        out, po = self._alloc(a, n, 5, np.uint64)
        ok, pk = self._alloc(a, n, 0, np.uint8)
        self._call("zc_fe_div", pa, pb, po, pk, n)
        return out, ok

    def fe_half(self, a): return self._un("zc_fe_half", a, 5)
    def fe_pow(self, a, e): return self._bin("zc_fe_pow", a, e, 5)

    def _fe_flag(self, name, a):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 0, np.uint8)
        self._call(name, pa, po, n)
        return out

    def fe_legendre_symbol(self, a): return self._fe_flag("zc_fe_legendre_symbol", a)
    def fe_is_positive(self, a): return self._fe_flag("zc_fe_is_positive", a)

    def fe_mod_sqrt(self, a, sign):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 5, np.uint64)
        ok, pk = self._alloc(a, n, 0, np.uint8)
        self._call("zc_fe_mod_sqrt", pa, C.c_int(int(sign)), po, pk, n)
        return out, ok

    def fe_from_bytes(self, b):
        b, pb, n = self._prep(b, 32, np.uint8)
        out, po = self._alloc_u64(b, n, 5)
        self._call("zc_fe_from_bytes", pb, po, n)
        return out

    def fe_to_bytes(self, a):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 32, np.uint8)
        self._call("zc_fe_to_bytes", pa, po, n)
        return out

    def fe_sqrt_ratio_i(self, u, v):
        u, pu, n = self._prep(u, 5, np.uint64)
        v, pv, _ = self._prep(v, 5, np.uint64)
        out, po = self._alloc(u, n, 5, np.uint64)
        sq, ps = self._alloc(u, n, 0, np.uint8)
        self._call("zc_fe_sqrt_ratio_i", pu, pv, po, ps, n)
        return out, sq

    def fe_inv_sqrt(self, a):
        """InvSqrt (field.rs:443-460): (1/sqrt(a) or sqrt(i/a), was_square)."""
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 5, np.uint64)
        sq, ps = self._alloc(a, n, 0, np.uint8)
        self._call("zc_fe_inv_sqrt", pa, po, ps, n)
        return out, sq

    def _alloc_u64(self, like, n, width):
        if _is_torch(like):
            import torch
            t = torch.empty((n, width), dtype=torch.int64, device=like.device)
            return t, t.data_ptr()
        a = np.empty((n, width), dtype=np.uint64)
        return a, a.ctypes.data

    # ------------------------------------------------------------------ Scalar (scalar.rs)
    def sc_add(self, a, b): return self._bin("zc_sc_add", a, b, 5)
    def sc_sub(self, a, b): return self._bin("zc_sc_sub", a, b, 5)
    def sc_mul(self, a, b): return self._bin("zc_sc_mul", a, b, 5)
    def sc_neg(self, a): return self._un("zc_sc_neg", a, 5)
    def sc_square(self, a): return self._un("zc_sc_square", a, 5)

    # the Scalar operations beside the default scalar-mul path (scalar.rs:165-182, 285-322, 352-415)
    def sc_half(self, a): return self._un("zc_sc_half", a, 5)
    def sc_pow(self, a, e): return self._bin("zc_sc_pow", a, e, 5)

    def sc_shr(self, a, shift):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 5, np.uint64)
        self._call("zc_sc_shr", pa, C.c_uint(int(shift)), po, n)
        return out

    def sc_into_bits(self, a):
        """into_bits: (n, 256) uint8, the bits of to_bytes(), least significant first."""
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 256, np.uint8)
        self._call("zc_sc_into_bits", pa, po, n)
        return out

    def sc_compute_naf(self, a, width=0):
        """compute_NAF (width 0) / compute_window_NAF(width 2..7): (n, 256) int8 digits."""
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 256, np.uint8)
        self._call("zc_sc_compute_naf", pa, C.c_uint(int(width)), po, n)
        if _is_torch(out):
            import torch
            return out.view(torch.int8)
        return out.view(np.int8)

    def sc_from_bytes(self, b):
        b, pb, n = self._prep(b, 32, np.uint8)
        out, po = self._alloc_u64(b, n, 5)
        ok, pk = self._alloc(b, n, 0, np.uint8)
        self._call("zc_sc_from_bytes", pb, po, pk, n)
        return out, ok

    def sc_to_bytes(self, a):
        a, pa, n = self._prep(a, 5, np.uint64)
        out, po = self._alloc(a, n, 32, np.uint8)
        self._call("zc_sc_to_bytes", pa, po, n)
        return out

    # ------------------------------------------------------------------ EdwardsPoint (edwards.rs)
    def ed_add(self, p, q): return self._bin("zc_ed_add", p, q, 20)
    def ed_sub(self, p, q): return self._bin("zc_ed_sub", p, q, 20)
    def ed_double(self, p): return self._un("zc_ed_double", p, 20)
    def ed_neg(self, p): return self._un("zc_ed_neg", p, 20)

    def ed_scalar_mul(self, p, k, out=None, flags=STRICT):
        p, pp, n = self._prep(p, 20, np.uint64)
        k, pk, nk = self._prep(k, 5, np.uint64)
        assert n == nk
        if out is None:
            out, po = self._alloc(p, n, 20, np.uint64)
        else:
            out, po, _ = self._prep(out, 20, np.uint64)
        self._call("zc_ed_scalar_mul", pp, pk, po, n, flags)
        return out

    def ed_mul_by_pow_2(self, p, kexp):
        p, pp, n = self._prep(p, 20, np.uint64)
        out, po = self._alloc(p, n, 20, np.uint64)
        self._call("zc_ed_mul_by_pow_2", pp, C.c_uint64(kexp), po, n)
        return out

    def ed_mul_by_cofactor(self, p):
        p, pp, n = self._prep(p, 20, np.uint64)
        out, po = self._alloc(p, n, 20, np.uint64)
        self._call("zc_ed_mul_by_cofactor", pp, po, n)
        return out

    def ed_to_affine(self, p):
        p, pp, n = self._prep(p, 20, np.uint64)
        xy, px = self._alloc(p, n, 10, np.uint64)
        ok, pk = self._alloc(p, n, 0, np.uint8)
        self._call("zc_ed_to_affine", pp, px, pk, n)
        return xy, ok

    def ed_eq(self, p, q):
        p, pp, n = self._prep(p, 20, np.uint64)
        q, pq, _ = self._prep(q, 20, np.uint64)
        eq, pe = self._alloc(p, n, 0, np.uint8)
        self._call("zc_ed_eq", pp, pq, pe, n)
        return eq

    def ed_compress(self, p):
        p, pp, n = self._prep(p, 20, np.uint64)
        out, po = self._alloc(p, n, 32, np.uint8)
        ok, pk = self._alloc(p, n, 0, np.uint8)
        self._call("zc_ed_compress", pp, po, pk, n)
        return out, ok

    def ed_decompress(self, b):
        b, pb, n = self._prep(b, 32, np.uint8)
        out, po = self._alloc_u64(b, n, 20)
        ok, pk = self._alloc(b, n, 0, np.uint8)
        self._call("zc_ed_decompress", pb, po, pk, n)
        return out, ok

    # ------------------------------------------------------------------ Ristretto (ristretto.rs)
    def ris_compress(self, p):
        p, pp, n = self._prep(p, 20, np.uint64)
        out, po = self._alloc(p, n, 32, np.uint8)
        self._call("zc_ris_compress", pp, po, n)
        return out

    def ris_decompress(self, b):
        b, pb, n = self._prep(b, 32, np.uint8)
        out, po = self._alloc_u64(b, n, 20)
        ok, pk = self._alloc(b, n, 0, np.uint8)
        self._call("zc_ris_decompress", pb, po, pk, n)
        return out, ok

    def ris_eq(self, p, q):
        p, pp, n = self._prep(p, 20, np.uint64)
        q, pq, _ = self._prep(q, 20, np.uint64)
        eq, pe = self._alloc(p, n, 0, np.uint8)
        self._call("zc_ris_eq", pp, pq, pe, n)
        return eq

    def ris_roundtrip_mul(self, b, k, out=None):
        b, pb, n = self._prep(b, 32, np.uint8)
        k, pk, _ = self._prep(k, 5, np.uint64)
        if out is None:
            out, po = self._alloc(b, n, 32, np.uint8)
        else:
            out, po, _ = self._prep(out, 32, np.uint8)
        ok, pko = self._alloc(b, n, 0, np.uint8)
        self._call("zc_ris_roundtrip_mul", pb, pk, po, pko, n)
        return out, ok

    # ------------------------------------------------------------------ next rows (N3, N4)
    def _flag(self, name, p):
        p, pp, n = self._prep(p, 20, np.uint64)
        v, pv = self._alloc(p, n, 0, np.uint8)
        self._call(name, pp, pv, n)
        return v

    def ed_is_valid(self, p): return self._flag("zc_ed_is_valid", p)
    def ris_is_valid(self, p): return self._flag("zc_ris_is_valid", p)

    def ris_elligator(self, r0):
        r0, pr, n = self._prep(r0, 5, np.uint64)
        out, po = self._alloc(r0, n, 20, np.uint64)
        self._call("zc_ris_elligator", pr, po, n)
        return out

    def ris_from_uniform_bytes(self, b):
        b, pb, n = self._prep(b, 64, np.uint8)
        out, po = self._alloc_u64(b, n, 20)
        self._call("zc_ris_from_uniform_bytes", pb, po, n)
        return out

    def proj_add(self, p, q): return self._bin("zc_proj_add", p, q, 15)
    def proj_double(self, p): return self._un("zc_proj_double", p, 15)

    def proj_to_extended(self, p):
        p, pp, n = self._prep(p, 15, np.uint64)
        out, po = self._alloc(p, n, 20, np.uint64)
        self._call("zc_proj_to_extended", pp, po, n)
        return out

    # ProjectivePoint beside add / double (edwards.rs:701-748, 787-912) and EdwardsPoint::coset4 (:603-610)
    def proj_neg(self, p): return self._un("zc_proj_neg", p, 15)
    def proj_sub(self, p, q): return self._bin("zc_proj_sub", p, q, 15)

    def proj_eq(self, p, q):
        p, pp, n = self._prep(p, 15, np.uint64)
        q, pq, _ = self._prep(q, 15, np.uint64)
        eq, pe = self._alloc(p, n, 0, np.uint8)
        self._call("zc_proj_eq", pp, pq, pe, n)
        return eq

    def proj_is_valid(self, p):
        p, pp, n = self._prep(p, 15, np.uint64)
        v, pv = self._alloc(p, n, 0, np.uint8)
        self._call("zc_proj_is_valid", pp, pv, n)
        return v

    def proj_scalar_mul(self, p, k):
        p, pp, n = self._prep(p, 15, np.uint64)
        k, pk, nk = self._prep(k, 5, np.uint64)
        assert n == nk
        out, po = self._alloc(p, n, 15, np.uint64)
        self._call("zc_proj_scalar_mul", pp, pk, po, n)
        return out

    def ed_coset4(self, p):
        """coset4: (n, 80) uint64 = four points per input point."""
        p, pp, n = self._prep(p, 20, np.uint64)
        out, po = self._alloc(p, n, 80, np.uint64)
        self._call("zc_ed_coset4", pp, po, n)
        return out

    # ------------------------------------------------------------------ fixed-base (key generation)
    def ed_mul_base(self, k):
        k, pk, n = self._prep(k, 5, np.uint64)
        out, po = self._alloc(k, n, 20, np.uint64)
        self._call("zc_ed_mul_base", pk, po, n)
        return out

    def ed_mul_base_wnaf(self, k, width):
        """window_naf_mul (edwards.rs:155-171) with the table indexed correctly, one launch; width 2..7."""
        k, pk, n = self._prep(k, 5, np.uint64)
        out, po = self._alloc(k, n, 20, np.uint64)
        self._call("zc_ed_mul_base_wnaf", pk, int(width), po, n)
        return out

    def msm_plan(self, n, points_aligned16=True):
        """What the bucket method would do for a shard of n pairs on this context (a query, no device work)."""
        v = (C.c_int32 * 17)()
        self._call("zc_msm_plan", int(n), 1 if points_aligned16 else 0, v, 17)
        g = v[7]
        return {"window_bits": v[0], "windows": v[1], "affine": bool(v[2]), "record_bytes": v[3], "record_stride": v[8], "run": v[4],
                "segment_buckets": v[5], "sort_passes": v[6], "window_groups": g,
                "group_windows": [v[9 + i] for i in range(g)], "group_runs": [v[13 + i] for i in range(g)]}

    def ris_mul_base_compress(self, k):
        k, pk, n = self._prep(k, 5, np.uint64)
        out, po = self._alloc(k, n, 32, np.uint8)
        self._call("zc_ris_mul_base_compress", pk, po, n)
        return out

    # ------------------------------------------------------------------ MSM (not in the reference)
    def msm(self, points, scalars):
        points, pp, n = self._prep(points, 20, np.uint64)
        scalars, pk, _ = self._prep(scalars, 5, np.uint64)
        out = np.empty((1, 20), dtype=np.uint64)
        self._call("zc_msm", pp, pk, n, out.ctypes.data)
        return out

    # ---- the exchange step of a sharded MSM (BASELINE configs[4])
    def msm_partial(self, points, scalars, out=None):
        """This device's sum left in device memory: a (1, 20) int64 torch CUDA tensor (asynchronous)."""
        import torch
        points, pp, n = self._prep(points, 20, np.uint64)
        scalars, pk, _ = self._prep(scalars, 5, np.uint64)
        if out is None:
            dev = points.device if _is_torch(points) else torch.device("cuda", torch.cuda.current_device())
            out = torch.empty((1, 20), dtype=torch.int64, device=dev)
            self._follow_torch_stream(out)
        self._call("zc_msm_partial", pp, pk, n, out.data_ptr())
        return out

    def ed_fold_ordered(self, parts):
        """((p_0 + p_1) + p_2) + ... in index order, one kernel launch; (count, 20) -> (1, 20)."""
        parts, pp, n = self._prep(parts, 20, np.uint64)
        out, po = self._alloc(parts, 1, 20, np.uint64)
        self._call("zc_ed_fold_ordered", pp, n, po)
        return out

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().zc_comm_unique_id(buf), "zc_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._call("zc_comm_init", buf, rank, world)

    def comm_destroy(self):
        self._call("zc_comm_destroy")

    def comm_size(self) -> int:
        """Ranks of the context's RCCL communicator as RCCL reports them (ncclCommCount); 0 without one."""
        r = C.c_int(0)
        self._call("zc_comm_size", C.byref(r))
        return r.value

    def msm_sharded(self, points, scalars):
        """This rank's shard of a global MSM through the library's own RCCL communicator
        (comm_init first): local bucket method, ncclAllGather of the 160-byte partial sums,
        ordered fold on the device.  Returns the global sum as a (1, 20) numpy array."""
        points, pp, n = self._prep(points, 20, np.uint64)
        scalars, pk, _ = self._prep(scalars, 5, np.uint64)
        out = np.empty((1, 20), dtype=np.uint64)
        self._call("zc_msm_sharded", pp, pk, n, out.ctypes.data)
        return out

    def set_stream_dev(self, slot, stream_handle):
        self._call("zc_ctx_set_stream_dev", slot, C.c_void_p(stream_handle), 1)
        self._pinned_stream = True

    @staticmethod
    def host_register(arr: np.ndarray):
        _lib.check(_lib.load().zc_host_register(arr.ctypes.data, arr.nbytes), "zc_host_register")

    @staticmethod
    def host_unregister(arr: np.ndarray):
        _lib.check(_lib.load().zc_host_unregister(arr.ctypes.data), "zc_host_unregister")
