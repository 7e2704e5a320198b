"""
ctypes loader for the CPU oracle (oracle/libzc_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by dusk_zerocaf_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzc_ref.so")
# tests/test_sanitizers.py: load another build of the same file (`make -C oracle asan`) instead -- never rebuilt here
_SO_OVERRIDE = os.environ.get("ZC_REF_SO")


BASE_FLAGS = "-O3 -fPIC -std=c11 -Wall -Wextra"
# The CPU baseline must not be understated by its build: SURVEY 8(d) asks for -march=native, and on the Xeon of the build
# container that is the faster build (+8 %); on the EPYC 9575F of the GPU boxes gcc 11 does not know the core and its
# -march=native code is 13 % SLOWER than the generic x86-64-v2 build (tools/debug/cpu_flags_probe.py: 5238 against 6010
# scalar-muls/s on one thread).  So build() compiles both ON THE HOST THAT RUNS THE ORACLE, times a small batch of
# double_and_add calls on each and keeps the faster; the choice is recorded beside the library and quoted in
# bench.py's cpu_baseline.sample.
CANDIDATE_MARCH = ("native", "x86-64-v2")


def _host_id() -> str:
    """Identifies the host the library was built FOR: the in-tree .so travels to the GPU box with the snapshot, so
    build() rebuilds (and re-chooses the flags) when the CPU model differs."""
    import hashlib
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return "%s|%s" % (model, hashlib.sha256(flags.encode()).hexdigest()[:16])


def _read_stamp():
    try:
        host, flags = open(_SO + ".host").read().split("\n")[:2]
        return host, flags
    except (OSError, ValueError):
        return None, None


def build_flags() -> str:
    """The compiler flags of the library in use (chosen by build() on this host)."""
    return _read_stamp()[1] or (BASE_FLAGS + " -march=native")


def _time_candidate(so: str) -> float:
    lib_ = C.CDLL(so)
    n = 192
    rng = np.random.default_rng(7)
    k = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
    k[:, 4] >>= np.uint64(8)
    base = np.tile(np.array([276718085098056, 1646536057461434, 2704687245600312, 2630386667454967, 13476148227069,
                             1303868825475266, 3250718520537114, 2702159777242978, 2702159776422297, 10555311626649, 1, 0, 0, 0, 0,
                             3634527586288175, 2006028620404053, 3424252198034825, 2478951925947079, 4567251727358], dtype=np.uint64), (n, 1))
    out = np.empty_like(base)
    best = 1e9
    import time
    for _ in range(3):
        t = time.perf_counter()
        lib_.zr_ed_scalar_mul_batch(base.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        best = min(best, time.perf_counter() - t)
    return best


def build(force: bool = False) -> str:
    """Compile oracle/libzc_ref.so on THIS host unless an up-to-date build for this very CPU is already there: both
    candidate -march settings, the faster one kept.  Safe against concurrent callers (ranks of one job): lock file +
    atomic rename."""
    import fcntl
    src, stamp_path = os.path.join(_HERE, "zc_ref.c"), _SO + ".host"
    want = _host_id()

    def fresh() -> bool:
        try:
            return (os.path.getmtime(_SO) >= os.path.getmtime(src) and os.path.getmtime(_SO) >= os.path.getmtime(os.path.join(_HERE, "zc_ref.h"))
                    and _read_stamp()[0] == want)
        except OSError:
            return False
    if not force and fresh():
        return _SO
    with open(_SO + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            built = []
            for march in CANDIDATE_MARCH:
                flags = "%s -march=%s" % (BASE_FLAGS, march)
                tmp = "%s.%d.%s.tmp" % (_SO, os.getpid(), march)
                try:
                    subprocess.check_call(["make", "-B", "-C", _HERE, "libzc_ref.so", "OUT=" + tmp, "CFLAGS=" + flags], stdout=subprocess.DEVNULL)
                    built.append((_time_candidate(tmp), flags, tmp))
                except (subprocess.CalledProcessError, OSError):
                    pass
            if not built:
                raise RuntimeError("oracle: gcc failed for every candidate build")
            built.sort()
            os.replace(built[0][2], _SO)
            for _, _, tmp in built[1:]:
                os.remove(tmp)
            with open(stamp_path + ".tmp", "w") as f:
                f.write("%s\n%s\n%s\n" % (want, built[0][1], "; ".join("%s: %.1f ms per 192 double_and_add" % (fl.rsplit(" ", 1)[1], t * 1e3) for t, fl, _ in built)))
            os.replace(stamp_path + ".tmp", stamp_path)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if _SO_OVERRIDE:
            _lib = C.CDLL(_SO_OVERRIDE)
        else:
            build()                                              # rebuilds when the .so was made for another CPU model
            _lib = C.CDLL(_SO)
    return _lib


def _u64(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == width, a.shape
    return a


def _u8(a, width=32):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2 and a.shape[1] == width, a.shape
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _binop(name, width):
    def f(a, b):
        a, b = _u64(a, width), _u64(b, width)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
        return out
    return f


def _unop(name, width):
    def f(a):
        a = _u64(a, width)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(out), C.c_size_t(a.shape[0]))
        return out
    return f


fe_add = _binop("zr_fe_add_batch", 5)
fe_sub = _binop("zr_fe_sub_batch", 5)
fe_mul = _binop("zr_fe_mul_batch", 5)
fe_neg = _unop("zr_fe_neg_batch", 5)
fe_square = _unop("zr_fe_square_batch", 5)
sc_add = _binop("zr_sc_add_batch", 5)
sc_sub = _binop("zr_sc_sub_batch", 5)
sc_mul = _binop("zr_sc_mul_batch", 5)
sc_neg = _unop("zr_sc_neg_batch", 5)
sc_square = _unop("zr_sc_square_batch", 5)
ed_add = _binop("zr_ed_add_batch", 20)
ed_sub = _binop("zr_ed_sub_batch", 20)
ed_double = _unop("zr_ed_double_batch", 20)
ed_neg = _unop("zr_ed_neg_batch", 20)


fe_pow = _binop("zr_fe_pow_batch", 5)
fe_half = _unop("zr_fe_half_batch", 5)


def fe_div(a, b):
    a, b = _u64(a, 5), _u64(b, 5)
    out = np.empty_like(a)
    ok = np.empty(a.shape[0], dtype=np.uint8)
    lib().zr_fe_div_batch(_p(a), _p(b), _p(out), _p(ok), C.c_size_t(a.shape[0]))
    return out, ok


def _fe_flag(name):
    def f(a):
        a = _u64(a, 5)
        out = np.empty(a.shape[0], dtype=np.uint8)
        getattr(lib(), name)(_p(a), _p(out), C.c_size_t(a.shape[0]))
        return out
    return f


fe_legendre_symbol = _fe_flag("zr_fe_legendre_symbol_batch")
fe_is_positive = _fe_flag("zr_fe_is_positive_batch")


def fe_mod_sqrt(a, sign):
    a = _u64(a, 5)
    out = np.empty_like(a)
    ok = np.empty(a.shape[0], dtype=np.uint8)
    lib().zr_fe_mod_sqrt_batch(_p(a), C.c_int(sign), _p(out), _p(ok), C.c_size_t(a.shape[0]))
    return out, ok


def fe_invert(a):
    a = _u64(a, 5)
    out = np.empty_like(a)
    ok = np.empty(a.shape[0], dtype=np.uint8)
    lib().zr_fe_invert_batch(_p(a), _p(out), _p(ok), C.c_size_t(a.shape[0]))
    return out, ok


def fe_from_bytes(b):
    b = _u8(b)
    out = np.empty((b.shape[0], 5), dtype=np.uint64)
    lib().zr_fe_from_bytes_batch(_p(b), _p(out), C.c_size_t(b.shape[0]))
    return out


def fe_to_bytes(a):
    a = _u64(a, 5)
    out = np.empty((a.shape[0], 32), dtype=np.uint8)
    lib().zr_fe_to_bytes_batch(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def fe_sqrt_ratio_i(u, v):
    u, v = _u64(u, 5), _u64(v, 5)
    out = np.empty_like(u)
    sq = np.empty(u.shape[0], dtype=np.uint8)
    lib().zr_fe_sqrt_ratio_i_batch(_p(u), _p(v), _p(out), _p(sq), C.c_size_t(u.shape[0]))
    return out, sq


sc_half = _unop("zr_sc_half_batch", 5)
sc_pow = _binop("zr_sc_pow_batch", 5)


def sc_shr(a, shift):
    a = _u64(a, 5)
    out = np.empty_like(a)
    lib().zr_sc_shr_batch(_p(a), C.c_uint(int(shift)), _p(out), C.c_size_t(a.shape[0]))
    return out


def sc_into_bits(a):
    a = _u64(a, 5)
    out = np.empty((a.shape[0], 256), dtype=np.uint8)
    lib().zr_sc_into_bits_batch(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def sc_compute_naf(a, width=0):
    a = _u64(a, 5)
    out = np.empty((a.shape[0], 256), dtype=np.int8)
    lib().zr_sc_compute_naf_batch(_p(a), C.c_uint(int(width)), _p(out), C.c_size_t(a.shape[0]))
    return out


def fe_inv_sqrt(a):
    a = _u64(a, 5)
    out = np.empty_like(a)
    sq = np.empty(a.shape[0], dtype=np.uint8)
    lib().zr_fe_inv_sqrt_batch(_p(a), _p(out), _p(sq), C.c_size_t(a.shape[0]))
    return out, sq


def sc_from_bytes(b):
    b = _u8(b)
    out = np.empty((b.shape[0], 5), dtype=np.uint64)
    ok = np.empty(b.shape[0], dtype=np.uint8)
    lib().zr_sc_from_bytes_batch(_p(b), _p(out), _p(ok), C.c_size_t(b.shape[0]))
    return out, ok


def sc_to_bytes(a):
    a = _u64(a, 5)
    out = np.empty((a.shape[0], 32), dtype=np.uint8)
    lib().zr_sc_to_bytes_batch(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def ed_scalar_mul(p, k):
    p, k = _u64(p, 20), _u64(k, 5)
    assert p.shape[0] == k.shape[0]
    out = np.empty_like(p)
    lib().zr_ed_scalar_mul_batch(_p(p), _p(k), _p(out), C.c_size_t(p.shape[0]))
    return out


def ed_scalar_mul_mode(p, k, mode):
    """mode 0 double_and_add, 1 ltr_bin_mul, 2 binary_naf_mul."""
    p, k = _u64(p, 20), _u64(k, 5)
    out = np.empty_like(p)
    lib().zr_ed_scalar_mul_mode_batch(_p(p), _p(k), _p(out), C.c_size_t(p.shape[0]), C.c_int(mode))
    return out


def ed_mul_by_pow_2(p, kexp):
    p = _u64(p, 20)
    out = np.empty_like(p)
    lib().zr_ed_mul_by_pow_2_batch(_p(p), C.c_uint64(kexp), _p(out), C.c_size_t(p.shape[0]))
    return out


def ed_to_affine(p):
    p = _u64(p, 20)
    xy = np.empty((p.shape[0], 10), dtype=np.uint64)
    ok = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_ed_to_affine_batch(_p(p), _p(xy), _p(ok), C.c_size_t(p.shape[0]))
    return xy, ok


def ed_eq(p, q):
    p, q = _u64(p, 20), _u64(q, 20)
    eq = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_ed_eq_batch(_p(p), _p(q), _p(eq), C.c_size_t(p.shape[0]))
    return eq


def ed_compress(p):
    p = _u64(p, 20)
    out = np.empty((p.shape[0], 32), dtype=np.uint8)
    ok = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_ed_compress_batch(_p(p), _p(out), _p(ok), C.c_size_t(p.shape[0]))
    return out, ok


def ed_decompress(b):
    b = _u8(b)
    out = np.empty((b.shape[0], 20), dtype=np.uint64)
    ok = np.empty(b.shape[0], dtype=np.uint8)
    lib().zr_ed_decompress_batch(_p(b), _p(out), _p(ok), C.c_size_t(b.shape[0]))
    return out, ok


def ris_compress(p):
    p = _u64(p, 20)
    out = np.empty((p.shape[0], 32), dtype=np.uint8)
    lib().zr_ris_compress_batch(_p(p), _p(out), C.c_size_t(p.shape[0]))
    return out


def ris_decompress(b):
    b = _u8(b)
    out = np.empty((b.shape[0], 20), dtype=np.uint64)
    ok = np.empty(b.shape[0], dtype=np.uint8)
    lib().zr_ris_decompress_batch(_p(b), _p(out), _p(ok), C.c_size_t(b.shape[0]))
    return out, ok


def ris_eq(p, q):
    p, q = _u64(p, 20), _u64(q, 20)
    eq = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_ris_eq_batch(_p(p), _p(q), _p(eq), C.c_size_t(p.shape[0]))
    return eq


def ris_roundtrip_mul(b, k):
    b, k = _u8(b), _u64(k, 5)
    out = np.empty_like(b)
    ok = np.empty(b.shape[0], dtype=np.uint8)
    lib().zr_ris_roundtrip_mul_batch(_p(b), _p(k), _p(out), _p(ok), C.c_size_t(b.shape[0]))
    return out, ok


proj_add = _binop("zr_proj_add_batch", 15)
proj_double = _unop("zr_proj_double_batch", 15)
proj_neg = _unop("zr_proj_neg_batch", 15)
proj_sub = _binop("zr_proj_sub_batch", 15)


def proj_eq(p, q):
    """(eq, ok): ok = 0 where the reference's inverse() would panic (Z = 0)."""
    p, q = _u64(p, 15), _u64(q, 15)
    eq = np.empty(p.shape[0], dtype=np.uint8)
    ok = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_proj_eq_batch(_p(p), _p(q), _p(eq), _p(ok), C.c_size_t(p.shape[0]))
    return eq, ok


def proj_is_valid(p):
    p = _u64(p, 15)
    v = np.empty(p.shape[0], dtype=np.uint8)
    lib().zr_proj_is_valid_batch(_p(p), _p(v), C.c_size_t(p.shape[0]))
    return v


def proj_scalar_mul(p, k):
    p, k = _u64(p, 15), _u64(k, 5)
    out = np.empty_like(p)
    lib().zr_proj_scalar_mul_batch(_p(p), _p(k), _p(out), C.c_size_t(p.shape[0]))
    return out


def ed_coset4(p):
    p = _u64(p, 20)
    out = np.empty((p.shape[0], 80), dtype=np.uint64)
    lib().zr_ed_coset4_batch(_p(p), _p(out), C.c_size_t(p.shape[0]))
    return out


def proj_to_extended(p):
    p = _u64(p, 15)
    out = np.empty((p.shape[0], 20), dtype=np.uint64)
    lib().zr_proj_to_extended_batch(_p(p), _p(out), C.c_size_t(p.shape[0]))
    return out


def _valid(name):
    def f(p):
        p = _u64(p, 20)
        v = np.empty(p.shape[0], dtype=np.uint8)
        getattr(lib(), name)(_p(p), _p(v), C.c_size_t(p.shape[0]))
        return v
    return f


ed_is_valid = _valid("zr_ed_is_valid_batch")
ris_is_valid = _valid("zr_ris_is_valid_batch")


def ris_elligator(r0):
    r0 = _u64(r0, 5)
    out = np.empty((r0.shape[0], 20), dtype=np.uint64)
    lib().zr_ris_elligator_batch(_p(r0), _p(out), C.c_size_t(r0.shape[0]))
    return out


def ris_from_uniform_bytes(b):
    b = _u8(b, 64)
    out = np.empty((b.shape[0], 20), dtype=np.uint64)
    lib().zr_ris_from_uniform_bytes_batch(_p(b), _p(out), C.c_size_t(b.shape[0]))
    return out


def msm_naive(p, k):
    p, k = _u64(p, 20), _u64(k, 5)
    out = np.empty((1, 20), dtype=np.uint64)
    lib().zr_msm_naive(_p(p), _p(k), C.c_size_t(p.shape[0]), _p(out))
    return out


# ---- the same batch functions on all host cores (ctypes drops the GIL during a call) ----------
def host_threads() -> int:
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = min(c, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, c)


def mt(fn, *arrays, threads=None, extra=()):
    """fn(*slices, *extra) over contiguous slices of the batch, one slice per thread; results (an
    array or a tuple of arrays) are concatenated in order -- identical to one call over the batch."""
    import concurrent.futures as cf
    n = len(arrays[0])
    t = max(1, min(threads or host_threads(), n))
    if t == 1:
        return fn(*arrays, *extra)
    bounds = [n * i // t for i in range(t + 1)]
    with cf.ThreadPoolExecutor(max_workers=t) as ex:
        parts = list(ex.map(lambda i: fn(*[a[bounds[i]:bounds[i + 1]] for a in arrays], *extra), range(t)))
    if isinstance(parts[0], tuple):
        return tuple(np.concatenate([p[j] for p in parts]) for j in range(len(parts[0])))
    return np.concatenate(parts)


def msm_naive_mt(p, k, threads=None):
    """sum_i k_i * P_i with the reference's own ops: per-thread partial sums over contiguous slices
    (zr_msm_naive), folded in slice order with the unified add.  Same group element as the one-thread
    sum (compare with ed_eq / encodings; the (X:Y:Z:T) limbs depend on the association)."""
    n = len(p)
    t = max(1, min(threads or host_threads(), n))
    bounds = [n * i // t for i in range(t + 1)]
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=t) as ex:
        parts = list(ex.map(lambda i: msm_naive(p[bounds[i]:bounds[i + 1]], k[bounds[i]:bounds[i + 1]]), range(t)))
    acc = parts[0]
    for q in parts[1:]:
        acc = ed_add(acc, q)
    return acc


# ---- single-element helpers used by the KAT tests (lists of python ints) ----
class _FE(C.Structure):
    _fields_ = [("l", C.c_uint64 * 5)]


class _PT(C.Structure):
    _fields_ = [("X", _FE), ("Y", _FE), ("Z", _FE), ("T", _FE)]


def _fe(l):
    return _FE((C.c_uint64 * 5)(*[int(x) for x in l]))


def _pt(p):
    return _PT(*[_fe(c) for c in p])


def _fl(f):
    return [int(x) for x in f.l]


def _pl(p):
    return [_fl(p.X), _fl(p.Y), _fl(p.Z), _fl(p.T)]


def call_fe(name, *args):
    """r = name(fe args...) for functions of the form void f(zr_fe *r, const zr_fe*...)."""
    r = _FE()
    rc = getattr(lib(), name)(C.byref(r), *[C.byref(_fe(a)) if isinstance(a, (list, tuple)) else a for a in args])
    return _fl(r), rc


def call_pt(name, *args):
    r = _PT()
    conv = []
    for a in args:
        if isinstance(a, (list, tuple)) and len(a) == 4 and isinstance(a[0], (list, tuple)):
            conv.append(C.byref(_pt(a)))
        elif isinstance(a, (list, tuple)):
            conv.append(C.byref(_fe(a)))
        else:
            conv.append(a)
    rc = getattr(lib(), name)(C.byref(r), *conv)
    return _pl(r), rc
