"""ctypes binding of libzerocaf_hip.so (the C ABI in include/zerocaf_hip.h).

The product path: there is no Python or CPU fallback here.  If the shared library is
missing, loading raises; if no GPU is visible, zc_ctx_create fails.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ZC_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("ZC_LIB_PATH") or os.path.join(HERE, "libzerocaf_hip.so")

_u64p = C.c_void_p
_u8p = C.c_void_p
_ctx = C.c_void_p
_n = C.c_size_t

# name -> argtypes (after the leading zc_ctx*); mirrors include/zerocaf_hip.h
SIGNATURES = {
    "zc_fe_add": [_u64p, _u64p, _u64p, _n],
    "zc_fe_sub": [_u64p, _u64p, _u64p, _n],
    "zc_fe_neg": [_u64p, _u64p, _n],
    "zc_fe_mul": [_u64p, _u64p, _u64p, _n],
    "zc_fe_square": [_u64p, _u64p, _n],
    "zc_fe_invert": [_u64p, _u64p, _u8p, _n],
    "zc_fe_div": [_u64p, _u64p, _u64p, _u8p, _n],
    "zc_fe_half": [_u64p, _u64p, _n],
    "zc_fe_pow": [_u64p, _u64p, _u64p, _n],
    "zc_fe_legendre_symbol": [_u64p, _u8p, _n],
    "zc_fe_is_positive": [_u64p, _u8p, _n],
    "zc_fe_mod_sqrt": [_u64p, C.c_int, _u64p, _u8p, _n],
    "zc_fe_from_bytes": [_u8p, _u64p, _n],
    "zc_fe_to_bytes": [_u64p, _u8p, _n],
    "zc_fe_sqrt_ratio_i": [_u64p, _u64p, _u64p, _u8p, _n],
    "zc_sc_add": [_u64p, _u64p, _u64p, _n],
    "zc_sc_sub": [_u64p, _u64p, _u64p, _n],
    "zc_sc_neg": [_u64p, _u64p, _n],
    "zc_sc_mul": [_u64p, _u64p, _u64p, _n],
    "zc_sc_square": [_u64p, _u64p, _n],
    "zc_fe_inv_sqrt": [_u64p, _u64p, _u8p, _n],
    "zc_sc_half": [_u64p, _u64p, _n],
    "zc_sc_pow": [_u64p, _u64p, _u64p, _n],
    "zc_sc_shr": [_u64p, C.c_uint, _u64p, _n],
    "zc_sc_into_bits": [_u64p, _u8p, _n],
    "zc_sc_compute_naf": [_u64p, C.c_uint, _u8p, _n],
    "zc_sc_from_bytes": [_u8p, _u64p, _u8p, _n],
    "zc_sc_to_bytes": [_u64p, _u8p, _n],
    "zc_ed_add": [_u64p, _u64p, _u64p, _n],
    "zc_ed_sub": [_u64p, _u64p, _u64p, _n],
    "zc_ed_double": [_u64p, _u64p, _n],
    "zc_ed_neg": [_u64p, _u64p, _n],
    "zc_ed_scalar_mul": [_u64p, _u64p, _u64p, _n, C.c_uint],
    "zc_ed_mul_by_pow_2": [_u64p, C.c_uint64, _u64p, _n],
    "zc_ed_mul_by_cofactor": [_u64p, _u64p, _n],
    "zc_ed_to_affine": [_u64p, _u64p, _u8p, _n],
    "zc_ed_eq": [_u64p, _u64p, _u8p, _n],
    "zc_ed_compress": [_u64p, _u8p, _u8p, _n],
    "zc_ed_decompress": [_u8p, _u64p, _u8p, _n],
    "zc_ris_compress": [_u64p, _u8p, _n],
    "zc_ris_decompress": [_u8p, _u64p, _u8p, _n],
    "zc_ris_eq": [_u64p, _u64p, _u8p, _n],
    "zc_ris_roundtrip_mul": [_u8p, _u64p, _u8p, _u8p, _n],
    "zc_ed_is_valid": [_u64p, _u8p, _n],
    "zc_ris_is_valid": [_u64p, _u8p, _n],
    "zc_ris_elligator": [_u64p, _u64p, _n],
    "zc_ris_from_uniform_bytes": [_u8p, _u64p, _n],
    "zc_proj_add": [_u64p, _u64p, _u64p, _n],
    "zc_proj_double": [_u64p, _u64p, _n],
    "zc_proj_to_extended": [_u64p, _u64p, _n],
    "zc_proj_neg": [_u64p, _u64p, _n],
    "zc_proj_sub": [_u64p, _u64p, _u64p, _n],
    "zc_proj_eq": [_u64p, _u64p, _u8p, _n],
    "zc_proj_is_valid": [_u64p, _u8p, _n],
    "zc_proj_scalar_mul": [_u64p, _u64p, _u64p, _n],
    "zc_ed_coset4": [_u64p, _u64p, _n],
    "zc_ed_mul_base": [_u64p, _u64p, _n],
    "zc_ris_mul_base_compress": [_u64p, _u8p, _n],
    "zc_ed_mul_base_wnaf": [_u64p, C.c_uint, _u64p, _n],
    "zc_msm": [_u64p, _u64p, _n, _u64p],
    "zc_msm_partial": [_u64p, _u64p, _n, _u64p],
    "zc_msm_plan": [_n, C.c_int, C.POINTER(C.c_int32), C.c_int],
    "zc_ed_fold_ordered": [_u64p, _n, _u64p],
    "zc_msm_sharded": [_u64p, _u64p, _n, _u64p],
    "zc_comm_init": [_u8p, C.c_int, C.c_int],
    "zc_comm_destroy": [],
    "zc_comm_size": [C.POINTER(C.c_int)],
    "zc_ctx_set_stream_dev": [C.c_int, C.c_void_p, C.c_int],
}
CONTEXT_SYMBOLS = ["zc_ctx_create", "zc_ctx_destroy", "zc_ctx_device", "zc_ctx_device_count", "zc_ctx_set_stream", "zc_ctx_synchronize",
                   "zc_device_count", "zc_last_error", "zc_version", "zc_host_register", "zc_host_unregister",
                   "zc_comm_unique_id"]
ALL_SYMBOLS = CONTEXT_SYMBOLS + list(SIGNATURES)

_lib = None


class ZerocafHipError(RuntimeError):
    pass


def _share_hip_runtime_with_torch() -> None:
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, the one our library needs); if our library
    pulled in /opt/rocm's copy first, a later `import torch` would bring a second runtime
    that sees no GPU.  So when torch is installed, map its copy before ours."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def _bind(path: str) -> C.CDLL:
    lib = C.CDLL(path)
    lib.zc_ctx_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(_ctx)]
    lib.zc_ctx_create.restype = C.c_int
    lib.zc_ctx_destroy.argtypes = [_ctx]
    lib.zc_ctx_device.argtypes = [_ctx, C.c_int]
    lib.zc_ctx_device_count.argtypes = [_ctx]
    lib.zc_ctx_set_stream.argtypes = [_ctx, C.c_void_p, C.c_int]
    lib.zc_ctx_synchronize.argtypes = [_ctx]
    lib.zc_device_count.restype = C.c_int
    lib.zc_last_error.restype = C.c_char_p
    lib.zc_version.restype = C.c_char_p
    lib.zc_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.zc_host_unregister.argtypes = [C.c_void_p]
    lib.zc_comm_unique_id.argtypes = [C.c_void_p]
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = [_ctx] + sig
        fn.restype = C.c_int
    return lib


def load() -> C.CDLL:
    """Load the HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZerocafHipError(
            "libzerocaf_hip.so is missing (%s): build it with `python -m dusk_zerocaf_amd.build`; "
            "there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    _lib = _bind(LIB_PATH)
    return _lib


TEST_LIB_PATH = os.path.join(HERE, "libzerocaf_hip_test.so")
_test_lib = None


def load_test_hooks() -> C.CDLL:
    """libzerocaf_hip_test.so: the same sources built with -DZC_TEST_HOOKS (fault injection for the table ring,
    the MSM's sort stage on its own).  For tests/ only -- nothing in the product path loads it."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise ZerocafHipError("libzerocaf_hip_test.so is missing: build it with `python -m dusk_zerocaf_amd.build`")
        _share_hip_runtime_with_torch()
        lib = _bind(TEST_LIB_PATH)
        # the twin must come from the very sources the product was built from (both embed their sha256): after a source
        # edit and a product-only rebuild the GPU tests would otherwise run a stale binary -- possibly with another ABI
        src = lambda l: l.zc_version().decode().rsplit("src:", 1)[-1]
        if os.path.exists(LIB_PATH) and src(lib) != src(load()):
            raise ZerocafHipError("libzerocaf_hip_test.so (src:%s) was not built from the product's sources (src:%s): "
                                  "rebuild both with `python -m dusk_zerocaf_amd.build`" % (src(lib)[:12], src(load())[:12]))
        _test_lib = lib
        _test_lib.zc_test_msm_sort.argtypes = [_ctx, _u64p, _n, C.c_int, C.c_void_p]
        _test_lib.zc_test_msm_sort.restype = C.c_int
        _test_lib.zc_test_odd_table.argtypes = [_ctx, C.c_void_p]
        _test_lib.zc_test_odd_table.restype = C.c_int
        _test_lib.zc_test_staged_launches.argtypes = [_ctx]
        _test_lib.zc_test_staged_launches.restype = C.c_longlong
    return _test_lib


def check(rc: int, what: str, lib=None) -> None:
    if rc != 0:
        raise ZerocafHipError("%s failed: status %d (%s)" % (what, rc, (lib or load()).zc_last_error().decode()))
