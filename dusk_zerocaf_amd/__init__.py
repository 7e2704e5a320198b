"""dusk_zerocaf_amd -- MI355X (gfx950) batched backend for dusk-zerocaf's hot path.

Everything numeric happens in libzerocaf_hip.so (hand-written HIP kernels behind the C ABI
of include/zerocaf_hip.h).  This package only loads it and mirrors the reference's
operator surface for batches.  No CPU fallback exists.
"""
from ._lib import ZerocafHipError, load, LIB_PATH, ALL_SYMBOLS  # noqa: F401
from .engine import Engine, STRICT, LTR_BIN, BINARY_NAF, FAST  # noqa: F401

__all__ = ["Engine", "STRICT", "LTR_BIN", "BINARY_NAF", "FAST", "ZerocafHipError", "load", "LIB_PATH", "ALL_SYMBOLS"]
