"""Build libzerocaf_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzerocaf_hip.so")
SOURCES = ["zerocaf_hip.hip"]
DEPS = ["zerocaf_hip.hip", "zc_kernels.hip.h", "zc_msm.hip.h", "zc_sort.hip.h", "zc_quad.hip.h", "zc_curve.hip.h", "zc_arith.hip.h", "zc_constants.hip.h",
        os.path.join("..", "..", "include", "zerocaf_hip.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 library)")


def sources_sha256() -> str:
    """sha256 over the kernel sources and the ABI header (DEPS, in order): identifies what a library was
    built from independently of the build (profiles/roofline_inputs.json is keyed by it)."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(open(os.path.join(CSRC, d), "rb").read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


UBENCH_LIB = os.path.join(HERE, "libzc_ubench.so")


def build_ubench(force: bool = False, verbose: bool = False) -> str:
    """libzc_ubench.so: the live instruction-rate measurement bench.py prices its roofline against
    (a measurement aid, not part of the product library)."""
    src = os.path.join(CSRC, "zc_ubench.hip")
    if not force and os.path.exists(UBENCH_LIB) and os.path.getmtime(UBENCH_LIB) >= os.path.getmtime(src):
        return UBENCH_LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", UBENCH_LIB, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return UBENCH_LIB


TEST_LIB = os.path.join(HERE, "libzerocaf_hip_test.so")


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, test_hooks: bool = False) -> str:
    """libzerocaf_hip.so (the product) and, with test_hooks=True (the test tier, __graft_entry__.build() and the command
    line unless `--product-only`), libzerocaf_hip_test.so: the same sources with -DZC_TEST_HOOKS -- the fault
    injection knobs, the path forcers and the sort-stage hook the GPU test tier uses; never loaded by the product
    path.  The two hipcc runs go side by side."""
    build_ubench(force, verbose)
    jobs = []
    for lib, extra in ((LIB, []), (TEST_LIB, ["-DZC_TEST_HOOKS"]))[:2 if test_hooks else 1]:
        if not force and not _stale(lib):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib,
               '-DZC_SRC_HASH="%s"' % sources_sha256()] + extra + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return LIB


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """A/B builds of the same ABI with other compile-time knobs: build/variants/<name>.so
    (load with ZC_LIB_PATH).  Not part of the product; build/ is git-ignored."""
    out_dir = os.path.join(os.path.dirname(HERE), "build", "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", out,
           '-DZC_SRC_HASH="variant:%s"' % name] + ["-D" + d for d in defines] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
        sys.exit(0)
    # both libraries by default (the test tier refuses a twin built from other sources); --product-only skips the twin
    print(build(force="--force" in sys.argv, verbose=True, test_hooks="--product-only" not in sys.argv))
