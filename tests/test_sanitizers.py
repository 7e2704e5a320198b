"""CPU tier: the sanitizers SURVEY section 5 asks for, actually run (`make -C oracle asan` was a target
nobody ran).  The oracle (`oracle/zc_ref.c`, gcc -fsanitize=address,undefined) and the host build of the DEVICE arithmetic
headers (`tests/emul`, g++ with the same flags and -fno-sanitize-recover=all) replay, in a child interpreter with the
sanitizer runtimes preloaded, the reference's known-answer tests, the emulation tier (raw 260-bit scalars, off-curve
points, junk encodings, every lazy-reduction bound) and the one-pass product's edge values: any out-of-bounds access,
signed overflow, misaligned or out-of-range shift aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_and_device_headers_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libzc_ref_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ZC_REF_SO=so, ZC_EMUL_SANITIZE="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_device_arith_emul.py")
                          ],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and " passed" in out.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
    # the sanitized libraries were the ones in use (not silently the regular builds)
    assert os.path.exists(os.path.join(ROOT, "tests", "emul", "libzc_emul_san.so")) and os.path.exists(so)
