"""GPU tier: the C++ host-side mirror (dusk_zerocaf_amd/include/zerocaf.hpp) reproduces the
reference's own unit-test assertions through the C ABI (tests/cpp/test_zerocaf_hpp.cpp)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_reference_style_checks():
    src = os.path.join(ROOT, "tests", "cpp", "test_zerocaf_hpp.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_zerocaf_hpp")
    libdir = os.path.join(ROOT, "dusk_zerocaf_amd")
    hpp = os.path.join(ROOT, "dusk_zerocaf_amd", "include", "zerocaf.hpp")
    newest = max(os.path.getmtime(src), os.path.getmtime(hpp))
    if shutil.which("g++") and (not os.path.exists(exe) or os.path.getmtime(exe) < newest):
        subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe, "-L", libdir, "-lzerocaf_hip",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all reference-style checks passed" in out.stdout
