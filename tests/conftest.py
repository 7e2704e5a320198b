import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible() -> bool:
    try:
        import dusk_zerocaf_amd as z
        return z.load().zc_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu tier is skipped, not failed (there is no CPU fallback
    to run it on).  With `-m gpu` on the GPU box nothing is skipped."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and not _gpu_visible():
        skip = pytest.mark.skip(reason="no HIP device visible: the gpu tier needs a real MI355X")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libzc_ref.so) -- the checker, never the product."""
    from oracle import zc_ref
    zc_ref.build()
    zc_ref.lib()
    return zc_ref
