import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libmppi_amd.so, built in-tree if needed (hipcc cross-compiles without a GPU)."""
    import mppi_generic_amd as m
    return m.load_library()


@pytest.fixture(scope="session")
def gpu(lib):
    """Fails loudly (never skips silently to a fallback) when a gpu-marked test runs without a device."""
    n = lib.mppi_device_count()
    assert n > 0, "gpu-marked test but no HIP device is visible; the product has no CPU path"
    return n
