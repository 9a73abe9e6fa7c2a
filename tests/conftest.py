import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The multi-rank tests run several ranks as handles of ONE process (a deployment runs one process per GPU): a rank's merge
# kernel spins until the peers' kernels have posted, so every in-process rank needs a hardware queue of its own.  The HIP
# runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES queues (default 4) — with four in-process ranks plus whatever
# an earlier test's library (RCCL) keeps alive that is one too few now and then.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
