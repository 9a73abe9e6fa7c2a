"""include/mppi_amd/plugin/parallel_utils.hpp against the reference's utils/parallel_utils.cuh:12-345: every Parallel1Dir
(THREAD_X .. THREAD_XYZ, GLOBAL_X/Y/Z, NONE) and Parallel2Dir direction and loadArrayParallel's runtime-count form, on a
standalone probe (tests/probes/parallel_index_probe.hip) a CPU test compiles and a gpu test runs."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "examples", "_build", "parallel_index_probe")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(REPO, "tests", "probes", "parallel_index_probe.hip")
    hdr = os.path.join(REPO, "include", "mppi_amd", "plugin", "parallel_utils.hpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return EXE
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17",
           "-I" + os.path.join(REPO, "include"), src, "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def test_parallel_probe_builds():
    """every direction of both enums instantiates (the reference declares fourteen + seven; a plugin may name any of them)"""
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_parallel_directions_on_the_device(gpu):
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "PARALLEL OK" in r.stdout
