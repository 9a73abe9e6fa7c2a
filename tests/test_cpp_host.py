"""The C++ host classes (include/mppi_amd/controllers.hpp) over the C ABI: the reference's cartpole example rebuilt
with g++ only (no hipcc, no Eigen) against libmppi_amd.so."""
import os
import subprocess

import pytest

import mppi_generic_amd as m

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "examples", "_build", "cartpole_example")


def _build(name="cartpole_example"):
    m.load_library()
    exe = os.path.join(os.path.dirname(EXE), name)
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.dirname(m.library_path())
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", name + ".cpp"), "-L" + lib_dir, "-lmppi_amd",
           "-Wl,-rpath," + lib_dir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_host_example_builds_and_fails_loudly_without_device(lib):
    exe = _build()
    _build("cartpole_plant_example")
    if lib.mppi_device_count() > 0:
        return
    r = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_host_cartpole_example_reaches_goal(gpu):
    """examples/cartpole_example.cu of the reference runs 5000 steps; after 600 (12 s of simulated time) the cart has
    reached the goal position of the example's cost"""
    exe = _build()
    r = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "AT GOAL" in r.stdout


@pytest.mark.gpu
def test_cpp_plant_example(gpu):
    """include/mppi_amd/plant.hpp: SimulatedPlant single-threaded (strides 1 and 2) and runControlLoop on a thread"""
    exe = _build("cartpole_plant_example")
    r = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "PLANT OK" in r.stdout and "stride 2: 301 iterations, last stride 2" in r.stdout
