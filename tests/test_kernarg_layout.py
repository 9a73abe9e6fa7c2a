"""include/mppi_amd/engine/kernarg_view.hpp: the role loops of the pipelined rollout kernels read their plugin objects from the
kernarg segment (s_load inside the step loop) instead of holding them in — spilled — SGPRs across it.  That needs the byte offset of
every by-value kernel argument; KernargLayout computes it from sizeof / alignof (each argument at the next multiple of its
alignment, explicit arguments first).  Here the formula is held to what the COMPILER says: the `.offset` fields of the code
object's kernel metadata, for a probe kernel with 4-, 8- and 16-byte aligned structs (CPU: hipcc cross-compiles), and the view
is exercised on the GPU against the named arguments."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "examples", "_build")
SRC = os.path.join(REPO, "tests", "probes", "kernarg_layout_probe.hip")
LLVM = "/opt/rocm/lib/llvm/bin/"


def _build():
    os.makedirs(OUT, exist_ok=True)
    exe, co = os.path.join(OUT, "kernarg_layout_probe"), os.path.join(OUT, "kernarg_layout_probe.co")
    hdr = os.path.join(REPO, "include", "mppi_amd", "engine", "kernarg_view.hpp")
    if not (os.path.exists(exe) and os.path.exists(co) and
            os.path.getmtime(exe) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr))):
        common = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), SRC]
        for cmd in (common + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", co], common + ["-o", exe]):
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
    return exe, co


def test_kernarg_layout_formula_matches_the_compilers_metadata():
    exe, co = _build()
    said = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    blk = notes[notes.index(".args:"):notes.index(".group_segment_fixed_size")]
    explicit = [(int(o), int(sz)) for o, sz, kind in
                re.findall(r"\.offset:\s*(\d+)\s*\n\s*\.size:\s*(\d+)\s*\n\s*\.value_kind:\s*(\w+)", blk) if not kind.startswith("hidden")]
    assert len(explicit) == 6, blk
    assert [o for o, _ in explicit] == said, (explicit, said)
    assert [sz for _, sz in explicit] == [20, 24, 32, 28, 4, 8]   # A4, B8 (padded to 8), C16 (padded to 16), D4, int, pointer
    assert said[2] % 16 == 0 and said[1] % 8 == 0                 # the alignment rule did something


@pytest.mark.gpu
def test_kernarg_view_reads_what_the_named_arguments_hold(gpu):
    exe, _ = _build()
    r = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
