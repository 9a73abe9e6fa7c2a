"""Builds libmppi_amd.so (the HIP engine + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Flags that matter:
  --offload-arch=gfx950   the only target (MI355X / CDNA4)
  -ffp-contract=off       the only fused multiply-adds are the explicit det::fma() calls, so device results match the
                          CPU oracle bit for bit (include/mppi_amd/det_math.h)
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB = os.path.join(LIB_DIR, "libmppi_amd.so")


def _sources():
    out = []
    for root in (CSRC, os.path.join(REPO, "include")):
        for d, _, files in os.walk(root):
            for f in files:
                if f.endswith((".hip", ".hpp", ".h")):
                    out.append(os.path.join(d, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, verbose=False):
    """Compile csrc/engine.hip -> lib/libmppi_amd.so.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
        "-I" + os.path.join(REPO, "include"), "-I" + CSRC,
        os.path.join(CSRC, "engine.hip"), "-o", LIB + ".tmp", "-ldl", "-lz",
    ]
    extra = os.environ.get("MPPI_HIPCC_EXTRA", "")
    if extra:
        cmd[1:1] = extra.split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
