"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol include/mppi_amd.h declares."""
import os
import re

import mppi_generic_amd as m

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "mppi_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), "libmppi_amd.so does not export " + name
        assert name in m.SIGNATURES, "mppi-generic_amd/capi.py does not bind " + name
    assert set(m.SIGNATURES) <= set(declared), set(m.SIGNATURES) - set(declared)


def test_library_is_in_tree_and_gfx950(lib):
    path = m.library_path()
    assert path.startswith(REPO) and os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob  # the embedded code object's target
    assert b"rolloutKernel" in blob


def test_no_cpu_fallback_without_device(lib):
    """Without a HIP device the product refuses to run (no compute calls are attempted here)."""
    if lib.mppi_device_count() > 0:
        return
    try:
        m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0)
    except m.MPPIError as e:
        assert e.status == 3
    else:
        raise AssertionError("mppi_create succeeded without a device")


def test_product_does_not_reference_the_oracle():
    """the product path must never route through oracle/ (it is test infrastructure)"""
    pkg = os.path.join(REPO, "mppi-generic_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "pyoracle" not in txt and "oracle/" not in txt and "liboracle" not in txt, os.path.join(d, f)
    for d, _, files in os.walk(os.path.join(REPO, "include")):
        for f in files:
            assert "#include \"oracle" not in open(os.path.join(d, f)).read()


def test_list_models_and_status_strings(lib):
    models = lib.mppi_list_models().decode().split("\n")
    assert {"cartpole", "double_integrator", "autorally_nn", "bicycle_slip_lstm", "racer_dubins"} <= set(models)
    assert lib.mppi_status_string(0) == b"ok" and lib.mppi_status_string(3) != b"ok"


def test_hand_written_dpp_instructions_have_no_hazard(lib):
    """LSTMQuadRows issues v_fmac_f32_dpp from inline assembly, which the compiler's hazard recogniser does not see: the
    built code objects must not write a DPP source register (or EXEC) with the VALU inside the hardware's wait-state window"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dpp_hazard_lint.py"), m.library_path()],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 hazard(s)" in r.stdout and not r.stdout.startswith("0 DPP"), r.stdout


def test_committed_isa_step_counts_match_the_sources(lib):
    """bench.py's issue floor multiplies a STATIC instruction count (profiles/r*_isa_step_counts.json) with issue intervals it
    measures live: the committed counts must be what the current sources compile to (regenerate with
    `python tools/isa_step_count.py profiles/r04_isa_step_counts.json` after touching a step loop)"""
    import glob
    import json
    import subprocess
    import sys
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_isa_step_counts.json")))
    assert files, "no committed ISA step counts"
    want = json.load(open(files[-1]))
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_step_count.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    for key, e in want.items():
        assert "error" not in got.get(key, {"error": 1}), (key, got.get(key))
        for f in ("loop_instructions", "steps_per_trip", "instructions_per_step", "vector_instructions_per_step"):
            assert got[key][f] == e[f], "%s.%s: committed %s, sources compile to %s" % (key, f, e[f], got[key][f])
    assert "double_integrator_robust" in lib.mppi_list_models().decode().split("\n")


def test_streamed_merge_first_trip_keeps_its_shape_in_the_isa(lib):
    """The streamed merge (rolloutPipelineKernel, STREAM_MERGE) is only worth its launch if the record loads of a sampler wave's
    first trip FLY UNDER its first draw — which the compiler undoes when left alone (round 4: selects next to the loads, the
    dead half of wider loads reused as another load's destination, the draw sunk below the merge: three exposed round trips,
    4.6 instead of 3.4 us to the first sample).  Checked on the ISA of the built unit, in program order of the sampler wave's
    entry path: tail loads and column-quad loads issued -> Philox rounds (v_mul_hi_u32) -> the first wait for a load.  And the
    kernel keeps everything in registers (no scratch)."""
    import re
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_tu
    path, _ = isa_tu.disassemble(os.path.join(REPO, "mppi-generic_amd", "csrc", "models", "cartpole.hip"))
    lines = open(path).read().split("\n")
    heads = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]{16} <", l)] + [len(lines)]
    body = None
    for a, b in zip(heads, heads[1:]):
        if "rolloutPipelineKernel" in lines[a] and "ELi1ELb1ELb0ELb0ELb1EE" in lines[a]:
            body = [l.split("//")[0].strip() for l in lines[a + 1:b]]
    assert body, "the STREAM_MERGE instantiation of rolloutPipelineKernel<Cartpole> is not in cartpole.hip's code object"
    first = lambda pat: next(i for i, l in enumerate(body) if re.search(pat, l))  # noqa: E731
    tails, quads = first(r"^global_load_dwordx2 "), first(r"^global_load_dwordx4 ")
    draw, wait = first(r"^v_mul_hi_u32 "), first(r"^s_waitcnt vmcnt")
    assert tails < draw and quads < draw, (tails, quads, draw)   # the loads are issued in front of the draw ...
    assert draw < wait, (draw, wait)                             # ... and nothing waits for them before the draw has started
    n_mul = sum(1 for l in body[draw:wait] if l.startswith(("v_mul_hi_u32", "s_mul_hi_u32")))
    assert n_mul >= 16, n_mul                                    # (the ten Philox rounds' high multiplies: the whole draw is in between)
    co = os.path.splitext(path)[0] + ".co"
    notes = subprocess.run([isa_tu.LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    for blk in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "rolloutPipelineKernel" in name and "ELi1ELb1ELb0ELb0ELb1EE" in name:
            assert int(re.search(r"private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0
            assert int(re.search(r"vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0
            break
    else:
        raise AssertionError("kernel metadata not found")
