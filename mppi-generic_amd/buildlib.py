"""Builds libmppi_amd.so (the HIP engine + C ABI + the in-tree model instantiations) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Flags that matter:
  --offload-arch=gfx950   the only target (MI355X / CDNA4)
  -ffp-contract=off       the only fused multiply-adds are the explicit det::fma() calls, so device results match the
                          CPU oracle bit for bit (include/mppi_amd/det_math.h)

One object per translation unit (csrc/engine_*.hip + csrc/models/*.hip), compiled in parallel and cached under
csrc/build/ with the compiler's own dependency files (-MD), then linked.  The library carries the digest of the
sources it was built from (mppi_source_hash()); build() rebuilds whenever the tree's digest differs from the
library's, so "the shipped .so matches the sources" is checked, not assumed.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD_DIR = os.path.join(CSRC, "build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB = os.path.join(LIB_DIR, "libmppi_amd.so")
INCLUDE = os.path.join(REPO, "include")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]

# Flags of single translation units (file name -> extra flags), decided by A/B measurement on the MI355X, never by guess.
#   cartpole.hip: the GCN scheduler's max-ILP strategy (same instructions, another order — results bit-identical, the parity
#   suite runs on it): Cartpole K=16384, T=100 iteration 24.27 -> 23.79 us in one session (round 4; `buildlib.py --variant
#   maxilp all -mllvm -amdgpu-sched-strategy=max-ilp` is the experiment).  The same flag on every unit: AutoRally-NN +-0 although
#   its dynamics wave loses 14 % of its instructions (83 of 83 s_nop per two steps are gone — they were not what bounds it),
#   LSTM + colored -1.3 %, DI Tube -1.5 %, Robust MPPI 3-12 % slower, the elevation models -1.3 .. +3.2 %: not applied there.
#   racer_dubins_elevation_lstm_steering.hip: same flag, same session: 2725.6 -> 2812.2 iterations/s (+3.2 %, K=16384, T=100,
#   colored noise), its parity tests bit-identical on the flagged build.
UNIT_FLAGS = {
    "cartpole.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "racer_dubins_elevation_lstm_steering.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
}


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _flags(tu=None):
    extra = os.environ.get("MPPI_HIPCC_EXTRA", "")
    unit = UNIT_FLAGS.get(os.path.basename(tu), []) if tu else []
    return FLAGS + unit + (extra.split() if extra else []) + ["-I" + INCLUDE, "-I" + CSRC]


def unit_flags(tu):
    """the per-unit flags of a translation unit (tools/isa_tu.py compiles with them too)"""
    return list(UNIT_FLAGS.get(os.path.basename(tu), []))


def translation_units():
    tus = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("engine_") and f.endswith(".hip"))
    mdir = os.path.join(CSRC, "models")
    tus += sorted(os.path.join(mdir, f) for f in os.listdir(mdir) if f.endswith(".hip"))
    return tus


def _sources():
    out = []
    for root in (CSRC, INCLUDE):
        for d, dirs, files in os.walk(root):
            dirs[:] = [x for x in dirs if x != "build"]
            for f in files:
                if f.endswith((".hip", ".hpp", ".h")):
                    out.append(os.path.join(d, f))
    return sorted(out)


def source_hash():
    """sha256 over (relative path, contents) of every source the library is built from + the compiler flags"""
    h = hashlib.sha256()
    h.update(" ".join(_flags()[:len(FLAGS)] + os.environ.get("MPPI_HIPCC_EXTRA", "").split()).encode())
    h.update(repr(sorted(UNIT_FLAGS.items())).encode())
    for s in _sources():
        h.update(os.path.relpath(s, REPO).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def library_hash():
    """the digest embedded in the built library (read from the file, without loading it)"""
    if not os.path.exists(LIB):
        return None
    with open(LIB, "rb") as f:
        blob = f.read()
    tag = b"MPPI_AMD_SOURCE_HASH="
    i = blob.find(tag)
    return blob[i + len(tag):i + len(tag) + 32].decode() if i >= 0 else None


def needs_build():
    return library_hash() != source_hash()


def _obj_path(tu):
    return os.path.join(BUILD_DIR, os.path.relpath(tu, CSRC).replace(os.sep, "_") + ".o")


def _obj_stale(tu, obj, flag_sig):
    dep = obj + ".d"
    sig = obj + ".flags"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(sig)):
        return True
    if open(sig).read() != flag_sig:
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    deps = text.split(":", 1)[1].split() if ":" in text else []
    for d in deps + [tu]:
        if not os.path.exists(d) or os.path.getmtime(d) > t:
            return True
    return False


def _compile(tu, verbose):
    obj = _obj_path(tu)
    flag_sig = " ".join(_flags(tu))
    if not _obj_stale(tu, obj, flag_sig):
        return obj, 0.0
    import time
    t0 = time.time()
    cmd = [_hipcc()] + _flags(tu) + ["-MD", "-MF", obj + ".d", "-c", tu, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (tu, r.stdout, r.stderr))
    with open(obj + ".flags", "w") as f:
        f.write(flag_sig)
    return obj, time.time() - t0


def build_variant(tag, extra_flags, only=None, verbose=False):
    """An experimental build for A/B measurements (tools/): the translation units matching `only` (substring; all when
    None) compiled with extra flags into csrc/build/variant_<tag>/, linked with the regular objects of the other units
    into lib/libmppi_amd_<tag>.so.  Load it with MPPI_AMD_LIB=<path>."""
    build()
    vdir = os.path.join(BUILD_DIR, "variant_" + tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    todo = []
    for tu in translation_units():
        if only is None or only in os.path.basename(tu):
            obj = os.path.join(vdir, os.path.basename(tu) + ".o")
            todo.append((tu, obj))
            objs.append(obj)
        else:
            objs.append(_obj_path(tu))

    def one(job):
        tu, obj = job
        cmd = [_hipcc()] + _flags(tu) + list(extra_flags) + ["-c", tu, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (tu, r.stdout, r.stderr))

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
        list(ex.map(one, todo))
    out = os.path.join(LIB_DIR, "libmppi_amd_%s.so" % tag)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", os.path.join(BUILD_DIR, "source_hash.o")] + objs + \
          ["-o", out, "-ldl", "-lz"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return out


def build(force=False, verbose=False, jobs=None):
    """Compile csrc/engine_*.hip + csrc/models/*.hip -> lib/libmppi_amd.so.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(BUILD_DIR, exist_ok=True)
    if force:
        import shutil
        for f in os.listdir(BUILD_DIR):
            path = os.path.join(BUILD_DIR, f)
            if os.path.isdir(path):  # build_variant()'s variant_<tag>/ directories
                shutil.rmtree(path)
            else:
                os.remove(path)
    digest = source_hash()
    hash_cpp = os.path.join(BUILD_DIR, "source_hash.cpp")
    with open(hash_cpp, "w") as f:
        f.write('extern "C" const char* mppi_source_hash_impl(void)\n{\n'
                '  static const char tag[] = "MPPI_AMD_SOURCE_HASH=%s";\n  return tag + 21;\n}\n' % digest)
    tus = translation_units()
    jobs = jobs or int(os.environ.get("MPPI_BUILD_JOBS", "0")) or min(len(tus), os.cpu_count() or 4)
    # the slowest units first (the NN models with their MFMA / RMPPI / pipeline instantiations)
    order = sorted(tus, key=lambda t: (0 if ("lstm" in t or "autorally" in t) else 1, t))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda t: _compile(t, verbose), order))
    if verbose:
        for (obj, dt), tu in zip(results, order):
            print("  %-40s %6.1f s" % (os.path.basename(tu), dt), file=sys.stderr)
    objs = [o for o, _ in results]
    hash_obj = os.path.join(BUILD_DIR, "source_hash.o")
    r = subprocess.run(["g++", "-O1", "-fPIC", "-c", hash_cpp, "-o", hash_obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed on source_hash.cpp:\n" + r.stdout + r.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", hash_obj] + objs + ["-o", LIB + ".tmp", "-ldl", "-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    assert library_hash() == digest, "embedded source hash not found in the linked library"
    return LIB


if __name__ == "__main__":
    import time
    t0 = time.time()
    if "--variant" in sys.argv:  # buildlib.py --variant <tag> <only-substring|all> <extra flags...>
        i = sys.argv.index("--variant")
        tag, only = sys.argv[i + 1], sys.argv[i + 2]
        print(build_variant(tag, sys.argv[i + 3:], None if only == "all" else only, verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print("build took %.1f s" % (time.time() - t0), file=sys.stderr)
