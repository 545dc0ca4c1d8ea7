"""Build libac_hip.so (gfx950) from anomaly-clustering_amd/csrc/*.hip with hipcc.

In-tree, explicit: one `hipcc -c` per translation unit (in parallel), one link.
The .so lands next to this file so that it travels with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "..", "include"))
GENERATIONS = os.path.normpath(os.path.join(HERE, "..", "..", "..", "scratch", "gemm_generations"))   # not part of the package
LIB = os.path.join(HERE, "libac_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-I", INCLUDE]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for root in (CSRC, GENERATIONS, INCLUDE):
        if not os.path.isdir(root):
            continue
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build(force=False, verbose=True, experimental=False):
    """Compile if the library is missing or older than any source/header.  Returns the path.
    experimental=True builds libac_hip_experimental.so with -DAC_EXPERIMENTAL instead: the superseded GEMM
    generations under scratch/gemm_generations/ become selectable through ac_set_gemm_policy(101..109) for same-box
    A/B runs (load it with AC_LIB=...).  The product library never contains them."""
    if experimental:
        return _build(force, verbose, LIB[:-3] + "_experimental.so", OBJ_DIR + "_experimental", ["-DAC_EXPERIMENTAL", "-I", CSRC, "-I", GENERATIONS])
    return _build(force, verbose, LIB, OBJ_DIR, [])


def _build(force, verbose, LIB, OBJ_DIR, extra):
    srcs = _sources()
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found at %s: cannot build libac_hip.so" % hipcc)
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


def build_variant(name, defines, verbose=True):
    """A diagnostic build libac_hip_<name>.so with extra -D flags (same-box A/B runs: scratch/ab_pairmin.py, AC_LIB=...)."""
    return _build(True, verbose, LIB[:-3] + "_" + name + ".so", OBJ_DIR + "_" + name, list(defines))


if __name__ == "__main__":
    if "--variant" in sys.argv:                          # build.py --variant NAME -DFOO=1 -DBAR=2
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], [f for f in sys.argv[i + 2:] if f.startswith("-D")])
    else:
        build(force="--force" in sys.argv, experimental="--experimental" in sys.argv)
