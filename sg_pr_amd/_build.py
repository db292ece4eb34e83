"""Build libsgpr_hip.so (the C-ABI in include/sgpr.h) for gfx950 with hipcc, in-tree.

Every HIP source is compiled to its own object (in parallel; an object is reused while it is newer than its source
and the shared headers) and the objects are linked into sg_pr_amd/lib/libsgpr_hip.so.  `force=True` (or
SGPR_FORCE_BUILD=1 in the environment) recompiles everything from scratch.
"""
import concurrent.futures
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libsgpr_hip.so")
SOURCES = ["sgpr_embed.hip", "sgpr_score.hip", "sgpr_metrics.hip", "sgpr_modules.hip", "sgpr_cluster.hip",
           "sgpr_api.hip"]
HEADERS = [os.path.join(REPO, "include", "sgpr.h"), os.path.join(CSRC, "sgpr_internal.hpp")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libsgpr_hip.so")
    return exe


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _newer_than(path, deps):
    if not os.path.exists(path):
        return False
    t = os.path.getmtime(path)
    return all(os.path.getmtime(d) <= t for d in deps)


def is_stale():
    deps = [os.path.join(CSRC, s) for s in _sources()] + HEADERS
    return not _newer_than(LIB_PATH, deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into sg_pr_amd/lib/libsgpr_hip.so."""
    force = force or os.environ.get("SGPR_FORCE_BUILD", "") not in ("", "0")
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    inc = ["-I" + os.path.join(REPO, "include"), "-I" + CSRC]

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), _obj(src)
        if not force and _newer_than(obj, [path] + HEADERS):
            return obj
        cmd = [hipcc] + FLAGS + inc + ["-c", path, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(obj + ".tmp", obj)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(_sources())) as pool:
        objs = list(pool.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
