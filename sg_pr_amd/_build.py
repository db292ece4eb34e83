"""Build libsgpr_hip.so (the C-ABI in include/sgpr.h) for gfx950 with hipcc, in-tree.

Every HIP source is compiled to its own object (in parallel) and the objects are linked into
sg_pr_amd/lib/libsgpr_hip.so.  Staleness is decided by CONTENT, never by timestamps or by where the code runs: the
sha256 of every source, header and the compiler flags is stored next to each object and next to the library
(`*.srchash`); an artefact is reused only while its recorded hash equals the hash of what is on disk now.
`force=True` (or SGPR_FORCE_BUILD=1 in the environment) recompiles everything from scratch.
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libsgpr_hip.so")
SOURCES = ["sgpr_embed.hip", "sgpr_score.hip", "sgpr_metrics.hip", "sgpr_modules.hip", "sgpr_cluster.hip",
           "sgpr_generic.hip", "sgpr_wide.hip", "sgpr_prep.hip", "sgpr_api.hip"]
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


def _hash_of(paths):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_hash():
    """sha256 over every HIP source, the shared headers and the compiler flags (what the library is a function of)."""
    return _hash_of([os.path.join(CSRC, s) for s in _sources()] + HEADERS)


def header_abi_version():
    """SGPR_ABI_VERSION of include/sgpr.h - what a loaded library must answer from sgpr_abi_version()."""
    with open(HEADERS[0]) as f:
        return int(re.search(r"#define\s+SGPR_ABI_VERSION\s+(\d+)", f.read()).group(1))


def _recorded(path):
    try:
        with open(path + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale():
    return not os.path.exists(LIB_PATH) or _recorded(LIB_PATH) != source_hash()


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
        want = _hash_of([path] + HEADERS)
        if not force and os.path.exists(obj) and _recorded(obj) == want:
            return obj
        cmd = [hipcc] + FLAGS + inc + ["-c", path, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(obj + ".tmp", obj)
        with open(obj + ".srchash", "w") as f:
            f.write(want)
        return obj

    total = source_hash()
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(_sources())) as pool:
        objs = list(pool.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(LIB_PATH + ".srchash", "w") as f:
        f.write(total)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
