"""Build libsgpr_hip.so (the C-ABI in include/sgpr.h) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsgpr_hip.so")
SOURCES = ["sgpr_embed.hip", "sgpr_score.hip", "sgpr_metrics.hip", "sgpr_api.hip"]
HEADERS = [os.path.join(REPO, "include", "sgpr.h"), os.path.join(CSRC, "sgpr_internal.hpp")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libsgpr_hip.so")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into sg_pr_amd/lib/libsgpr_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
