"""In-tree build of libmofreak_hip.so (gfx950 kernels + C ABI) with hipcc.  No torch involved."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmofreak_hip.so")
DEBUG_LIB_PATH = os.path.join(PKG_DIR, "libmofreak_hip_debug.so")  # -DMOFREAK_DEBUG_BOUNDS: checked LDS accesses / stores
SOURCES = ["kernels.hip", "tile_kernel.hip", "bow_kernel.hip", "detect_kernel.hip", "capi.cpp", "tables.cpp", "format.cpp"]
HEADERS = ["tables.h", "device_types.h", "device_helpers.h", "mip_lane_order.inc", os.path.join("..", "..", "include", "mofreak_hip.h")]
# -ffp-contract=off / -fno-fast-math: a handful of float/double expressions restate reference
# expressions whose rounding is part of the result.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmofreak_hip.so cannot be built (there is no CPU fallback)")
    return exe


def is_stale(lib: str = LIB_PATH) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_native(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    """Compile the shared library if it is missing or older than its sources; returns its path.
    debug=True: the bounds-checking build of the same sources (tests only; never what api.load() picks up)."""
    if debug:
        if not force and not is_stale(DEBUG_LIB_PATH):
            return DEBUG_LIB_PATH
        cmd = [hipcc(), *FLAGS, "-DMOFREAK_DEBUG_BOUNDS", "-o", DEBUG_LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return DEBUG_LIB_PATH
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc(), *FLAGS, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


DIST_LIB_PATH = os.path.join(PKG_DIR, "libmofreak_dist.so")  # include/mofreak_dist.h over RCCL (host code only: no kernels of its own)
DIST_SOURCES = ["dist_rccl.cpp"]
DIST_HEADERS = ["dist_gather.h", os.path.join("..", "..", "include", "mofreak_dist.h"), os.path.join("..", "..", "include", "mofreak_hip.h")]


def build_dist(force: bool = False, verbose: bool = False) -> str:
    """libmofreak_dist.so: the N-GPU exchange step over rccl.h (links /opt/rocm's librccl and HIP runtime: it is loaded by the
    C++ host side, never into a Python process that has torch's own copies of those libraries)."""
    if not force and os.path.exists(DIST_LIB_PATH):
        t = os.path.getmtime(DIST_LIB_PATH)
        if not any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in DIST_SOURCES + DIST_HEADERS):
            return DIST_LIB_PATH
    rocm = os.path.dirname(os.path.dirname(hipcc()))
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "c++", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include",
           *[os.path.join(CSRC, s) for s in DIST_SOURCES], "-o", DIST_LIB_PATH, f"-L{rocm}/lib", "-lrccl", "-lamdhip64", f"-Wl,-rpath,{rocm}/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DIST_LIB_PATH


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
    print(build_dist(force=True, verbose=True))
