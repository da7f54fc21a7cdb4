"""In-tree build of libmofreak_hip.so (gfx950 kernels + C ABI) with hipcc.  No torch involved."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmofreak_hip.so")
DEBUG_LIB_PATH = os.path.join(PKG_DIR, "libmofreak_hip_debug.so")  # -DMOFREAK_DEBUG_BOUNDS: checked LDS accesses / stores
SOURCES = ["tile_kernel.hip", "kernels.hip", "bow_kernel.hip", "detect_kernel.hip", "format_kernel.hip", "capi.cpp", "tables.cpp", "format.cpp"]
HEADERS = ["tables.h", "device_types.h", "device_helpers.h", "mip_lane_order.inc", "mip_lane.h", "resize_axis.h", os.path.join("..", "..", "include", "mofreak_hip.h")]
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


def _compile_and_link(out: str, extra: list[str], verbose: bool) -> None:
    """One hipcc process per source (the tile kernel's per-ROI-side MIP instantiations take minutes; the other files
    compile beside it), objects under build/obj/<library name>/, then one link."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(os.path.dirname(PKG_DIR), "build", "obj", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]

    def one(src: str) -> str:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), *compile_flags, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_native(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    """Compile the shared library if it is missing or older than its sources; returns its path.
    debug=True: the bounds-checking build of the same sources (tests only; never what api.load() picks up)."""
    if debug:
        if not force and not is_stale(DEBUG_LIB_PATH):
            return DEBUG_LIB_PATH
        _compile_and_link(DEBUG_LIB_PATH, ["-DMOFREAK_DEBUG_BOUNDS"], verbose)
        return DEBUG_LIB_PATH
    if not force and not is_stale():
        return LIB_PATH
    _compile_and_link(LIB_PATH, [], verbose)
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
