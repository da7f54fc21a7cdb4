"""The drop-in boundary: libmofreak_hip.so loads without a GPU and exports exactly what
include/mofreak_hip.h declares (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import mofreak_amd as M
from mofreak_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mofreak_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mofreak_[a-z0-9_]+)\s*\(", src)))


def test_header_is_plain_c(tmp_path):
    # the header must compile as C89-ish C with nothing but <stddef.h>/<stdint.h>: no torch / HIP types in the ABI
    c = tmp_path / "t.c"
    c.write_text('#include "mofreak_hip.h"\nint main(void){ mofreak_params p; (void)p; return sizeof(mofreak_row) == 32 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    subprocess.check_call([str(exe)])
    assert "torch" not in open(HEADER).read().lower().replace("torch's current stream", "")


def test_library_exports_every_declared_symbol(native_lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(native_lib, n), f"{n} declared in mofreak_hip.h but not exported"
    assert sorted(api.EXPORTS) == names, "api.EXPORTS out of date with the header"
    assert native_lib.mofreak_abi_version() == 3


def test_python_constants_are_the_headers():
    """Every enumerated value api.py spells out again (error codes, memory flags, path modes, bit modes) is the header's."""
    src = open(HEADER).read()
    defines = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+MOFREAK_([A-Z0-9_]+)\s+\(?(-?[0-9a-fx]+u?)\)?", src)}
    for name in ("ERR_BAD_ARG", "ERR_HIP", "ERR_OOM", "ERR_UNSUPPORTED", "ERR_NO_DEVICE", "ERR_ROI", "ERR_CAPACITY", "MEM_DEVICE", "MEM_HOST",
                 "ROWS_DEVICE", "BITS_SSE", "BITS_NATURAL", "BITS_SSE_SIGNED", "FP_X87", "FP_SSE", "PATH_AUTO", "PATH_GATHER"):
        assert name in defines, name
        assert getattr(api, name) == defines[name], name


def test_default_params_are_the_reference_constants(native_lib):
    p = api.default_params()
    assert p.struct_size == C.sizeof(api.Params)
    assert (p.gap_for_frame_difference, p.mip_theta, p.freak_n_octaves) == (5, 288, 4)  # MoFREAKUtilities.cpp:378, :48
    assert p.freak_pattern_scale == 22.0 and p.freak_orientation_normalized == 1 and p.freak_scale_normalized == 1
    assert p.brisk_fp_model == api.FP_X87 == 0  # the reference as it was built: Visual Studio 2010, Win32, x87
    assert p.freak_bit_mode == M.BITS_SSE


def test_create_rejects_bad_arguments(native_lib):
    h = C.c_void_p()
    p = api.default_params()
    p.struct_size = 4
    assert native_lib.mofreak_create(M.TABLES_ONLY, C.byref(p), C.byref(h)) == api.ERR_BAD_ARG
    assert b"struct_size" in native_lib.mofreak_last_error(None)
    assert native_lib.mofreak_create(M.TABLES_ONLY, None, None) == api.ERR_BAD_ARG
    native_lib.mofreak_destroy(None)  # no-op


def test_struct_layouts():
    assert M.ROW_DTYPE.itemsize == 32 and M.KEYPOINT_DTYPE.itemsize == 12
    assert M.ROW_DTYPE.fields["appearance"][1] == 16 and M.ROW_DTYPE.fields["motion"][1] == 24
    kp = np.zeros(2, M.KEYPOINT_DTYPE)
    assert kp.view(np.float32).reshape(2, 3).shape == (2, 3)


def test_no_oracle_in_the_product():
    """The product must not import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mofreak_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "mofreak_oracle" not in txt and "libmofreak_oracle" not in txt, f
    out = subprocess.run(["ldd", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_no_fill_on_the_null_stream():
    """The library's streams are created non-blocking: they do not wait for the null stream, which is where a plain hipMemset
    runs (and it returns before the fill is done).  Round 4 met that twice -- a fill that overwrote live data, words cleared at
    context creation -- so every fill in csrc/ names its stream."""
    import re
    csrc = os.path.join(ROOT, "mofreak_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".h")):
            code = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f), errors="ignore").read())  # (comments may talk about it)
            assert not re.search(r"\bhipMemset\s*\(", code), f


def test_host_code_under_address_sanitizer(tmp_path):
    """SURVEY.md section 5 (sanitizers): the C ABI's host code (tables, .mofreak text, argument checks) and the C++
    facade's reader / writer built with -fsanitize=address,undefined, kernel launchers stubbed out (no GPU involved;
    GPU sanitizers are not available on the pool)."""
    import subprocess
    host = os.path.join(ROOT, "mofreak_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s", "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([os.path.join(host, "asan_selftest")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan selftest ok" in out.stdout, out.stdout + out.stderr
