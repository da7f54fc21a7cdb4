"""ctypes binding of the CPU parity oracle (oracle/libmofreak_oracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libmofreak_oracle.so")

BITS_SSE, BITS_NATURAL, BITS_SSE_SIGNED = 0, 1, 2


class Row(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("frame_number", C.c_int32), ("scale", C.c_float),
                ("appearance", C.c_uint8 * 8), ("motion", C.c_uint8 * 8)]


ROW_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("frame_number", "<i4"), ("scale", "<f4"),
                      ("appearance", "u1", (8,)), ("motion", "u1", (8,))])
assert ROW_DTYPE.itemsize == 32 and C.sizeof(Row) == 32

_lib = None


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("mofreak_oracle.c", "mofreak_oracle.h", "brisk_oracle.c", "brisk_oracle.h")]
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    # MOFREAK_ORACLE_LIBRARY: e.g. the `make -C oracle asan` build (run with LD_PRELOAD=libasan.so)
    L = C.CDLL(os.environ.get("MOFREAK_ORACLE_LIBRARY", LIB_PATH))
    u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.mo_freak_create.restype = C.c_void_p
    L.mo_freak_create.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mo_freak_destroy.argtypes = [C.c_void_p]
    L.mo_freak_get_pairs.argtypes = [C.c_void_p, u8p]
    L.mo_freak_get_orientation.argtypes = [C.c_void_p, i32p]
    L.mo_freak_get_pattern.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p]
    L.mo_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.mo_absdiff.argtypes = [u8p, u8p, u8p, C.c_int, C.c_int]
    L.mo_integral.argtypes = [u8p, C.c_int, C.c_int, i32p]
    L.mo_freak_scale_index.argtypes = [C.c_void_p, C.c_float]
    L.mo_freak_theta_index.argtypes = [C.c_int, C.c_int]
    L.mo_freak_theta_index_atan2f.argtypes = [C.c_int, C.c_int]
    L.mo_freak_compute.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, f32p, C.c_int, u8p, u8p, i32p, i32p]
    L.mo_resize_linear_8u.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    L.mo_resize_axis_table.argtypes = [C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_short)]
    L.mo_mip.restype = C.c_uint
    L.mo_mip.argtypes = [u8p, u8p, C.c_int, C.c_int]
    L.mo_mip_descriptor.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, u8p]
    L.mo_extract_pair.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, u8p, u8p]
    L.mo_extract_stream.restype = C.c_long
    L.mo_extract_stream.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p,
                                    C.POINTER(C.c_long), C.c_void_p, C.c_long]
    L.mo_bow_match.argtypes = [u8p, u8p, C.c_int, C.c_int]
    L.mo_bow_histogram.argtypes = [u8p, C.c_long, u8p, C.c_int, C.c_int, f32p]
    L.mo_format_row.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    # detector (brisk_oracle.h)
    L.mo_brisk_create.restype = C.c_void_p
    L.mo_brisk_create.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mo_brisk_destroy.argtypes = [C.c_void_p]
    L.mo_brisk_layers.argtypes = [C.c_void_p]
    L.mo_brisk_layer_info.argtypes = [C.c_void_p, C.c_int, i32p, i32p, f32p, f32p]
    L.mo_brisk_layer_image.restype = C.c_void_p
    L.mo_brisk_layer_image.argtypes = [C.c_void_p, C.c_int]
    L.mo_brisk_layer_scores.restype = C.c_void_p
    L.mo_brisk_layer_scores.argtypes = [C.c_void_p, C.c_int]
    L.mo_brisk_get_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.mo_brisk_layer_points.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int]
    L.mo_brisk_detect.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.mo_brisk_halfsample.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.mo_brisk_twothirdsample.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.mo_oast9_16_is_corner.argtypes = [u8p, C.c_int, C.c_int]
    L.mo_oast9_16_score.argtypes = [u8p, C.c_int, C.c_int]
    L.mo_agast5_8_score.argtypes = [u8p, C.c_int, C.c_int]
    L.mo_oast9_16_detect.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int]
    L.mo_brisk_subpixel2d.restype = C.c_float
    L.mo_brisk_subpixel2d.argtypes = [i32p, f32p, f32p]
    L.mo_brisk_refine1d.restype = C.c_float
    L.mo_brisk_refine1d.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, f32p]
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Freak:
    """cv::FREAK with the reference's default constructor arguments (MoFREAKUtilities.cpp:427)."""

    def __init__(self, pattern_scale=22.0, n_octaves=4, orientation_normalized=True, scale_normalized=True,
                 bit_mode=BITS_SSE):
        self.h = lib().mo_freak_create(pattern_scale, n_octaves, int(orientation_normalized),
                                       int(scale_normalized), bit_mode)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mo_freak_destroy(self.h)
            self.h = None

    def scale_index(self, size: float) -> int:
        return lib().mo_freak_scale_index(self.h, float(np.float32(size)))

    def pattern_sizes(self):
        class _F(C.Structure):
            _fields_ = [("pattern_scale", C.c_float), ("n_octaves", C.c_int), ("on", C.c_int), ("sn", C.c_int),
                        ("bit_mode", C.c_int), ("lut", C.c_void_p), ("pattern_sizes", C.c_int * 64)]
        return list(_F.from_address(self.h).pattern_sizes)

    def description_pairs(self) -> np.ndarray:
        out = np.zeros((512, 2), np.uint8)
        lib().mo_freak_get_pairs(self.h, _u8(out))
        return out

    def orientation_pairs(self) -> np.ndarray:
        out = np.zeros((45, 4), np.int32)
        lib().mo_freak_get_orientation(self.h, _i32(out))
        return out

    def pattern(self, scale: int, rot: int) -> np.ndarray:
        out = np.zeros((43, 3), np.float32)
        lib().mo_freak_get_pattern(self.h, scale, rot, _f32(out))
        return out

    def compute(self, img: np.ndarray, kps: np.ndarray):
        """-> valid (n,), desc64 (n,64), theta (n,), dirs (n,2)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 3)
        n = kps.shape[0]
        H, W = img.shape
        valid = np.zeros(n, np.uint8)
        desc = np.zeros((n, 64), np.uint8)
        theta = np.zeros(n, np.int32)
        dirs = np.zeros((n, 2), np.int32)
        lib().mo_freak_compute(self.h, _u8(img), W, H, _f32(kps), n, _u8(valid), _u8(desc), _i32(theta), _i32(dirs))
        return valid, desc, theta, dirs

    def extract_pair(self, cur: np.ndarray, prev: np.ndarray, kps: np.ndarray):
        """-> desc16 (n,16), valid (n,)."""
        cur = np.ascontiguousarray(cur, dtype=np.uint8)
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 3)
        n = kps.shape[0]
        H, W = cur.shape
        desc = np.zeros((n, 16), np.uint8)
        valid = np.zeros(n, np.uint8)
        lib().mo_extract_pair(self.h, _u8(cur), _u8(prev), W, H, _f32(kps), n, _u8(desc), _u8(valid))
        return desc, valid

    def extract_stream(self, frames: np.ndarray, kps: np.ndarray, kp_offsets, gap: int = 5) -> np.ndarray:
        """frames (T,H,W); keypoints CSR per processed frame -> structured rows (ROW_DTYPE)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 3)
        T, H, W = frames.shape
        offs = np.ascontiguousarray(kp_offsets, dtype=np.int64)
        assert offs.shape[0] == max(T - gap, 0) + 1
        cap = int(offs[-1]) + 1
        rows = np.zeros(cap, dtype=ROW_DTYPE)
        n = lib().mo_extract_stream(self.h, _u8(frames), T, W, H, gap, _f32(kps),
                                    offs.ctypes.data_as(C.POINTER(C.c_long)), rows.ctypes.data, cap)
        return rows[:n].copy()


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W, _ = bgr.shape
    out = np.empty((H, W), np.uint8)
    lib().mo_bgr2gray(_u8(bgr), W, H, _u8(out))
    return out


def absdiff(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    d = np.empty_like(a)
    H, W = a.shape
    lib().mo_absdiff(_u8(a), _u8(b), _u8(d), W, H)
    return d


def integral(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.empty((H + 1, W + 1), np.int32)
    lib().mo_integral(_u8(img), W, H, _i32(out))
    return out


def resize_linear(src: np.ndarray, dw: int = 19, dh: int = 19) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().mo_resize_linear_8u(_u8(src), sw, sw, sh, _u8(dst), dw, dh)
    return dst


def resize_axis_table(ssize: int, dsize: int, is_x: bool):
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros(2 * dsize, np.int16)
    dmax = lib().mo_resize_axis_table(ssize, dsize, int(is_x), _i32(ofs), coef.ctypes.data_as(C.POINTER(C.c_short)))
    return ofs, coef.reshape(dsize, 2), dmax


def mip(cur19: np.ndarray, prev19: np.ndarray, x: int, y: int) -> int:
    cur19 = np.ascontiguousarray(cur19, np.uint8)
    prev19 = np.ascontiguousarray(prev19, np.uint8)
    assert cur19.size == 361 and prev19.size == 361
    return int(lib().mo_mip(_u8(cur19), _u8(prev19), x, y))


def mip_descriptor(cur, prev, size, x, y):
    cur = np.ascontiguousarray(cur, np.uint8)
    prev = np.ascontiguousarray(prev, np.uint8)
    H, W = cur.shape
    out = np.zeros(8, np.uint8)
    rc = lib().mo_mip_descriptor(_u8(cur), _u8(prev), W, H, float(np.float32(size)), int(x), int(y), _u8(out))
    return rc, out


def theta_index(d0: int, d1: int) -> int:
    return lib().mo_freak_theta_index(int(d0), int(d1))


def theta_index_atan2f(d0: int, d1: int) -> int:
    return lib().mo_freak_theta_index_atan2f(int(d0), int(d1))


def bow_assign(desc: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 16)
    codebook = np.ascontiguousarray(codebook, np.uint8).reshape(-1, 16)
    return np.array([lib().mo_bow_match(_u8(desc[k]), _u8(codebook), codebook.shape[0], 16) for k in range(desc.shape[0])], np.int32)


def bow_histogram(desc: np.ndarray, codebook: np.ndarray):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 16)
    codebook = np.ascontiguousarray(codebook, np.uint8).reshape(-1, 16)
    hist = np.zeros(codebook.shape[0], np.float32)
    ok = lib().mo_bow_histogram(_u8(desc), desc.shape[0], _u8(codebook), codebook.shape[0], 16, _f32(hist))
    return hist, bool(ok)


def format_rows(rows: np.ndarray) -> bytes:
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    buf = C.create_string_buffer(512)
    out = []
    for i in range(rows.shape[0]):
        n = lib().mo_format_row(rows[i:i + 1].ctypes.data, buf, 512)
        out.append(buf.raw[:n])
    return b"".join(out)


# ------------------------------------------------------------------ detector (oracle/brisk_oracle.c)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("response", "<f4"), ("layer", "<i4")])


class Brisk:
    """BriskScaleSpace over one image: pyramid, per-layer OAST points, the score cache, getKeypoints."""

    def __init__(self, img: np.ndarray, octaves: int = 3):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        self.h = lib().mo_brisk_create(_u8(img), W, W, H, octaves)
        self.n_layers = lib().mo_brisk_layers(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mo_brisk_destroy(self.h)
            self.h = None

    def layer_info(self, i):
        w, h = C.c_int32(), C.c_int32()
        sc, of = C.c_float(), C.c_float()
        lib().mo_brisk_layer_info(self.h, i, C.byref(w), C.byref(h), C.byref(sc), C.byref(of))
        return w.value, h.value, np.float32(sc.value), np.float32(of.value)

    def _plane(self, fn, i):
        w, h, _, _ = self.layer_info(i)
        ptr = fn(self.h, i)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w)).copy()

    def layer_image(self, i):
        return self._plane(lib().mo_brisk_layer_image, i)

    def layer_scores(self, i):
        return self._plane(lib().mo_brisk_layer_scores, i)

    def get_keypoints(self, threshold: int = 30, cap: int = 1 << 20) -> np.ndarray:
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = lib().mo_brisk_get_keypoints(self.h, threshold, out.ctypes.data, cap)
        assert n <= cap
        return out[:n].copy()

    def layer_points(self, i) -> np.ndarray:
        n = lib().mo_brisk_layer_points(self.h, i, None, 0)
        xy = np.zeros((max(n, 1), 2), np.int32)
        lib().mo_brisk_layer_points(self.h, i, _i32(xy), n)
        return xy[:n]


BRISK_FP_X87, BRISK_FP_SSE = 0, 1


def brisk_set_fp_model(model: int) -> None:
    """What a float expression of brisk.cpp means (brisk_oracle.h): BRISK_FP_X87 (default: the reference as built, Visual
    Studio 2010 Win32) or BRISK_FP_SSE.  Process-wide."""
    lib().mo_brisk_set_fp_model(int(model))


def brisk_get_fp_model() -> int:
    return int(lib().mo_brisk_get_fp_model())


def brisk_detect(img, threshold=30, octaves=3, cap=1 << 20) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().mo_brisk_detect(_u8(img), W, W, H, threshold, octaves, out.ctypes.data, cap)
    assert n <= cap
    return out[:n].copy()


def brisk_halfsample(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h // 2, w // 2), np.uint8)
    lib().mo_brisk_halfsample(_u8(src), w, h, _u8(dst))
    return dst


def brisk_twothirdsample(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((2 * (h // 3), 2 * (w // 3)), np.uint8)
    lib().mo_brisk_twothirdsample(_u8(src), w, h, _u8(dst))
    return dst


def oast_score(img, x, y, bmin=0):
    img = np.ascontiguousarray(img, np.uint8)
    p = C.cast(img.ctypes.data + y * img.shape[1] + x, C.POINTER(C.c_uint8))
    return lib().mo_oast9_16_score(p, img.shape[1], bmin)


def agast58_score(img, x, y, bmin=0):
    img = np.ascontiguousarray(img, np.uint8)
    p = C.cast(img.ctypes.data + y * img.shape[1] + x, C.POINTER(C.c_uint8))
    return lib().mo_agast5_8_score(p, img.shape[1], bmin)


def oast_detect(img, b):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    n = lib().mo_oast9_16_detect(_u8(img), w, h, b, None, 0)
    xy = np.zeros((max(n, 1), 2), np.int32)
    lib().mo_oast9_16_detect(_u8(img), w, h, b, _i32(xy), n)
    return xy[:n]


def brisk_subpixel2d(s9):
    s = np.ascontiguousarray(s9, np.int32)
    dx, dy = C.c_float(), C.c_float()
    m = lib().mo_brisk_subpixel2d(_i32(s), C.byref(dx), C.byref(dy))
    return np.float32(m), np.float32(dx.value), np.float32(dy.value)


def brisk_refine1d(variant, s_05, s0, s05):
    mx = C.c_float()
    r = lib().mo_brisk_refine1d(variant, float(np.float32(s_05)), float(np.float32(s0)), float(np.float32(s05)), C.byref(mx))
    return np.float32(r), np.float32(mx.value)
