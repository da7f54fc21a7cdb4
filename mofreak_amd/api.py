"""ctypes binding of libmofreak_hip.so -- the C ABI declared in include/mofreak_hip.h.

This module is plumbing: it hands pointers to the C ABI and nothing else.  Device memory comes from
torch tensors (``tensor.data_ptr()``); there is NO CPU implementation behind it -- if the shared library
is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# MOFREAK_HIP_LIBRARY points at another build of the same ABI (kernel experiments); default: the in-tree build
LIB_PATH = os.environ.get("MOFREAK_HIP_LIBRARY") or os.path.join(PKG_DIR, "libmofreak_hip.so")

OK = 0
ERR_BAD_ARG, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_ROI, ERR_CAPACITY = -1, -2, -3, -4, -5, -6, -7
MEM_DEVICE, MEM_HOST, ROWS_DEVICE = 0, 1, 2
BITS_SSE, BITS_NATURAL, BITS_SSE_SIGNED = 0, 1, 2
FP_X87, FP_SSE = 0, 1  # mofreak_params.brisk_fp_model
TABLES_ONLY = -1
PATH_AUTO, PATH_GATHER = 0, 1

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4")])
ROW_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("frame_number", "<i4"), ("scale", "<f4"),
                      ("appearance", "u1", (8,)), ("motion", "u1", (8,))])
assert ROW_DTYPE.itemsize == 32

# every symbol include/mofreak_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mofreak_abi_version", "mofreak_build_flags", "mofreak_default_params", "mofreak_create", "mofreak_destroy", "mofreak_last_error",
    "mofreak_set_stream", "mofreak_synchronize", "mofreak_reserve", "mofreak_check_status",
    "mofreak_set_profiling", "mofreak_get_profile", "mofreak_set_path", "mofreak_get_tile_stamps", "mofreak_bgr_to_gray", "mofreak_bow_assign", "mofreak_bow_histogram",
    "mofreak_extract_pairs", "mofreak_compact_rows", "mofreak_extract_stream", "mofreak_format_rows", "mofreak_format_rows_device",
    "mofreak_extract_stream_pipelined", "mofreak_extract_clips", "mofreak_compute_clips", "mofreak_host_alloc", "mofreak_host_free",
    "mofreak_device_alloc", "mofreak_device_free", "mofreak_copy_to_host",
    "mofreak_parse_rows", "mofreak_diff_integral", "mofreak_mip19", "mofreak_roi19", "mofreak_freak_info",
    "mofreak_theta_index", "mofreak_pattern_sizes", "mofreak_scale_index", "mofreak_table_pattern",
    "mofreak_table_orientation", "mofreak_table_bit_pairs", "mofreak_table_resize", "mofreak_table_mip_positions",
    "mofreak_detect_pairs", "mofreak_detect_set_capacity", "mofreak_brisk_pyramid", "mofreak_compute_stream", "mofreak_stream_open", "mofreak_stream_push", "mofreak_stream_push_frames", "mofreak_stream_frames", "mofreak_stream_close",
]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("gap_for_frame_difference", C.c_int32), ("mip_theta", C.c_int32),
                ("freak_pattern_scale", C.c_float), ("freak_n_octaves", C.c_int32),
                ("freak_orientation_normalized", C.c_int32), ("freak_scale_normalized", C.c_int32),
                ("freak_bit_mode", C.c_int32), ("brisk_fp_model", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("bin_ms", C.c_double), ("tile_ms", C.c_double), ("gather_ms", C.c_double), ("calls", C.c_int64),
                ("pairs", C.c_int64), ("descriptors", C.c_int64)]


class MoFREAKError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmofreak_hip error {code}: {msg}")
        self.code = code


_lib = None
_PINNED: dict = {}  # address of a host_alloc() array -> the finalizer that releases its pages


def _free_pinned(lib, address: int) -> None:
    lib.mofreak_host_free(None, C.c_void_p(address))  # ctx NULL: page-locked memory outlives the context it came from


def load() -> C.CDLL:
    """Load libmofreak_hip.so (built in-tree by mofreak_amd.build / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MoFREAKError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run `python -m mofreak_amd.build` "
                           "(hipcc, gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64, and a
    # process that loads /opt/rocm's copy first and torch's second ends up with two HSA runtimes, the second of
    # which sees no GPU.  Importing torch first makes our NEEDED libamdhip64.so.7 bind to the copy torch loaded.
    if os.environ.get("MOFREAK_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.mofreak_abi_version.restype = i32
    L.mofreak_default_params.argtypes = [C.POINTER(Params)]
    L.mofreak_create.argtypes = [i32, C.POINTER(Params), C.POINTER(vp)]
    L.mofreak_destroy.argtypes = [vp]
    L.mofreak_destroy.restype = None
    L.mofreak_last_error.argtypes = [vp]
    L.mofreak_last_error.restype = C.c_char_p
    L.mofreak_set_stream.argtypes = [vp, vp]
    L.mofreak_synchronize.argtypes = [vp]
    L.mofreak_reserve.argtypes = [vp, i32, i32, i32]
    L.mofreak_check_status.argtypes = [vp]
    L.mofreak_set_profiling.argtypes = [vp, i32]
    L.mofreak_set_path.argtypes = [vp, i32]
    L.mofreak_get_tile_stamps.argtypes = [vp, vp, i32, i32]
    L.mofreak_get_profile.argtypes = [vp, C.POINTER(Profile), i32]
    L.mofreak_bgr_to_gray.argtypes = [vp, vp, i32, i32, i64, i64, i32, vp, C.c_uint]
    L.mofreak_bow_assign.argtypes = [vp, vp, vp, i64, vp, i32, vp, C.c_uint]
    L.mofreak_bow_histogram.argtypes = [vp, vp, vp, i64, vp, i32, vp, C.POINTER(C.c_int32), C.c_uint]
    L.mofreak_extract_pairs.argtypes = [vp, vp, vp, i32, i32, i64, i64, i32, vp, vp, i64, vp, vp, C.c_uint]
    L.mofreak_detect_pairs.argtypes = [vp, vp, vp, i32, i32, i64, i64, i32, i32, i32, vp, i64, vp, vp, vp, C.POINTER(i64), C.c_uint]
    L.mofreak_detect_set_capacity.argtypes = [vp, i32]
    L.mofreak_stream_open.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.mofreak_stream_push.argtypes = [vp, vp, i32, i64, vp, i64, vp, i64, C.POINTER(i64), C.c_uint]
    L.mofreak_stream_push_frames.argtypes = [vp, vp, i32, i32, vp, i64, vp, i64, C.POINTER(i64)]
    L.mofreak_stream_frames.argtypes = [vp]
    L.mofreak_stream_frames.restype = i64
    L.mofreak_stream_close.argtypes = [vp]
    L.mofreak_stream_close.restype = None
    L.mofreak_compute_stream.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i64, C.POINTER(i64), C.POINTER(i64), C.c_uint]
    L.mofreak_brisk_pyramid.argtypes = [vp, vp, i32, i32, i64, i32, vp, vp, vp, vp, C.POINTER(C.c_int), C.c_uint]
    L.mofreak_compact_rows.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, i64, C.POINTER(i64), C.c_uint]
    L.mofreak_extract_stream.argtypes = [vp, vp, i32, i32, i32, vp, vp, i64, vp, i64, C.POINTER(i64), C.c_uint]
    L.mofreak_extract_stream_pipelined.argtypes = [vp, vp, i32, i32, i32, i32, vp, i64, vp, i64, C.POINTER(i64)]
    L.mofreak_extract_clips.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i64, vp, i64, vp, C.POINTER(i64), C.c_uint]
    L.mofreak_compute_clips.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, C.POINTER(i64), C.POINTER(i64), C.c_uint]
    L.mofreak_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mofreak_host_free.argtypes = [vp, vp]
    L.mofreak_format_rows.argtypes = [vp, i64, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mofreak_format_rows_device.argtypes = [vp, vp, i64, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_int, vp]
    L.mofreak_parse_rows.argtypes = [C.c_char_p, C.c_size_t, vp, i64, C.POINTER(i64)]
    L.mofreak_diff_integral.argtypes = [vp, vp, vp, i32, i32, i64, i64, i32, vp, C.c_uint]
    L.mofreak_mip19.argtypes = [vp, vp, vp, i64, vp, C.c_uint]
    L.mofreak_roi19.argtypes = [vp, vp, vp, i32, i32, vp, i64, vp, C.c_uint]
    L.mofreak_freak_info.argtypes = [vp, vp, vp, i32, i32, vp, i64, vp, C.c_uint]
    L.mofreak_theta_index.argtypes = [vp, vp, i64, vp, C.c_uint]
    L.mofreak_pattern_sizes.argtypes = [vp, vp]
    L.mofreak_scale_index.argtypes = [vp, C.c_float, C.POINTER(C.c_int32)]
    L.mofreak_table_pattern.argtypes = [vp, i32, i32, vp]
    L.mofreak_table_orientation.argtypes = [vp, vp]
    L.mofreak_table_bit_pairs.argtypes = [vp, vp]
    L.mofreak_table_resize.argtypes = [vp, i32, vp]
    L.mofreak_table_mip_positions.argtypes = [vp, i32, vp, C.POINTER(C.c_int32)]
    _lib = L
    return L


def default_params(**overrides) -> Params:
    p = Params()
    load().mofreak_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(a) -> int:
    """Address of a numpy array, a torch tensor, or None."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch.Tensor


def _count_keypoints(kps) -> int:
    """(n, 3) float32 array/tensor, a flat one of 3n floats, or a KEYPOINT_DTYPE array."""
    if isinstance(kps, np.ndarray) and kps.dtype == KEYPOINT_DTYPE:
        return int(kps.shape[0])
    n = int(kps.shape[0])
    return n if len(kps.shape) == 2 else n // 3


def _row_capacity(rows) -> int:
    """Rows a buffer can hold: a ROW_DTYPE array, or any byte buffer / tensor (32 bytes per row)."""
    if isinstance(rows, np.ndarray):
        return int(rows.shape[0]) if rows.dtype == ROW_DTYPE else int(rows.nbytes // 32)
    return int(rows.numel() * rows.element_size() // 32)


def _is_host(*arrays) -> bool:
    kinds = set()
    for a in arrays:
        if a is None:
            continue
        kinds.add(isinstance(a, np.ndarray) or (hasattr(a, "is_cuda") and not a.is_cuda))
    if len(kinds) > 1:
        raise ValueError("mixing host and device buffers in one call")
    return kinds.pop() if kinds else True


class Context:
    """One mofreak_ctx.  ``device=TABLES_ONLY`` builds the host tables without touching a GPU."""

    def __init__(self, device: int = 0, **param_overrides):
        self._lib = load()
        self.params = default_params(**param_overrides)
        h = C.c_void_p()
        rc = self._lib.mofreak_create(device, C.byref(self.params), C.byref(h))
        if rc != OK:
            raise MoFREAKError(rc, (self._lib.mofreak_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        self._pinned: dict[int, int] = {}   # numpy data address -> page-locked allocation (host_alloc)
        self._streams: list = []            # open FrameStreams: they hold a pointer into this context
        self._det_cand_cap = 131072         # the library's default candidate capacity per pair (capi.cpp)

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            for st in list(self._streams):  # a stream must be closed before its context (mofreak_hip.h)
                st.close()
            self._lib.mofreak_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != OK:
            raise MoFREAKError(rc, (self._lib.mofreak_last_error(self._h) or b"").decode())

    # ---- stream / workspace
    def set_stream(self, stream_ptr: int | None):
        self._check(self._lib.mofreak_set_stream(self._h, stream_ptr or None))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def synchronize(self):
        self._check(self._lib.mofreak_synchronize(self._h))

    def reserve(self, W: int, H: int, chunk_pairs: int = 0):
        self._check(self._lib.mofreak_reserve(self._h, W, H, chunk_pairs))

    def check_status(self):
        self._check(self._lib.mofreak_check_status(self._h))

    def set_path(self, path: int):
        """PATH_AUTO (tile kernel + gather path for large keypoints) or PATH_GATHER (gather path for everything)."""
        self._check(self._lib.mofreak_set_path(self._h, path))

    def get_tile_stamps(self, reset: bool = True) -> np.ndarray:
        out = np.zeros(32, np.uint64)
        self._check(self._lib.mofreak_get_tile_stamps(self._h, _ptr(out), 32, int(reset)))
        return out

    def set_profiling(self, enable: bool):
        self._check(self._lib.mofreak_set_profiling(self._h, int(enable)))

    def get_profile(self, reset: bool = True) -> dict:
        p = Profile()
        self._check(self._lib.mofreak_get_profile(self._h, C.byref(p), int(reset)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    # ---- frame preparation
    def bgr_to_gray(self, bgr, W, H, n_frames, gray_out, row_stride=None, frame_stride=None):
        """bgr: (n_frames, H, W, 3) u8 numpy array or torch cuda tensor -> gray_out (n_frames, H, W)."""
        host = _is_host(bgr, gray_out)
        row_stride = 3 * W if row_stride is None else row_stride
        frame_stride = row_stride * H if frame_stride is None else frame_stride
        self._check(self._lib.mofreak_bgr_to_gray(self._h, _ptr(bgr), W, H, row_stride, frame_stride, n_frames, _ptr(gray_out),
                                                  MEM_HOST if host else MEM_DEVICE))

    def bgr_to_gray_host(self, bgr: np.ndarray) -> np.ndarray:
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if bgr.ndim == 3:
            bgr = bgr[None]
        n, H, W, _ = bgr.shape
        out = np.empty((n, H, W), np.uint8)
        self.bgr_to_gray(bgr, W, H, n, out)
        return out

    # ---- keypoint detector
    def detect_pairs(self, cur, prev, W, H, n_pairs, out_kps, out_offsets, threshold=30, octaves=3, out_response=None,
                     out_layer=None, capacity=None, row_stride=None, pair_stride=None) -> int:
        """Raw call (numpy = host, torch cuda tensors = device).  Returns the number of keypoints."""
        host = _is_host(cur, prev, out_kps, out_offsets, out_response, out_layer)
        row_stride = W if row_stride is None else row_stride
        pair_stride = W * H if pair_stride is None else pair_stride
        capacity = _count_keypoints(out_kps) if capacity is None else capacity
        n = C.c_int64(0)
        self._check(self._lib.mofreak_detect_pairs(self._h, _ptr(cur), _ptr(prev), W, H, row_stride, pair_stride, n_pairs, threshold,
                                                   octaves, _ptr(out_kps), capacity, _ptr(out_offsets), _ptr(out_response),
                                                   _ptr(out_layer), C.byref(n), MEM_HOST if host else MEM_DEVICE))
        return n.value

    def detect_pairs_host(self, cur: np.ndarray, prev, threshold=30, octaves=3, capacity=1 << 20):
        """cur, prev: (n_pairs, H, W) u8 (prev None: search cur itself) -> (kps (n,3) f32, offsets (n_pairs+1,) i64,
        response (n,) f32, layer (n,) i32) in the reference's keypoint order."""
        cur = np.ascontiguousarray(cur, np.uint8)
        if cur.ndim == 2:
            cur = cur[None]
        prev = None if prev is None else np.ascontiguousarray(prev, np.uint8).reshape(cur.shape)
        n_pairs, H, W = cur.shape
        kps = np.zeros((capacity, 3), np.float32)
        offs = np.zeros(n_pairs + 1, np.int64)
        resp = np.zeros(capacity, np.float32)
        layer = np.zeros(capacity, np.int32)
        n = self.detect_pairs(cur, prev, W, H, n_pairs, kps, offs, threshold, octaves, resp, layer, capacity=capacity)
        return kps[:n].copy(), offs, resp[:n].copy(), layer[:n].copy()

    def compute_stream(self, frames, T, W, H, rows_out, threshold=30, octaves=3, capacity=None):
        """Detector + descriptors + compaction for a gray frame stack (numpy = host, torch cuda = device).
        Returns (n_rows, n_keypoints)."""
        host = _is_host(frames, rows_out)
        capacity = int(rows_out.shape[0]) if capacity is None else capacity
        n_rows, n_kp = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.mofreak_compute_stream(self._h, _ptr(frames), T, W, H, threshold, octaves, _ptr(rows_out), capacity,
                                                     C.byref(n_rows), C.byref(n_kp), MEM_HOST if host else MEM_DEVICE))
        return n_rows.value, n_kp.value

    def compute_stream_host(self, frames: np.ndarray, threshold=30, octaves=3, capacity=None) -> np.ndarray:
        """(T, H, W) u8 -> the rows of the .mofreak file the reference would write for this clip (ROW_DTYPE)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        T, H, W = frames.shape
        capacity = max(1, T) * 8192 if capacity is None else capacity
        for _attempt in range(4):
            rows = np.zeros(capacity, ROW_DTYPE)
            n_rows, n_kp = C.c_int64(0), C.c_int64(0)
            rc = self._lib.mofreak_compute_stream(self._h, _ptr(frames), T, W, H, threshold, octaves, _ptr(rows), capacity,
                                                  C.byref(n_rows), C.byref(n_kp), MEM_HOST)
            if rc == OK:
                return rows[:n_rows.value].copy()
            if rc != ERR_CAPACITY:
                self._check(rc)
            # Two different things run out with this code: the rows buffer -- then the call says how many rows there
            # are, and exactly that many are asked for again -- or the detector's candidate list of some frame pair,
            # which a larger rows buffer cannot help: raise that capacity instead.
            if n_rows.value > capacity:
                capacity = int(n_rows.value)
            else:
                self._det_cand_cap *= 4
                if self._det_cand_cap > (1 << 24):
                    break
                self.set_detect_capacity(self._det_cand_cap)
        self._check(rc)

    def open_stream(self, W: int, H: int, use_detector: bool = True, threshold: int = 30, octaves: int = 3) -> "FrameStream":
        """Frame-at-a-time interface over a device ring of gap + 1 frames (mofreak_stream_*)."""
        return FrameStream(self, W, H, use_detector, threshold, octaves)

    def set_detect_capacity(self, candidates_per_pair: int):
        """Candidates per pair the detector reserves room for; 0: the default again -- room by frame size, grown on demand, at
        most 131072 per pair (mofreak_detect_set_capacity)."""
        self._check(self._lib.mofreak_detect_set_capacity(self._h, candidates_per_pair))
        self._det_cand_cap = int(candidates_per_pair) or 131072

    def brisk_pyramid_host(self, img: np.ndarray, octaves=3, scores=True):
        """-> list of (layer image, score map or None, scale, offset) per pyramid layer."""
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        dims = np.zeros(16, np.int32)
        so = np.zeros(16, np.float32)
        nl = C.c_int(0)
        self._check(self._lib.mofreak_brisk_pyramid(self._h, _ptr(img), W, H, W, octaves, None, None, _ptr(dims), _ptr(so),
                                                    C.byref(nl), MEM_HOST))
        sizes = [int(dims[2 * i]) * int(dims[2 * i + 1]) for i in range(nl.value)]
        layers = np.zeros(sum(sizes), np.uint8)
        sc = np.zeros(sum(sizes), np.uint8) if scores else None
        self._check(self._lib.mofreak_brisk_pyramid(self._h, _ptr(img), W, H, W, octaves, _ptr(layers), _ptr(sc), None, None, None,
                                                    MEM_HOST))
        out, o = [], 0
        for i in range(nl.value):
            w, h = int(dims[2 * i]), int(dims[2 * i + 1])
            out.append((layers[o:o + sizes[i]].reshape(h, w), None if sc is None else sc[o:o + sizes[i]].reshape(h, w),
                        np.float32(so[2 * i]), np.float32(so[2 * i + 1])))
            o += sizes[i]
        return out

    # ---- bag-of-words assignment
    def bow_assign(self, desc, codebook, out_index, valid=None, n=None):
        host = _is_host(desc, codebook, out_index, valid)
        n = int(desc.shape[0]) if n is None else n
        self._check(self._lib.mofreak_bow_assign(self._h, _ptr(desc), _ptr(valid), n, _ptr(codebook), int(codebook.shape[0]),
                                                 _ptr(out_index), MEM_HOST if host else MEM_DEVICE))

    def bow_assign_host(self, desc: np.ndarray, codebook: np.ndarray, valid=None) -> np.ndarray:
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 16)
        codebook = np.ascontiguousarray(codebook, np.uint8).reshape(-1, 16)
        out = np.zeros(desc.shape[0], np.int32)
        self.bow_assign(desc, codebook, out, valid=None if valid is None else np.ascontiguousarray(valid, np.uint8))
        return out

    def bow_histogram(self, desc, codebook, hist_out, valid=None, n=None) -> bool:
        host = _is_host(desc, codebook, hist_out, valid)
        n = int(desc.shape[0]) if n is None else n
        ok = C.c_int32(0)
        self._check(self._lib.mofreak_bow_histogram(self._h, _ptr(desc), _ptr(valid), n, _ptr(codebook), int(codebook.shape[0]),
                                                    _ptr(hist_out), C.byref(ok), MEM_HOST if host else MEM_DEVICE))
        return bool(ok.value)

    def bow_histogram_host(self, desc: np.ndarray, codebook: np.ndarray, valid=None):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 16)
        codebook = np.ascontiguousarray(codebook, np.uint8).reshape(-1, 16)
        hist = np.zeros(codebook.shape[0], np.float32)
        ok = self.bow_histogram(desc, codebook, hist, valid=None if valid is None else np.ascontiguousarray(valid, np.uint8))
        return hist, ok

    # ---- hot path
    def extract_pairs(self, cur, prev, W, H, n_pairs, kps, out_desc, out_valid, kp_offsets=None, n_kp=None,
                      row_stride=None, pair_stride=None):
        """Raw call: buffers are numpy arrays (host) or torch cuda tensors (device), all of one kind."""
        host = _is_host(cur, prev, kps, out_desc, out_valid, kp_offsets)
        row_stride = W if row_stride is None else row_stride
        pair_stride = W * H if pair_stride is None else pair_stride
        if n_kp is None:
            n_kp = _count_keypoints(kps)
        self._check(self._lib.mofreak_extract_pairs(self._h, _ptr(cur), _ptr(prev), W, H, row_stride, pair_stride,
                                                    n_pairs, _ptr(kps), _ptr(kp_offsets), n_kp, _ptr(out_desc),
                                                    _ptr(out_valid), MEM_HOST if host else MEM_DEVICE))

    def extract_pairs_host(self, cur: np.ndarray, prev: np.ndarray, kps: np.ndarray, kp_offsets=None):
        """cur, prev: (n_pairs, H, W) u8; kps: (n, 3) f32 -> (desc16 (n_out,16), valid (n_out,))."""
        cur = np.ascontiguousarray(cur, np.uint8)
        prev = np.ascontiguousarray(prev, np.uint8)
        if cur.ndim == 2:
            cur, prev = cur[None], prev[None]
        n_pairs, H, W = cur.shape
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        n_kp = kps.shape[0]
        if kp_offsets is not None:
            kp_offsets = np.ascontiguousarray(kp_offsets, np.int64)
            n_out = n_kp
        else:
            n_out = n_pairs * n_kp
        desc = np.zeros((n_out, 16), np.uint8)
        valid = np.zeros(n_out, np.uint8)
        self.extract_pairs(cur, prev, W, H, n_pairs, kps, desc, valid, kp_offsets=kp_offsets, n_kp=n_kp)
        return desc, valid

    def compact_rows(self, kps, n_pairs, first_frame_number, desc, valid, rows_out, kp_offsets=None, n_kp=None) -> int:
        host = _is_host(kps, desc, valid, rows_out, kp_offsets)
        if n_kp is None:
            n_kp = _count_keypoints(kps)
        cap = _row_capacity(rows_out)
        n = C.c_int64(0)
        self._check(self._lib.mofreak_compact_rows(self._h, _ptr(kps), _ptr(kp_offsets), n_kp, n_pairs,
                                                   first_frame_number, _ptr(desc), _ptr(valid), _ptr(rows_out), cap,
                                                   C.byref(n), MEM_HOST if host else MEM_DEVICE))
        return n.value

    def extract_stream(self, frames, T, W, H, kps, rows_out, kp_offsets=None, n_kp=None) -> int:
        host = _is_host(frames, kps, rows_out, kp_offsets)
        if n_kp is None:
            n_kp = _count_keypoints(kps)
        cap = _row_capacity(rows_out)
        n = C.c_int64(0)
        self._check(self._lib.mofreak_extract_stream(self._h, _ptr(frames), T, W, H, _ptr(kps), _ptr(kp_offsets), n_kp,
                                                     _ptr(rows_out), cap, C.byref(n), MEM_HOST if host else MEM_DEVICE))
        return n.value

    def extract_stream_host(self, frames: np.ndarray, kps: np.ndarray, kp_offsets=None) -> np.ndarray:
        """frames (T,H,W) u8, keypoints shared by all processed frames or CSR -> structured rows."""
        frames = np.ascontiguousarray(frames, np.uint8)
        T, H, W = frames.shape
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        gap = self.params.gap_for_frame_difference
        n_pairs = max(T - gap, 0)
        if kp_offsets is not None:
            kp_offsets = np.ascontiguousarray(kp_offsets, np.int64)
            cap = kps.shape[0]
        else:
            cap = n_pairs * kps.shape[0]
        rows = np.zeros(max(cap, 1), ROW_DTYPE)
        n = self.extract_stream(frames, T, W, H, kps, rows, kp_offsets=kp_offsets, n_kp=kps.shape[0])
        return rows[:n].copy()

    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """A page-locked host array (mofreak_host_alloc): frames decoded into it, or rows received into it, move by DMA
        straight from / to it in extract_stream_pipelined / extract_clips.  The memory belongs to the ARRAY, not to the
        context: it is released when the last view of it is gone (or by host_free), never by close() under a live view."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.mofreak_host_alloc(self._h, n, C.byref(p)))
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        # every view of arr keeps `buf` alive through its base chain; when buf goes, the pages go
        fin = weakref.finalize(buf, _free_pinned, self._lib, p.value)
        _PINNED[arr.ctypes.data] = fin
        return arr

    def format_rows_device(self, rows, n_rows: int | None = None, row_starts=None, out: np.ndarray | None = None):
        """writeMoFREAKFeaturesToFile's text made on the device (mofreak_format_rows_device) from rows in DEVICE memory (a torch
        CUDA tensor / anything with data_ptr(), or an int address) into a page-locked host array the device writes directly.
        row_starts: first row of every video -> returns (text array, offsets) with offsets[i] : offsets[i + 1] the text of
        video i; without it (text array, total).  `out`: a host_alloc() array to reuse (grown when too small).
        Raises MoFREAKError(ERR_UNSUPPORTED) for rows the device leaves to format_rows (see include/mofreak_hip.h)."""
        ptr = rows if isinstance(rows, int) else rows.data_ptr()
        if n_rows is None:
            n_rows = int(rows.numel() * rows.element_size() // 32)
        starts = None if row_starts is None else np.ascontiguousarray(row_starts, dtype=np.int64)
        n_seg = 0 if starts is None else len(starts)
        seg = np.zeros(n_seg + 1, dtype=np.uint64)
        need = C.c_size_t(0)
        if out is None or out.nbytes < n_rows * 80:  # a row is ~75 characters; ERR_CAPACITY below covers the rest
            out = self.host_alloc(max(int(n_rows) * 96, 256))
        for _ in range(2):
            rc = self._lib.mofreak_format_rows_device(self._h, C.c_void_p(ptr), n_rows, _ptr(out), out.nbytes, C.byref(need),
                                                      None if starts is None else _ptr(starts), n_seg, _ptr(seg) if n_seg else None)
            if rc != ERR_CAPACITY:
                break
            out = self.host_alloc(int(need.value) + 256)
        self._check(rc)
        if starts is None:
            return out, int(need.value)
        return out, seg.astype(np.int64)

    def host_free(self, arr: np.ndarray) -> None:
        """Release a host_alloc() array now (the caller promises that no view of it is used afterwards)."""
        fin = _PINNED.pop(arr.ctypes.data, None)
        if fin is not None:
            fin()

    def extract_clips(self, clips, kps: np.ndarray, chunk_frames: int = 0, rows_out=None):
        """Many gray stacks (each (T_c, H, W) u8, C-contiguous, same H x W) in one pipelined pass: mofreak_extract_clips.
        rows_out: None (a new numpy array), a numpy ROW_DTYPE array (page-locked from host_alloc: DMA in place), or a
        torch uint8 CUDA tensor (rows stay on the device, 32 bytes each).  Returns (rows or row count, clip_row_offsets):
        rows of clip c are rows[offsets[c]:offsets[c + 1]]; with a device tensor the first item is the number of rows."""
        n = len(clips)
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        gap = self.params.gap_for_frame_difference
        H, W = (clips[0].shape[1], clips[0].shape[2]) if n else (1, 1)
        ptrs = (C.c_void_p * max(n, 1))()
        counts = np.zeros(max(n, 1), np.int32)
        for i, c in enumerate(clips):
            assert c.dtype == np.uint8 and c.ndim == 3 and c.flags.c_contiguous and c.shape[1:] == (H, W), "clip: (T, H, W) uint8, C-contiguous"
            ptrs[i] = c.ctypes.data
            counts[i] = c.shape[0]
        cap = int(sum(max(int(t) - gap, 0) for t in counts[:n])) * kps.shape[0]
        on_device = rows_out is not None and not isinstance(rows_out, np.ndarray)
        if on_device:
            assert rows_out.is_cuda and rows_out.is_contiguous() and rows_out.element_size() == 1
            rows_ptr, rows_cap = C.c_void_p(rows_out.data_ptr()), rows_out.numel() // 32
        else:
            rows = rows_out if rows_out is not None else np.zeros(max(cap, 1), ROW_DTYPE)
            assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
            rows_ptr, rows_cap = _ptr(rows), rows.shape[0]
        offs = np.zeros(n + 1, np.int64)
        total = C.c_int64(0)
        self._check(self._lib.mofreak_extract_clips(self._h, C.cast(ptrs, C.c_void_p), _ptr(counts), n, W, H, chunk_frames, _ptr(kps),
                                                    kps.shape[0], rows_ptr, rows_cap, _ptr(offs), C.byref(total),
                                                    ROWS_DEVICE if on_device else 0))
        if on_device:
            return total.value, offs
        return (rows[:total.value] if rows_out is not None else rows[:total.value].copy()), offs

    def compute_clips(self, clips, threshold: int = 30, octaves: int = 3, chunk_frames: int = 0, rows_out=None, rows_per_pair: int = 8192):
        """Many gray stacks in one pipelined pass with the reference's own keypoint source, the BRISK detector on every pair's
        difference image: mofreak_compute_clips (the rows of one compute_stream_host call per clip, clip after clip).
        rows_out as in extract_clips; the number of rows is not known up front: a buffer that turns out too small (numpy: a
        new one is made; a caller's buffer: MoFREAKError ERR_CAPACITY) -- rows_per_pair sizes the first attempt.
        Returns (rows or row count, clip_row_offsets, keypoints detected)."""
        n = len(clips)
        gap = self.params.gap_for_frame_difference
        H, W = (clips[0].shape[1], clips[0].shape[2]) if n else (1, 1)
        ptrs = (C.c_void_p * max(n, 1))()
        counts = np.zeros(max(n, 1), np.int32)
        for i, c in enumerate(clips):
            assert c.dtype == np.uint8 and c.ndim == 3 and c.flags.c_contiguous and c.shape[1:] == (H, W), "clip: (T, H, W) uint8, C-contiguous"
            ptrs[i] = c.ctypes.data
            counts[i] = c.shape[0]
        n_pairs = int(sum(max(int(t) - gap, 0) for t in counts[:n]))
        on_device = rows_out is not None and not isinstance(rows_out, np.ndarray)
        offs = np.zeros(n + 1, np.int64)
        total, n_kp = C.c_int64(0), C.c_int64(0)
        rows = None
        for attempt in range(2):
            if on_device:
                assert rows_out.is_cuda and rows_out.is_contiguous() and rows_out.element_size() == 1
                rows_ptr, rows_cap = C.c_void_p(rows_out.data_ptr()), rows_out.numel() // 32
            else:
                rows = rows_out if rows_out is not None else np.zeros(max(total.value, n_pairs * rows_per_pair, 1), ROW_DTYPE)
                assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
                rows_ptr, rows_cap = _ptr(rows), rows.shape[0]
            rc = self._lib.mofreak_compute_clips(self._h, C.cast(ptrs, C.c_void_p), _ptr(counts), n, W, H, chunk_frames, threshold, octaves, rows_ptr, rows_cap,
                                                 _ptr(offs), C.byref(total), C.byref(n_kp), ROWS_DEVICE if on_device else 0)
            if rc != ERR_CAPACITY or rows_out is not None:
                break
        self._check(rc)
        if on_device:
            return total.value, offs, n_kp.value
        return (rows[:total.value] if rows_out is not None else rows[:total.value].copy()), offs, n_kp.value

    def extract_stream_pipelined_host(self, frames: np.ndarray, kps: np.ndarray, chunk_frames: int = 256,
                                      rows_out: np.ndarray | None = None) -> np.ndarray:
        """A long host-resident gray stack (T,H,W) u8 through the chunked, copy/compute-overlapped frame loop; the
        rows of extract_stream_host(frames, kps).  frames / rows_out from host_alloc() are copied by DMA in place."""
        assert frames.dtype == np.uint8 and frames.flags.c_contiguous and frames.ndim == 3
        T, H, W = frames.shape
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        cap = max(T - self.params.gap_for_frame_difference, 0) * kps.shape[0]
        rows = rows_out if rows_out is not None else np.zeros(max(cap, 1), ROW_DTYPE)
        assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
        n = C.c_int64(0)
        self._check(self._lib.mofreak_extract_stream_pipelined(self._h, _ptr(frames), T, W, H, chunk_frames, _ptr(kps),
                                                               kps.shape[0], _ptr(rows), rows.shape[0], C.byref(n)))
        return rows[:n.value] if rows_out is not None else rows[:n.value].copy()

    # ---- component entry points
    def diff_integral_host(self, cur: np.ndarray, prev: np.ndarray) -> np.ndarray:
        cur = np.ascontiguousarray(cur, np.uint8)
        prev = np.ascontiguousarray(prev, np.uint8)
        if cur.ndim == 2:
            cur, prev = cur[None], prev[None]
        n, H, W = cur.shape
        out = np.zeros((n, H + 1, W + 1), np.int32)
        self._check(self._lib.mofreak_diff_integral(self._h, _ptr(cur), _ptr(prev), W, H, W, W * H, n, _ptr(out), MEM_HOST))
        return out

    def mip19_host(self, cur19: np.ndarray, prev19: np.ndarray) -> np.ndarray:
        cur19 = np.ascontiguousarray(cur19, np.uint8).reshape(-1, 361)
        prev19 = np.ascontiguousarray(prev19, np.uint8).reshape(-1, 361)
        out = np.zeros((cur19.shape[0], 8), np.uint8)
        self._check(self._lib.mofreak_mip19(self._h, _ptr(cur19), _ptr(prev19), cur19.shape[0], _ptr(out), MEM_HOST))
        return out

    def roi19_host(self, cur: np.ndarray, prev: np.ndarray, kps: np.ndarray) -> np.ndarray:
        cur = np.ascontiguousarray(cur, np.uint8)
        prev = np.ascontiguousarray(prev, np.uint8)
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        H, W = cur.shape
        out = np.zeros((kps.shape[0], 2, 19, 19), np.uint8)
        self._check(self._lib.mofreak_roi19(self._h, _ptr(cur), _ptr(prev), W, H, _ptr(kps), kps.shape[0], _ptr(out), MEM_HOST))
        return out

    def freak_info_host(self, cur: np.ndarray, prev: np.ndarray, kps: np.ndarray) -> np.ndarray:
        cur = np.ascontiguousarray(cur, np.uint8)
        prev = np.ascontiguousarray(prev, np.uint8)
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        H, W = cur.shape
        out = np.zeros((kps.shape[0], 4), np.int32)
        self._check(self._lib.mofreak_freak_info(self._h, _ptr(cur), _ptr(prev), W, H, _ptr(kps), kps.shape[0], _ptr(out), MEM_HOST))
        return out

    def theta_index_host(self, dirs: np.ndarray) -> np.ndarray:
        dirs = np.ascontiguousarray(dirs, np.int32).reshape(-1, 2)
        out = np.zeros(dirs.shape[0], np.int32)
        self._check(self._lib.mofreak_theta_index(self._h, _ptr(dirs), dirs.shape[0], _ptr(out), MEM_HOST))
        return out

    # ---- host tables
    def pattern_sizes(self) -> np.ndarray:
        out = np.zeros(64, np.int32)
        self._check(self._lib.mofreak_pattern_sizes(self._h, _ptr(out)))
        return out

    def scale_index(self, size: float) -> int:
        v = C.c_int32(0)
        self._check(self._lib.mofreak_scale_index(self._h, float(np.float32(size)), C.byref(v)))
        return v.value

    def table_pattern(self, scale: int, rot: int) -> np.ndarray:
        out = np.zeros((43, 3), np.float32)
        self._check(self._lib.mofreak_table_pattern(self._h, scale, rot, _ptr(out)))
        return out

    def table_orientation(self) -> np.ndarray:
        out = np.zeros((45, 4), np.int32)
        self._check(self._lib.mofreak_table_orientation(self._h, _ptr(out)))
        return out

    def table_bit_pairs(self) -> np.ndarray:
        out = np.zeros((64, 2), np.uint8)
        self._check(self._lib.mofreak_table_bit_pairs(self._h, _ptr(out)))
        return out

    def table_resize(self, L: int) -> np.ndarray:
        out = np.zeros((2, 19, 4), np.int16)
        self._check(self._lib.mofreak_table_resize(self._h, L, _ptr(out)))
        return out

    def table_mip_positions(self, L: int) -> np.ndarray:
        """The tile kernel's MIP sampling order for ROI side L: positions (frame * 368 + row * 19 + col), pass-major."""
        out = np.zeros(320, np.uint16)
        n = C.c_int32(0)
        self._check(self._lib.mofreak_table_mip_positions(self._h, L, _ptr(out), C.byref(n)))
        return out[:n.value].copy()


def format_rows(rows: np.ndarray) -> bytes:
    """writeMoFREAKFeaturesToFile's text for structured rows (host only, no GPU needed)."""
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    L = load()
    need = C.c_size_t(0)
    rc = L.mofreak_format_rows(_ptr(rows), rows.shape[0], None, 0, C.byref(need))
    if rc != OK:
        raise MoFREAKError(rc, "format_rows")
    buf = C.create_string_buffer(need.value + 1)
    rc = L.mofreak_format_rows(_ptr(rows), rows.shape[0], buf, need.value, C.byref(need))
    if rc != OK:
        raise MoFREAKError(rc, "format_rows")
    return buf.raw[:need.value]


def parse_rows(text: bytes) -> np.ndarray:
    """The row parser of readMoFREAKFeatures, in file order."""
    L = load()
    n = C.c_int64(0)
    rc = L.mofreak_parse_rows(text, len(text), None, 0, C.byref(n))
    if rc != OK:
        raise MoFREAKError(rc, "parse_rows: malformed .mofreak text")
    rows = np.zeros(max(n.value, 1), ROW_DTYPE)
    rc = L.mofreak_parse_rows(text, len(text), _ptr(rows), n.value, C.byref(n))
    if rc != OK:
        raise MoFREAKError(rc, "parse_rows")
    return rows[:n.value].copy()


class FrameStream:
    """capture >> frame, one frame at a time (MoFREAKUtilities.cpp:402-411): push() returns that frame's rows."""

    def __init__(self, ctx: "Context", W, H, use_detector, threshold, octaves):
        self._ctx = ctx
        self._h = C.c_void_p()
        ctx._check(ctx._lib.mofreak_stream_open(ctx._h, W, H, int(use_detector), threshold, octaves, C.byref(self._h)))
        self.W, self.H = W, H
        ctx._streams.append(self)

    def push(self, frame: np.ndarray, kps: np.ndarray | None = None, capacity: int = 1 << 16) -> np.ndarray:
        """frame: (H, W) gray or (H, W, 3) BGR uint8 host array; kps: (n, 3) float32 when the stream has no detector."""
        frame = np.ascontiguousarray(frame, np.uint8)
        channels = 1 if frame.ndim == 2 else int(frame.shape[2])
        if frame.ndim not in (2, 3) or frame.shape[:2] != (self.H, self.W) or channels not in (1, 3):
            raise ValueError(f"frame of shape {frame.shape}: expected ({self.H}, {self.W}) gray or ({self.H}, {self.W}, 3) BGR")
        if not self._h:
            raise MoFREAKError(ERR_BAD_ARG, "stream is closed")
        k = None if kps is None else np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        rows = np.zeros(capacity, ROW_DTYPE)
        n = C.c_int64(0)
        # a frame is consumed even when its rows do not fit: size `capacity` for the densest frame expected
        self._ctx._check(self._ctx._lib.mofreak_stream_push(self._h, _ptr(frame), channels, channels * self.W, _ptr(k),
                                                            0 if k is None else len(k), _ptr(rows), capacity, C.byref(n), MEM_HOST))
        return rows[:n.value].copy()

    def push_frames(self, frames: np.ndarray, kps: np.ndarray | None = None, chunk_frames: int = 0, rows_out: np.ndarray | None = None,
                    rows_per_pair: int = 8192) -> np.ndarray:
        """A chunk of (n, H, W) gray frames at once (mofreak_stream_push_frames): the rows of the pairs it completes.  frames
        and rows_out from Context.host_alloc() move by DMA in place; rows_out None: a new array sized for the chunk.
        A stream opened with use_detector=True takes no keypoints (kps None): they are found on the device window by window;
        the number of rows is then not known up front -- rows_per_pair sizes the array made here, and a chunk whose rows do not
        fit raises ERR_CAPACITY (its frames are consumed all the same: size rows_out generously)."""
        assert frames.dtype == np.uint8 and frames.ndim == 3 and frames.flags.c_contiguous and frames.shape[1:] == (self.H, self.W)
        if not self._h:
            raise MoFREAKError(ERR_BAD_ARG, "stream is closed")
        k = np.zeros((0, 3), np.float32) if kps is None else np.ascontiguousarray(kps, np.float32).reshape(-1, 3)
        rows = rows_out if rows_out is not None else np.zeros(max(frames.shape[0] * (rows_per_pair if kps is None else len(k)), 1), ROW_DTYPE)
        assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
        n = C.c_int64(0)
        self._ctx._check(self._ctx._lib.mofreak_stream_push_frames(self._h, _ptr(frames), frames.shape[0], chunk_frames, _ptr(k) if len(k) else None, len(k), _ptr(rows),
                                                                   rows.shape[0], C.byref(n)))
        return rows[:n.value] if rows_out is not None else rows[:n.value].copy()

    @property
    def frames(self) -> int:
        return int(self._ctx._lib.mofreak_stream_frames(self._h))

    def close(self):
        if self._h:
            self._ctx._lib.mofreak_stream_close(self._h)
            self._h = C.c_void_p()
            if self in self._ctx._streams:
                self._ctx._streams.remove(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
