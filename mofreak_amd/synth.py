"""Deterministic synthetic inputs for the MoFREAK path.

The reference ships no data and there is no video decoder on either box, so every test and
benchmark feeds gray u8 frame stacks generated here, on the host, from a seed:

    E(x,y)   = 0.5 + 0.5 sin(2pi x/211) sin(2pi y/173)
    F_t(x,y) = clamp_u8( rha(128 + E (40 sin(2pi(x+0.5t)/97) + 30 sin(2pi(y-0.3t)/61)
                                      + 20 sin(2pi(x+y+0.8t)/29))) + n )
    n in [-3, 3] from splitmix64(SEED ^ (t<<40 | y<<20 | x)),  rha = round half away from zero.

The constants differ from the ones SURVEY.md 8(d) proposed (60/45/25 amplitudes, periods 37/23/11,
speeds 3/2/5, noise +-12): measured on the oracle those saturate 99.97% of the MIP bits to 1, which
would leave the motion half of the descriptor untested.  With the envelope E and sub-pixel speeds the
C2 grid gives 55% ones in the motion bytes, 56% in the appearance bytes and 112 distinct FREAK
orientations.

Pair i of a stack is (cur = F[i+gap], prev = F[i]) -- the pairing of
MoFREAKUtilities::computeMoFREAKFromFile (reference MoFREAKUtilities.cpp:378, 391-401, 485-488).
"""
from __future__ import annotations

import numpy as np

SEED = 0x4D6F465245414B  # "MoFREAK"
GAP_FOR_FRAME_DIFFERENCE = 5  # reference MoFREAKUtilities.cpp:378


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_frame(t: int, W: int, H: int, seed: int = SEED) -> np.ndarray:
    """One H x W uint8 frame F_t."""
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    a = 40.0 * np.sin(2.0 * np.pi * (x + 0.5 * t) / 97.0)
    b = 30.0 * np.sin(2.0 * np.pi * (y - 0.3 * t) / 61.0)
    xy = np.arange(W + H, dtype=np.float64)
    c = 20.0 * np.sin(2.0 * np.pi * (xy + 0.8 * t) / 29.0)
    xi = np.arange(W, dtype=np.int64)
    yi = np.arange(H, dtype=np.int64)
    env = 0.5 + 0.5 * np.sin(2.0 * np.pi * x / 211.0)[None, :] * np.sin(2.0 * np.pi * y / 173.0)[:, None]
    v = 128.0 + env * (a[None, :] + b[:, None] + c[(yi[:, None] + xi[None, :])])
    r = np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)
    key = (np.uint64(t) << np.uint64(40)) | (yi[:, None].astype(np.uint64) << np.uint64(20)) | xi[None, :].astype(np.uint64)
    h = _splitmix64(np.uint64(seed) ^ key)
    n = (h % np.uint64(7)).astype(np.int64) - 3
    return np.clip(r + n, 0, 255).astype(np.uint8)


def synth_stack(T: int, W: int, H: int, t0: int = 0, seed: int = SEED) -> np.ndarray:
    """T x H x W uint8 stack F_{t0} .. F_{t0+T-1}."""
    out = np.empty((T, H, W), dtype=np.uint8)
    for t in range(T):
        out[t] = synth_frame(t0 + t, W, H, seed)
    return out


def dense_grid(W: int, H: int, step: int, size: float, lo: int, hi_x: int | None = None,
               hi_y: int | None = None) -> np.ndarray:
    """Keypoints (x, y, size) float32 on the grid x = step*i, y = step*j with lo < x < hi_x, lo < y < hi_y.

    Row-major (y outer, x inner), the order a detector sweeping the image would produce.
    Defaults hi_x = W - lo, hi_y = H - lo.
    """
    hi_x = W - lo if hi_x is None else hi_x
    hi_y = H - lo if hi_y is None else hi_y
    xs = [x for x in range(0, W, step) if lo < x < hi_x]
    ys = [y for y in range(0, H, step) if lo < y < hi_y]
    kp = np.empty((len(ys) * len(xs), 3), dtype=np.float32)
    kp[:, 0] = np.tile(np.asarray(xs, dtype=np.float32), len(ys))
    kp[:, 1] = np.repeat(np.asarray(ys, dtype=np.float32), len(xs))
    kp[:, 2] = np.float32(size)
    return kp


# BASELINE.json configs: name -> (W, H, grid step, keypoint size, lo bound)
CONFIGS = {
    "C1": dict(W=320, H=240, step=16, size=7.0, lo=23),
    "C2": dict(W=640, H=480, step=16, size=12.0, lo=38),
    "C3": dict(W=1920, H=1080, step=8, size=12.0, lo=38),
    "C4": dict(W=320, H=240, step=16, size=12.0, lo=38),
    "C5": dict(W=720, H=576, step=8, size=12.0, lo=38),
}


def clip_lengths(n_clips: int, seed: int = 0x4D6F4652, median: float = 80.0, sigma: float = 0.6, lo: int = 20, hi: int = 650) -> np.ndarray:
    """Frame counts of an HMDB51-shaped batch (BASELINE config 4): seeded log-normal, median 80 frames,
    clamped to [20, 650] -- uneven on purpose, to exercise the load balance of the one-video-per-GPU sharding."""
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(median), sigma, n_clips)), lo, hi).astype(np.int64)


def clip_pool(n_distinct: int, max_len: int, W: int, H: int) -> list[np.ndarray]:
    """A few distinct synthetic clips of max_len frames; clip i of a batch is pool[i % n][:length_i] (generating
    thousands of distinct clips on the host would dominate a benchmark run; the device work does not depend on it)."""
    return [np.stack([synth_frame(10_000 * k + t, W, H) for t in range(max_len)]) for k in range(n_distinct)]


def config_grid(name: str) -> np.ndarray:
    c = CONFIGS[name]
    return dense_grid(c["W"], c["H"], c["step"], c["size"], c["lo"])


def random_keypoints(rng: np.random.Generator, n: int, W: int, H: int,
                     sizes=(8.4, 12.0, 18.0, 27.0, 40.5), integer_xy: bool = False) -> np.ndarray:
    """n keypoints spread over the whole image (so some fall to FREAK's border filter) with mixed sizes."""
    kp = np.empty((n, 3), dtype=np.float32)
    kp[:, 0] = rng.uniform(-4, W + 4, n)
    kp[:, 1] = rng.uniform(-4, H + 4, n)
    if integer_xy:
        kp[:, :2] = np.floor(kp[:, :2])
    kp[:, 2] = rng.choice(np.asarray(sizes, dtype=np.float32), n)
    return kp


def moving_objects_stack(T: int, W: int, H: int, n_objects: int | None = None, seed: int = 7) -> np.ndarray:
    """T gray frames of high-contrast rectangles and discs drifting over the textured background of synth_stack.

    The smooth synthetic frames above have no frame difference large enough for a corner detector (|cur - prev|
    peaks near 30); the detector row needs moving structure with real corners at several
    scales.  Objects: about one per 80 x 80 px, sides 6..64 px, gray 0..255, speed up to 3 px/frame.
    """
    rng = np.random.default_rng(seed)
    if n_objects is None:
        n_objects = max(4, (W * H) // 6400)
    kind = rng.integers(0, 2, n_objects)
    cx, cy = rng.uniform(0, W, n_objects), rng.uniform(0, H, n_objects)
    sx, sy = rng.uniform(6, 64, n_objects), rng.uniform(6, 64, n_objects)
    vx, vy = rng.uniform(-3, 3, n_objects), rng.uniform(-3, 3, n_objects)
    gray = rng.integers(0, 256, n_objects)
    base = synth_stack(T, W, H)
    out = np.empty((T, H, W), np.uint8)
    for t in range(T):
        f = base[t].copy()
        for k in range(n_objects):
            x0, y0 = cx[k] + vx[k] * t, cy[k] + vy[k] * t
            x0, y0 = x0 % W, y0 % H
            # only the object's bounding box is touched (clipped at the frame border)
            xa, xb = max(int(np.floor(x0 - sx[k] / 2)), 0), min(int(np.ceil(x0 + sx[k] / 2)) + 1, W)
            ya, yb = max(int(np.floor(y0 - sy[k] / 2)), 0), min(int(np.ceil(y0 + sy[k] / 2)) + 1, H)
            if xa >= xb or ya >= yb:
                continue
            yy, xx = np.mgrid[ya:yb, xa:xb]
            if kind[k] == 0:
                m = (np.abs(xx - x0) <= sx[k] / 2) & (np.abs(yy - y0) <= sy[k] / 2)
            else:
                m = ((xx - x0) / (sx[k] / 2)) ** 2 + ((yy - y0) / (sy[k] / 2)) ** 2 <= 1.0
            f[ya:yb, xa:xb][m] = gray[k]
        out[t] = f
    return out
