"""Known-answer tests for the detector oracle (oracle/brisk_oracle.c).

The reference holds no vectors for the detector and its sources do not build here (see brisk_oracle.h), so the
oracle is pinned by independent formulations written here: the SIMD instruction sequences of the two resamplers
emulated literally in numpy, the segment test as a max-of-min over arcs (no bisection), exact quadratics for the
sub-pixel fits, and synthetic blobs with a known position and scale for the whole detector.
"""
import numpy as np
import pytest

import oracle_lib as O
from mofreak_amd import synth


# ---------------------------------------------------------------- SSE emulation of the resamplers
def _avg(a, b):
    return ((a.astype(np.int32) + b.astype(np.int32) + 1) >> 1).astype(np.uint8)


def _halfsample_simd(src):
    """brisk.cpp:1840-1972 instruction by instruction: 16-byte loads, _mm_avg_epu8, shift/mask/packus, scalar tails."""
    h, w = src.shape
    flat = np.concatenate([src.reshape(-1), np.zeros(64, np.uint8)])
    dw, dh = w // 2, h // 2
    dst = np.zeros(dh * dw + 64, np.uint8)
    leftover, noleftover = (w % 16) // 2, w % 16 == 0
    hsize = w // 16
    end, half_end = hsize // 2, hsize % 2 == 1
    for row in range(dh):
        p1, p2, pd = 2 * row * w, (2 * row + 1) * w, row * dw
        for _ in range(end):
            r1 = _avg(flat[p1:p1 + 16], flat[p2:p2 + 16])
            r2 = _avg(flat[p1 + 16:p1 + 32], flat[p2 + 16:p2 + 32])
            p1 += 32
            p2 += 32
            even = np.concatenate([r1[0::2], r2[0::2]])                      # and(mask) + packus
            odd = np.concatenate([np.append(r1[1:], 0)[0::2], np.append(r2[1:], 0)[0::2]])  # srli 1, and, packus
            dst[pd:pd + 16] = _avg(even, odd)
            pd += 16
        if half_end:
            r1 = _avg(flat[p1:p1 + 16], flat[p2:p2 + 16]).astype(np.int32)
            p1 += 16
            p2 += 16
            dst[pd:pd + 8] = (r1[0::2] + r1[1::2]) // 2
            pd += 8
        if not noleftover:
            for k in range(leftover):
                dst[pd] = (int(flat[p1 + k]) + int(flat[p1 + k + 1]) + int(flat[p2 + k]) + int(flat[p2 + k + 1])) // 4
                pd += 1
    return dst[:dh * dw].reshape(dh, dw)


def _shuffle(v, mask):
    out = np.zeros(16, np.uint8)
    for i, m in enumerate(mask):
        if not (m & 0x80):
            out[i] = v[m & 15]
    return out


def _twothird_simd(src):
    """brisk.cpp:1974-2065: _mm_shuffle_epi8 with the three masks, full / masked stores, scalar remainder."""
    h, w = src.shape
    set_epi8 = lambda *a: list(reversed(a))  # _mm_set_epi8 lists byte 15 first
    mask1 = set_epi8(0x80, 0x80, 0x80, 0x80, 0x80, 0x80, 0x80, 12, 0x80, 10, 0x80, 7, 0x80, 4, 0x80, 1)
    mask2 = set_epi8(0x80, 0x80, 0x80, 0x80, 0x80, 0x80, 12, 0x80, 10, 0x80, 7, 0x80, 4, 0x80, 1, 0x80)
    mask = set_epi8(0x80, 0x80, 0x80, 0x80, 0x80, 0x80, 14, 12, 11, 9, 8, 6, 5, 3, 2, 0)
    flat = np.concatenate([src.reshape(-1), np.zeros(64, np.uint8)])
    dw, dh = (w // 3) * 2, (h // 3) * 2
    dst = np.zeros(dh * dw + 64, np.uint8)
    leftover = ((w // 3) * 3) % 15
    hsize = w // 15
    row = row_dest = 0
    while (row + 2) * w < w * h:
        p1, p2, p3 = row * w, (row + 1) * w, (row + 2) * w
        d1, d2 = row_dest * dw, (row_dest + 1) * dw
        for i in range(hsize):
            first, second, third = flat[p1:p1 + 16], flat[p2:p2 + 16], flat[p3:p3 + 16]
            res = []
            for a in (first, third):
                mixed = _avg(_avg(a, second), a)
                t1 = _shuffle(mixed, mask1) | _shuffle(mixed, mask2)
                t2 = _shuffle(mixed, mask)
                res.append(_avg(_avg(t2, t1), t2))
            n = 10 if i * 10 + 16 > dw else 16  # masked store keeps ten bytes
            dst[d1:d1 + n] = res[0][:n]
            dst[d2:d2 + n] = res[1][:n]
            p1 += 15; p2 += 15; p3 += 15; d1 += 10; d2 += 10
        for _ in range(0, leftover, 3):
            A, B, Cc = flat[p1:p1 + 3].astype(int), flat[p2:p2 + 3].astype(int), flat[p3:p3 + 3].astype(int)
            p1 += 3; p2 += 3; p3 += 3
            dst[d1] = ((4 * A[0] + 2 * (A[1] + B[0]) + B[1]) // 9) & 0xff
            dst[d1 + 1] = ((4 * A[2] + 2 * (A[1] + B[2]) + B[1]) // 9) & 0xff
            dst[d2] = ((4 * Cc[0] + 2 * (Cc[1] + B[0]) + B[1]) // 9) & 0xff
            dst[d2 + 1] = ((4 * Cc[2] + 2 * (Cc[1] + B[2]) + B[1]) // 9) & 0xff
            d1 += 2; d2 += 2
        row += 3
        row_dest += 2
    return dst[:dh * dw].reshape(dh, dw)


@pytest.mark.parametrize("w,h", [(32, 8), (48, 10), (64, 7), (80, 6), (212, 12), (106, 9), (53, 8), (1920, 4), (100, 5)])
def test_halfsample_matches_the_sse_sequence(w, h):
    src = np.random.default_rng(w * 1000 + h).integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(O.brisk_halfsample(src), _halfsample_simd(src))


@pytest.mark.parametrize("w,h", [(45, 6), (320, 9), (1920, 6), (100, 7), (47, 12), (160, 10), (33, 3)])
def test_twothirdsample_matches_the_ssse3_sequence(w, h):
    src = np.random.default_rng(w * 1000 + h).integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(O.brisk_twothirdsample(src), _twothird_simd(src))


def test_pyramid_shapes_scales_offsets():
    b = O.Brisk(np.zeros((1080, 1920), np.uint8), 3)
    got = [b.layer_info(i) for i in range(b.n_layers)]
    assert got == [(1920, 1080, 1.0, 0.0), (1280, 720, 1.5, 0.25), (960, 540, 2.0, 0.5), (640, 360, 3.0, 1.0),
                   (480, 270, 4.0, 1.5), (320, 180, 6.0, 2.5)]
    assert O.Brisk(np.zeros((60, 80), np.uint8), 0).n_layers == 1


# ---------------------------------------------------------------- segment tests without bisection
C16 = [(-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3),
       (-1, 3), (-2, 2), (-3, 1)]
C8 = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]


def _score_maxmin(img, circle, arc):
    """max b in [1, 254] with an arc of `arc` pixels all > c + b or all < c - b  ==  max over arcs of min |diff| - 1."""
    n = len(circle)
    r = max(abs(d) for p in circle for d in p)
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[r:h - r, r:w - r]
    ring = np.stack([I[r + dy:h - r + dy, r + dx:w - r + dx] - c for dx, dy in circle])
    best = np.zeros_like(c)
    for s in range(n):
        idx = [(s + k) % n for k in range(arc)]
        best = np.maximum(best, np.maximum(ring[idx].min(0), (-ring[idx]).min(0)))
    out = np.zeros((h, w), np.int32)
    out[r:h - r, r:w - r] = np.maximum(best - 1, 0)
    return out


def _test_image(seed, h=40, w=56):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 40, (h, w)).astype(np.uint8)
    for _ in range(12):
        x, y, a, b = rng.integers(0, w), rng.integers(0, h), rng.integers(2, 14), rng.integers(2, 14)
        img[y:y + b, x:x + a] = rng.integers(0, 256)
    return img


def test_oast_and_agast_scores_equal_the_max_min_form():
    for seed in range(4):
        img = _test_image(seed)
        want16, want8 = _score_maxmin(img, C16, 9), _score_maxmin(img, C8, 5)
        h, w = img.shape
        got16 = np.zeros_like(want16)
        got8 = np.zeros_like(want8)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                got16[y, x] = O.oast_score(img, x, y, 0)
        for y in range(1, h - 1):
            for x in range(1, w - 1):
                got8[y, x] = O.agast58_score(img, x, y, 0)
        assert np.array_equal(got16, want16) and np.array_equal(got8, want8)
        assert want16.max() > 50 and (want16 > 0).mean() > 0.02
        # with a floor b the bisection returns max(b, score): what getAgastPoints caches for detected points
        ys, xs = np.nonzero(want16 >= 30)
        for x, y in zip(xs[:50], ys[:50]):
            assert O.oast_score(img, int(x), int(y), 30) == want16[y, x]
        assert O.oast_score(img, 5, 5, 30) == max(30, want16[5, 5])


def test_oast_detect_is_the_thresholded_score_in_raster_order():
    img = _test_image(11, 48, 70)
    S = _score_maxmin(img, C16, 9)
    for b in (30, 8, 1):
        ys, xs = np.nonzero(S[3:-3, 3:-3] >= b)
        want = np.stack([xs + 3, ys + 3], 1)
        assert np.array_equal(O.oast_detect(img, b), want) and len(want) > 5


# ---------------------------------------------------------------- sub-pixel fits
def _patch(fn):
    """s_0_0, s_0_1, s_0_2, s_1_0, ...: first index x, second y, both in {-1, 0, 1}."""
    return [int(fn(x, y)) for x in (-1, 0, 1) for y in (-1, 0, 1)]


def test_subpixel2d_recovers_an_exact_paraboloid():
    m, dx, dy = O.brisk_subpixel2d(_patch(lambda x, y: 200 - (4 * x - 1) ** 2 - 2 * (2 * y + 1) ** 2))
    assert abs(dx - 0.25) < 1e-6 and abs(dy + 0.5) < 1e-6 and abs(m - 200) < 1e-4
    m, dx, dy = O.brisk_subpixel2d(_patch(lambda x, y: 90 - 10 * x * x - 10 * y * y))
    assert (m, dx, dy) == (90, 0, 0)
    m, dx, dy = O.brisk_subpixel2d([50] * 9)  # flat: H_det == 0
    assert (dx, dy) == (0, 0) and abs(m - 50) < 1e-5
    # a saddle has no interior maximum: the best patch corner wins
    m, dx, dy = O.brisk_subpixel2d(_patch(lambda x, y: 100 + 5 * x * x - 3 * y * y + 2 * x - y))
    assert (dx, dy) == (1, -1) and abs(m - 105) < 1e-4
    # a plane degenerates (H_det == 0): centre value of the fit, no offset
    m, dx, dy = O.brisk_subpixel2d(_patch(lambda x, y: 100 + 10 * x - 5 * y))
    assert (dx, dy) == (0, 0) and abs(m - 100) < 1e-4
    # vertex outside the patch in x: clamped candidates; the reference returns delta_y = delta_x there (sic)
    m, dx, dy = O.brisk_subpixel2d(_patch(lambda x, y: 300 - (x - 2) ** 2 * 8 - 3 * y * y))
    assert dx == 1 and dy == dx


@pytest.mark.parametrize("variant,xs", [(0, (0.75, 1.0, 1.5)), (1, (2 / 3, 1.0, 4 / 3)), (2, (0.5, 1.0, 1.5))])
def test_refine1d_recovers_the_vertex_of_a_parabola(variant, xs):
    for peak in (0.9, 1.0, 1.2):
        f = lambda s: 80.0 - 40.0 * (s - peak) ** 2
        r, mx = O.brisk_refine1d(variant, f(xs[0]), f(xs[1]), f(xs[2]))
        assert abs(r - peak) < 2e-3 and abs(mx - 80.0) < 5e-2
    # convex samples: the largest sample and its fixed scale
    r, mx = O.brisk_refine1d(variant, 10, 5, 30)
    assert mx == 30 and r == np.float32([1.5, 1.3333333333333333, 1.5][variant])
    r, mx = O.brisk_refine1d(variant, 40, 5, 30)
    assert mx == 40 and r == np.float32([0.75, 0.6666666666666666, 0.7][variant])
    # vertex below the range: clamped
    f = lambda s: 80.0 - 40.0 * (s - 0.2) ** 2
    r, _ = O.brisk_refine1d(variant, f(xs[0]), f(xs[1]), f(xs[2]))
    assert r == np.float32([0.75, 0.6666666666666666, 0.7][variant])


# ---------------------------------------------------------------- the whole detector
def test_detector_finds_blob_corners_with_plausible_scale():
    img = np.zeros((120, 160), np.uint8)
    img[40:80, 50:110] = 200  # one bright rectangle: four corners
    k = O.brisk_detect(img, 30, 3)
    assert len(k) >= 4
    for cx, cy in [(50, 40), (109, 40), (50, 79), (109, 79)]:
        d = np.hypot(k["x"] - cx, k["y"] - cy)
        assert d.min() < 3.0
    assert np.all(k["size"] >= 12 * 0.7 - 1e-3) and np.all(k["response"] > 30)
    assert np.all(np.diff(k["layer"]) >= 0)  # layer-major emission order


def test_detector_on_moving_objects_is_deterministic_and_layered():
    fr = synth.moving_objects_stack(6, 320, 240)
    d = O.absdiff(fr[5], fr[0])
    b = O.Brisk(d)
    k1 = b.get_keypoints(30)
    k2 = O.brisk_detect(d, 30, 3)
    assert k1.tobytes() == k2.tobytes() and len(k1) > 100
    assert set(np.unique(k1["layer"])) == set(range(6))
    # the cache holds at least the detected points' scores, and only values the max-min form gives
    for i in range(b.n_layers):
        img, sc = b.layer_image(i), b.layer_scores(i)
        S = _score_maxmin(img, C16, 9)
        pts = b.layer_points(i)
        assert np.all(sc[pts[:, 1], pts[:, 0]] >= 30)
        filled = sc > 0
        assert np.array_equal(sc[filled], S[filled].astype(np.uint8))
    # no keypoints on an image without structure above the threshold
    assert len(O.brisk_detect(O.absdiff(*synth.synth_stack(6, 160, 120)[[5, 0]]), 30, 3)) == 0
