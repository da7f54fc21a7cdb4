"""Known-answer tests that pin the CPU oracle (KATs 1-10).

The reference ships no tests or golden vectors, so these hand-derivable cases -- plus an independent
pure-Python/numpy restatement of each stage -- are what the oracle is anchored on.  Reference lines
cited are in src/MoFREAK/MoFREAKUtilities.cpp unless stated.
"""
import math

import numpy as np
import pytest

from mofreak_amd import synth

CENTERS = [(5, 5), (5, 9), (5, 13), (9, 5), (9, 13), (13, 5), (13, 9), (13, 13)]  # :308-316 (x=col, y=row)
OFFSETS = [(-4, 0), (-3, 3), (0, 4), (3, 3), (4, 0), (3, -3), (0, -4), (-3, -3)]  # :56-70 (dx, dy)


# ------------------------------------------------------------------ independent restatements (pure Python)
def py_mip(cur, prev, x, y, theta=288):
    """motionInterchangePattern (:46-99): 9 contiguous bytes from the ROI's top-left, NOT a 3x3 block."""
    cur = np.asarray(cur, np.int64).reshape(-1)
    prev = np.asarray(prev, np.int64).reshape(-1)
    d = 0
    for i, (dx, dy) in enumerate(OFFSETS):
        a = (y - 1) * 19 + (x - 1)
        b = (y + dy - 1) * 19 + (x + dx - 1)
        ssd = int(((cur[a:a + 9] - prev[b:b + 9]) ** 2).sum())
        if ssd > theta:
            d |= 1 << i
    return d


def py_resize(src, dsize=19):
    """cv::resize 8UC1 INTER_LINEAR restated with numpy scalars."""
    src = np.asarray(src, np.int64)
    sh, sw = src.shape

    def axis(ssize, is_x):
        scale = 1.0 / (dsize / ssize)
        ofs, coef, dmax = [], [], dsize
        for d in range(dsize):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(float(f)))
            f = np.float32(f - np.float32(s))
            if is_x:
                if s < 0:
                    f, s = np.float32(0), 0
                if s + 1 >= ssize:
                    dmax = min(dmax, d)
                    if s >= ssize - 1:
                        f, s = np.float32(0), ssize - 1
            c0 = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048)))
            c1 = int(np.rint(f * np.float32(2048)))
            ofs.append(s)
            coef.append((c0, c1))
        return ofs, coef, dmax

    xofs, ialpha, xmax = axis(sw, True)
    yofs, ibeta, _ = axis(sh, False)
    out = np.zeros((dsize, dsize), np.uint8)
    for dy in range(dsize):
        r0 = min(max(yofs[dy], 0), sh - 1)
        r1 = min(max(yofs[dy] + 1, 0), sh - 1)
        b0, b1 = ibeta[dy]
        for dx in range(dsize):
            sx = xofs[dx]
            if dx < xmax:
                a0, a1 = ialpha[dx]
                t0 = src[r0, sx] * a0 + src[r0, sx + 1] * a1
                t1 = src[r1, sx] * a0 + src[r1, sx + 1] * a1
            else:
                t0, t1 = src[r0, sx] * 2048, src[r1, sx] * 2048
            out[dy, dx] = ((((b0 * (int(t0) >> 4)) >> 16) + ((b1 * (int(t1) >> 4)) >> 16) + 2) >> 2) & 0xFF
    return out


DEF_PAIRS_HEAD = [404, 431, 818, 511, 181, 52, 311, 874, 774, 543, 719, 230, 417, 205, 11, 560]
DEF_PAIRS_HEAD_IJ = [(28, 26), (29, 25), (40, 38), (32, 15), (19, 10), (10, 7), (25, 11), (42, 13), (39, 33),
                     (33, 15), (38, 16), (21, 20), (29, 11), (20, 15), (5, 1), (33, 32)]


def py_pattern(scale_idx, rot, pattern_scale=22.0, n_octaves=4):
    """patternLookup[scale][rot] (freak.cpp buildPattern) as float32 (x, y, sigma)."""
    scale_step = 2.0 ** (n_octaves / 64.0)
    scaling = scale_step ** scale_idx
    theta = float(rot) * 2 * 3.1415926535897932384626433832795 / 256.0
    big_r, small_r = 2.0 / 3.0, 2.0 / 24.0
    unit = (big_r - small_r) / 21.0
    radius = [big_r, big_r - 6 * unit, big_r - 11 * unit, big_r - 15 * unit, big_r - 18 * unit, big_r - 20 * unit,
              small_r, 0.0]
    sigma = [r / 2.0 for r in radius[:7]] + [radius[6] / 2.0]
    n = [6, 6, 6, 6, 6, 6, 6, 1]
    pts = []
    for i in range(8):
        for k in range(n[i]):
            beta = math.pi / n[i] * (i % 2)
            alpha = float(k) * 2 * math.pi / float(n[i]) + beta + theta
            pts.append((np.float32(radius[i] * math.cos(alpha) * scaling * pattern_scale),
                        np.float32(radius[i] * math.sin(alpha) * scaling * pattern_scale),
                        np.float32(sigma[i] * scaling * pattern_scale)))
    return pts


def py_mean_intensity(integ, kx, ky, P):
    """FREAK::meanIntensity, box branch, with numpy float32/float64 scalars."""
    xf = np.float32(P[0] + np.float32(kx))
    yf = np.float32(P[1] + np.float32(ky))
    r = P[2]
    x_left = int(np.float64(np.float32(xf - r)) + 0.5)
    y_top = int(np.float64(np.float32(yf - r)) + 0.5)
    x_right = int(np.float64(np.float32(xf + r)) + 1.5)
    y_bottom = int(np.float64(np.float32(yf + r)) + 1.5)
    v = int(integ[y_bottom, x_right]) - int(integ[y_bottom, x_left]) + int(integ[y_top, x_left]) - int(integ[y_top, x_right])
    return (v // ((x_right - x_left) * (y_bottom - y_top))) & 0xFF  # v >= 0: floor == C truncation


# ------------------------------------------------------------------ KAT 1-5: MIP
def test_kat1_mip_constant_frames_give_zero(oracle):
    f = np.full((19, 19), 77, np.uint8)
    for (x, y) in CENTERS + [(9, 9)]:
        assert oracle.mip(f, f, x, y) == 0


def test_kat2_mip_black_vs_white_gives_ff(oracle):
    cur = np.zeros((19, 19), np.uint8)
    prev = np.full((19, 19), 255, np.uint8)
    for (x, y) in CENTERS:
        assert oracle.mip(cur, prev, x, y) == 0xFF


def test_kat3_mip_threshold_is_strict(oracle):
    # :91 `if (ssd > THETA)` with THETA = 288 (:48)
    for i, (dx, dy) in enumerate(OFFSETS):
        x, y = 9, 9
        cur = np.zeros(361, np.uint8)
        prev = np.zeros(361, np.uint8)
        b = (y + dy - 1) * 19 + (x + dx - 1)
        prev[b] = 12
        prev[b + 1] = 12  # ssd = 288 -> bit stays 0
        assert (oracle.mip(cur, prev, x, y) >> i) & 1 == 0
        prev[b + 1] = 0
        prev[b] = 17      # ssd = 289 -> bit set
        assert (oracle.mip(cur, prev, x, y) >> i) & 1 == 1


def test_kat4_mip_strip_is_nine_contiguous_bytes(oracle):
    # :79-88 walk patch.data with p++: byte 6 of the strip is IN, the third row of a true 3x3 block is OUT
    for (x, y) in CENTERS:
        for i, (dx, dy) in enumerate(OFFSETS):
            cur = np.zeros(361, np.uint8)
            prev = np.zeros(361, np.uint8)
            prev[(y - 1 + dy) * 19 + (x - 1 + dx) + 6] = 255
            assert (oracle.mip(cur, prev, x, y) >> i) & 1 == 1, "strip byte 6 must count"
            prev[:] = 0
            prev[(y + 1 + dy) * 19 + (x + dx)] = 255
            got = (oracle.mip(cur, prev, x, y) >> i) & 1
            inside = 0 <= ((y + 1 + dy) * 19 + (x + dx)) - ((y - 1 + dy) * 19 + (x - 1 + dx)) < 9
            assert got == int(inside) == 0, "third row of a true 3x3 block must NOT count"


def test_kat4b_mip_strip_wraps_rows_for_x13(oracle):
    # centre x=13, offset (+4,0): strip starts at column 16 -> columns 16,17,18 then 0..5 of the NEXT row
    x, y, i = 13, 9, 4
    cur = np.zeros(361, np.uint8)
    prev = np.zeros(361, np.uint8)
    prev[(y - 1) * 19 + 19 + 2] = 255  # row y, column 2: strip byte 5
    assert (oracle.mip(cur, prev, x, y) >> i) & 1 == 1
    # every strip the 8 centres use stays inside the 361-byte buffer
    lo = min((y + dy - 1) * 19 + (x + dx - 1) for (x, y) in CENTERS for (dx, dy) in OFFSETS)
    hi = max((y + dy - 1) * 19 + (x + dx - 1) + 8 for (x, y) in CENTERS for (dx, dy) in OFFSETS)
    assert lo >= 0 and hi < 361


def test_kat5_mip_bit_and_byte_order(oracle):
    rng = np.random.default_rng(5)
    # byte index = centre index (:308-323), bit index = offset index (:56-70, :74, :95)
    H = W = 64
    yy, xx = np.mgrid[0:H, 0:W]
    cur = (3 * xx + 2 * yy).astype(np.uint8)  # a ramp: the 8 offsets see SSDs on both sides of THETA
    prev = (cur.astype(np.int64) + rng.integers(-2, 3, (H, W))).clip(0, 255).astype(np.uint8)
    rc, out = oracle.mip_descriptor(cur, prev, 19.0, 30, 30)  # L = 19: resize is the identity
    assert rc == 0
    roi_c = cur[30 - 9:30 - 9 + 19, 30 - 9:30 - 9 + 19]
    roi_p = prev[30 - 9:30 - 9 + 19, 30 - 9:30 - 9 + 19]
    for c, (cx, cy) in enumerate(CENTERS):
        assert out[c] == py_mip(roi_c, roi_p, cx, cy)
    assert len(set(out.tolist())) > 1


def test_mip_matches_python_restatement_on_random_buffers(oracle):
    rng = np.random.default_rng(11)
    for _ in range(50):
        cur = rng.integers(0, 256, 361, dtype=np.uint8)
        prev = (cur.astype(np.int64) + rng.integers(-8, 9, 361)).clip(0, 255).astype(np.uint8)
        for (x, y) in CENTERS + [(9, 9)]:
            assert oracle.mip(cur, prev, x, y) == py_mip(cur, prev, x, y)


# ------------------------------------------------------------------ KAT 6: resize
def test_kat6_resize_identity_and_2x2(oracle):
    rng = np.random.default_rng(6)
    src = rng.integers(0, 256, (19, 19), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(src), src)
    src = rng.integers(0, 256, (38, 38), dtype=np.uint8)
    s = src.astype(np.int64)
    want = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear(src), want)


@pytest.mark.parametrize("L", [2, 5, 9, 12, 13, 19, 27, 38, 41, 57, 100])
def test_kat6_resize_matches_python_restatement(oracle, L):
    rng = np.random.default_rng(L)
    src = rng.integers(0, 256, (L, L), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(src), py_resize(src))


def test_kat6_resize_clamps(oracle):
    # upsample L=12: dx=0 has sx=-1 -> clamped to (0, fx=0); dx=18 has sx=11=L-1 -> pure copy
    ofs, coef, xmax = oracle.resize_axis_table(12, 19, True)
    assert ofs[0] == 0 and tuple(coef[0]) == (2048, 0)
    assert ofs[18] == 11 and tuple(coef[18]) == (2048, 0) and xmax == 18
    # y axis: fy is NOT reset on clamp
    yofs, ycoef, _ = oracle.resize_axis_table(12, 19, False)
    assert yofs[0] == -1 and ycoef[0][1] > 0
    src = np.zeros((12, 12), np.uint8)
    src[:, 0] = 200
    src[:, 11] = 100
    out = oracle.resize_linear(src)
    assert out[5, 0] == 200 and out[5, 18] == 100


@pytest.mark.parametrize("L", list(range(2, 73)) + [100])
def test_resize_stays_within_one_level_of_torch_s_bilinear(oracle, L):
    """An implementation nobody here wrote: torch's bilinear interpolation (pixel centres at half-integers, edges clamped, no
    antialiasing -- the convention cv::resize INTER_LINEAR is restated with), in float64.  The oracle's 11-bit fixed point with its
    two roundings must stay within one grey level of it for every ROI side the path can meet, enlarging and reducing."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(L)
    ramp = (np.add.outer(np.arange(L), np.arange(L)) * 255 // max(2 * L - 2, 1)).astype(np.uint8)
    for src in (rng.integers(0, 256, (L, L), dtype=np.uint8), ramp, (rng.integers(0, 2, (L, L)) * 255).astype(np.uint8)):
        ref = F.interpolate(torch.from_numpy(src.astype(np.float64))[None, None], size=(19, 19), mode="bilinear", align_corners=False,
                            antialias=False)[0, 0].numpy()
        assert np.abs(oracle.resize_linear(src).astype(np.float64) - ref).max() < 1.0


# ------------------------------------------------------------------ absdiff / integral
def test_absdiff_and_integral_match_numpy(oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    b = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    d = oracle.absdiff(a, b)
    assert np.array_equal(d, np.abs(a.astype(np.int16) - b.astype(np.int16)).astype(np.uint8))
    I = oracle.integral(d)
    want = np.zeros((38, 54), np.int64)
    want[1:, 1:] = d.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(I, want)


# ------------------------------------------------------------------ KAT 7-8 + FREAK internals
def test_freak_pattern_sizes_and_scale_index(oracle):
    f = oracle.Freak()
    ps = f.pattern_sizes()
    assert ps[:13] == [23, 24, 25, 27, 28, 29, 30, 31, 33, 34, 35, 37, 38] and ps[63] == 339
    for size, idx in [(7, 0), (8.4, 4), (12, 12), (18, 22), (27, 31), (36, 38), (40.5, 41), (107, 63), (3.0, 0), (1e6, 63)]:
        assert f.scale_index(size) == idx


def test_freak_def_pairs_decode(oracle):
    # allPairs index i(i-1)/2 + j
    for idx, (i, j) in zip(DEF_PAIRS_HEAD, DEF_PAIRS_HEAD_IJ):
        assert i * (i - 1) // 2 + j == idx


def test_kat7_constant_image_separates_bit_modes(oracle):
    img = np.full((200, 200), 90, np.uint8)
    kp = np.float32([[100, 100, 12.0]])
    for mode, want in [(oracle.BITS_SSE, 0xFF), (oracle.BITS_NATURAL, 0x00), (oracle.BITS_SSE_SIGNED, 0x00)]:
        valid, desc, theta, dirs = oracle.Freak(bit_mode=mode).compute(img, kp)
        assert valid[0] == 1 and theta[0] == 0 and tuple(dirs[0]) == (0, 0)
        assert (desc[0] == want).all()


def test_kat8_border_filter_is_inclusive(oracle):
    # erased iff x <= ps || y <= ps || x >= cols - ps || y >= rows - ps; ps = 38 for size 12
    img = np.zeros((240, 320), np.uint8)
    f = oracle.Freak()
    kp = np.float32([[38, 120, 12], [39, 120, 12], [38.5, 120, 12], [160, 38, 12], [160, 39, 12],
                     [320 - 38, 120, 12], [320 - 39, 120, 12], [160, 240 - 38, 12], [160, 240 - 39, 12],
                     [160, 120, 0.0], [160, 120, 1e-8], [160, 120, 2e-7], [-5, 10, 12], [160, 120, 200.0]])
    valid = f.compute(img, kp)[0]
    assert valid.tolist() == [0, 1, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0]


def test_freak_matches_python_restatement(oracle):
    """Full FREAK chain on a few keypoints, restated independently in Python (mode S and N)."""
    fr = synth.synth_stack(6, 200, 160)
    diff = oracle.absdiff(fr[5], fr[0])
    integ = oracle.integral(diff)
    kps = np.float32([[100, 80, 12.0], [61.5, 70.25, 8.4], [120, 90, 18.0]])
    f = oracle.Freak()
    valid, desc, theta, dirs = f.compute(diff, kps)
    assert valid.all()
    # orientation pairs and weights from scale 0 / rot 0
    P0 = py_pattern(0, 0)
    orient = [(0, 3), (1, 4), (2, 5), (0, 2), (1, 3), (2, 4), (3, 5), (4, 0), (5, 1)]
    ij = [(6 * r + a, 6 * r + b) for r in range(4) for (a, b) in orient]
    ij += [(b + k, b + k + 3) for b in (24, 30, 36) for k in range(3)]
    assert len(ij) == 45
    all_pairs = [(i, j) for i in range(1, 43) for j in range(i)]
    for k, (kx, ky, size) in enumerate(kps):
        idx = f.scale_index(size)
        v0 = [py_mean_intensity(integ, kx, ky, p) for p in py_pattern(idx, 0)]
        d0 = d1 = 0
        for (i, j) in ij:
            dx = np.float32(P0[i][0] - P0[j][0])
            dy = np.float32(P0[i][1] - P0[j][1])
            nsq = np.float32(np.float32(dx * dx) + np.float32(dy * dy))
            wdx = int(np.float64(np.float32(dx / nsq)) * 4096.0 + 0.5)
            wdy = int(np.float64(np.float32(dy / nsq)) * 4096.0 + 0.5)
            delta = v0[i] - v0[j]
            d0 += int(delta * wdx / 2048)  # int() truncates toward zero like C's integer division
            d1 += int(delta * wdy / 2048)
        assert (d0, d1) == tuple(dirs[k])
        a = np.float32(math.atan2(float(np.float32(d1)), float(np.float32(d0))))
        angle = np.float32(np.float64(a) * (180.0 / 3.1415926535897932384626433832795))
        t = int(np.float64(np.float32(256) * angle) * (1 / 360.0) + 0.5)
        t = t + 256 if t < 0 else t
        t = t - 256 if t >= 256 else t
        assert t == theta[k]
        v = [py_mean_intensity(integ, kx, ky, p) for p in py_pattern(idx, t)]
        # mode S: byte B (first 16), bit s <- pair 16 s + (15 - B)
        dp = _def_pairs(oracle)
        for B in range(16):
            byte = 0
            for s in range(8):
                i, j = all_pairs[dp[16 * s + (15 - B)]]
                byte |= int(v[i] >= v[j]) << s
            assert byte == desc[k, B]
        # mode N
        descN = oracle.Freak(bit_mode=oracle.BITS_NATURAL).compute(diff, kps[k:k + 1])[1]
        for B in range(16):
            byte = 0
            for s in range(8):
                i, j = all_pairs[dp[8 * B + s]]
                byte |= int(v[i] > v[j]) << s
            assert byte == descN[0, B]


def _def_pairs(oracle):
    """FREAK_DEF_PAIRS as compiled into the oracle (read back through its pair table)."""
    ij = oracle.Freak().description_pairs()
    out = [int(i) * (int(i) - 1) // 2 + int(j) for i, j in ij]
    assert out[:16] == DEF_PAIRS_HEAD and len(set(out)) == 512 and max(out) < 903
    return out


def test_theta_index_basics(oracle):
    assert oracle.theta_index(0, 0) == 0
    assert oracle.theta_index(10, 0) == 0
    assert oracle.theta_index(0, 10) == 64
    assert oracle.theta_index(-10, 0) == 128      # +180 degrees -> 128
    # truncation toward zero is asymmetric for negative angles: -90 deg -> 256*(-90)/360 + 0.5 = -63.5 -> -63 -> 193
    assert oracle.theta_index(0, -10) == 193


def test_theta_atan2f_disagreement_rate(oracle):
    """H2: how often glibc atan2f and (float)atan2(double) give a different bin (informational bound)."""
    rng = np.random.default_rng(2)
    d = rng.integers(-6000, 6001, (200000, 2))
    bad = sum(oracle.theta_index(int(a), int(b)) != oracle.theta_index_atan2f(int(a), int(b)) for a, b in d[:20000])
    assert bad <= 5


# ------------------------------------------------------------------ KAT 9-10: rows
def test_kat9_row_text(oracle):
    rows = np.zeros(2, oracle.ROW_DTYPE)
    rows[0] = (48, 48, 4, 12, list(range(10, 18)), list(range(200, 208)))
    rows[1] = (123.4567, 0.5, 17, 14.4, [0] * 8, [255] * 8)
    txt = oracle.format_rows(rows)
    assert txt == (b"48 48 4 12 0 0 10 11 12 13 14 15 16 17 200 201 202 203 204 205 206 207 \n"
                   b"123.457 0.5 17 14.4 0 0 0 0 0 0 0 0 0 0 255 255 255 255 255 255 255 255 \n")


def test_kat10_frame_labelling_and_pairing(oracle):
    # 7-frame clip -> processed frames 5 and 6 labelled 4 and 5; prev = frame 0 and 1 (:391-401, :485-488)
    fr = synth.synth_stack(7, 160, 120)
    kp = synth.dense_grid(160, 120, 16, 7.0, 23)
    offs = np.array([0, len(kp), 2 * len(kp)], np.int64)
    rows = oracle.Freak().extract_stream(fr, np.concatenate([kp, kp]), offs)
    assert sorted(set(rows["frame_number"].tolist())) == [4, 5]
    n = len(kp)
    assert len(rows) == 2 * n
    d0, v0 = oracle.Freak().extract_pair(fr[5], fr[0], kp)
    d1, v1 = oracle.Freak().extract_pair(fr[6], fr[1], kp)
    assert v0.all() and v1.all()
    assert np.array_equal(np.concatenate([rows["appearance"], rows["motion"]], 1), np.concatenate([d0, d1]))
    assert np.array_equal(rows["x"][:n], kp[:, 0]) and np.array_equal(rows["scale"][n:], kp[:, 2])
    # fewer than gap+1 frames: nothing
    assert len(oracle.Freak().extract_stream(fr[:5], kp[:0], np.zeros(1, np.int64))) == 0


def test_extract_pair_feeds_diff_to_freak_and_gray_to_mip(oracle):
    """:428 FREAK runs on the difference image, :460 MIP on (current, previous) gray frames."""
    fr = synth.synth_stack(6, 200, 160)
    kp = np.float32([[100, 80, 12.0], [90.7, 70.2, 12.0]])
    desc, valid = oracle.Freak().extract_pair(fr[5], fr[0], kp)
    assert valid.all()
    d64 = oracle.Freak().compute(oracle.absdiff(fr[5], fr[0]), kp)[1]
    assert np.array_equal(desc[:, :8], d64[:, :8])
    for k in range(2):
        rc, mot = oracle.mip_descriptor(fr[5], fr[0], 12.0, int(kp[k, 0]), int(kp[k, 1]))
        assert rc == 0 and np.array_equal(desc[k, 8:], mot)
        # and the MIP itself = resize of the 12x12 ROI at (x - 6, y - 6) + 8 centre codes
        x, y = int(kp[k, 0]), int(kp[k, 1])
        c19 = py_resize(fr[5][y - 6:y + 6, x - 6:x + 6])
        p19 = py_resize(fr[0][y - 6:y + 6, x - 6:x + 6])
        assert mot.tolist() == [py_mip(c19, p19, cx, cy) for (cx, cy) in CENTERS]


def test_bgr2gray_known_answers(oracle):
    # OpenCV 2.4.x RGB2Gray<uchar>: (1868 B + 9617 G + 4899 R + 8192) >> 14; the weights sum to 2^14
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert oracle.bgr2gray(px)[0].tolist() == [255, 0, 29, 150, 76, (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14]
    assert 1868 + 9617 + 4899 == 1 << 14



def test_bgr2gray_stays_within_one_level_of_bt601_and_pillow(oracle):
    """cv::cvtColor(BGR2GRAY) is restated from recalled fixed-point constants; two things nobody here wrote agree with it: the
    BT.601 luma in floating point (to the rounding) and Pillow's convert("L") (its own fixed point: equal on all but a few pixels
    in a thousand, never more than one level apart)."""
    from PIL import Image
    rng = np.random.default_rng(44)
    bgr = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    g = oracle.bgr2gray(bgr).astype(np.float64)
    luma = 0.114 * bgr[..., 0] + 0.587 * bgr[..., 1] + 0.299 * bgr[..., 2]
    assert np.abs(g - luma).max() <= 0.51
    pil = np.asarray(Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]), "RGB").convert("L")).astype(np.float64)
    assert np.abs(g - pil).max() <= 1.0 and (g != pil).mean() < 0.01
