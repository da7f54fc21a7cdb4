"""An independent, sequential, pure-Python restatement of the keypoint search of the reference's vendored BRISK
detector -- BriskScaleSpace::getKeypoints with isMax2D, refine3D, getScoreMaxAbove / getScoreMaxBelow, subpixel2D and
the three refine1D variants (src/MoFREAK/brisk.cpp:590-704, 838-934, 937-1104, 1106-1416, 1418-1644) -- on top of
BriskLayer's lazily filled score cache (:1685-1745).

TEST INFRASTRUCTURE ONLY, and it pins nothing: the reference has no detector tests and does not build here, so this
file is one more reading of the same source.  What it is for: oracle/brisk_oracle.c and the device kernels were
written side by side, so a misreading they share passes every parity test.  This restatement was written from
brisk.cpp alone, in the reference's own control flow -- one candidate at a time, scores asked for one at a time,
every answer stored in a dictionary the way `uint8_t& score = *(scores_.data + ...)` stores it -- and shares no code
or data structure with the oracle: the pyramid comes from the literal SSE emulations of tests/test_brisk_oracle.py,
the corner scores from the max-over-arcs-of-min form.  tests/test_brisk_sequential.py compares the two on tie-heavy
inputs, where the outcome depends on which cache cells are filled when.

Arithmetic: C `float` is numpy.float32.  Two readings of "a float expression", selectable with set_fp_model():
  "sse"  (default; what the oracle and the device implement) every float operation rounds to float, an expression with
         a double literal in it (`/6.0`, `*3.0`, `0.75`) is evaluated in double; results are rounded to float where the
         reference assigns them to a float;
  "x87"  what a 32-bit MSVC 2010 build without /arch:SSE2 does (the reference's project file names no toolset or /arch):
         every intermediate of an expression is kept at the FPU's 53-bit precision and rounded to float only where it is
         assigned to a float variable, passed as a float argument or returned as float.
tests/test_brisk_sequential.py counts how many keypoints differ between the two -- the size of that open parity risk.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32   # a value STORED in a C float: rounds in both models
_X87 = [False]


def set_fp_model(name: str) -> None:
    assert name in ("sse", "x87")
    _X87[0] = name == "x87"


# ONE float-typed operation inside an expression (operands are floats, or ints converted to float): rounded to float
# under SSE, carried at the FPU's 53 bits under x87.  f32() marks the places where the reference stores into a float.
def fmul(a, b):
    return float(a) * float(b) if _X87[0] else np.float32(np.float32(a) * np.float32(b))


def fadd(a, b):
    return float(a) + float(b) if _X87[0] else np.float32(np.float32(a) + np.float32(b))


def fsub(a, b):
    return float(a) - float(b) if _X87[0] else np.float32(np.float32(a) - np.float32(b))


def fdiv(a, b):
    return float(a) / float(b) if _X87[0] else np.float32(np.float32(a) / np.float32(b))


C16 = [(-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3),
       (-1, 3), (-2, 2), (-3, 1)]
C8 = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]


def dense_scores(img, circle, arc):
    """The largest b for which `arc` contiguous circle pixels are all > c + b or all < c - b (0 if none): what
    cornerScore's bisection converges to (tests/test_brisk_oracle.py checks that form against the decision trees)."""
    n = len(circle)
    r = max(abs(d) for p in circle for d in p)
    h, w = img.shape
    out = np.zeros((h, w), np.int32)
    if h <= 2 * r or w <= 2 * r:
        return out
    I = img.astype(np.int32)
    c = I[r:h - r, r:w - r]
    ring = np.stack([I[r + dy:h - r + dy, r + dx:w - r + dx] - c for dx, dy in circle])
    best = np.zeros_like(c)
    for s in range(n):
        idx = [(s + k) % n for k in range(arc)]
        best = np.maximum(best, np.maximum(ring[idx].min(0), (-ring[idx]).min(0)))
    out[r:h - r, r:w - r] = np.maximum(best - 1, 0)
    return out


def c_int(v) -> int:
    """C conversion of a floating value to int: truncation toward zero."""
    return int(v)


class Layer:
    """BriskLayer (:1647-1745): image, scale/offset, and the score cache as the reference fills it."""

    def __init__(self, img, scale, offset):
        self.img = img
        self.rows, self.cols = img.shape
        self.scale, self.offset = f32(scale), f32(offset)
        self.S = dense_scores(img, C16, 9)     # what cornerScore returns for a floor of 0
        self.S58 = dense_scores(img, C8, 5)
        self.cache = {}                        # (x, y) -> score: scores_ (zeros where never written)

    def raw(self, x, y) -> int:
        """scores_.data[y * cols + x]: what isMax2D reads (:840-843)."""
        return self.cache.get((x, y), 0)

    def agast_points(self, threshold):
        """getAgastPoints (:1676-1689): OAST 9/16 corners in raster order, their scores written to the cache."""
        ys, xs = np.nonzero(self.S[3:self.rows - 3, 3:self.cols - 3] >= threshold) if self.rows > 6 and self.cols > 6 else ([], [])
        pts = [(int(x) + 3, int(y) + 3) for x, y in zip(xs, ys)]
        for x, y in pts:
            self.cache[(x, y)] = int(self.S[y, x])  # cornerScore with the detector's threshold as the floor: S >= threshold
        return pts

    def score(self, x: int, y: int, threshold: int = 1) -> int:
        """getAgastScore(int, int, threshold) (:1690-1699): cached values above 2 come back as they are, anything else is
        computed (floor threshold - 1), zeroed below the threshold and WRITTEN to the cache (score is a reference)."""
        if x < 3 or y < 3:
            return 0
        if x >= self.cols - 3 or y >= self.rows - 3:
            return 0
        s = self.cache.get((x, y), 0)
        if s > 2:
            return s
        s = max(threshold - 1, int(self.S[y, x]))
        if s < threshold:
            s = 0
        self.cache[(x, y)] = s
        return s

    def score58(self, x: int, y: int, threshold: int = 1) -> int:
        """getAgastScore_5_8 (:1701-1708): never cached."""
        if x < 2 or y < 2:
            return 0
        if x >= self.cols - 2 or y >= self.rows - 2:
            return 0
        s = max(threshold - 1, int(self.S58[y, x]))
        return 0 if s < threshold else s

    def score_f(self, xf, yf, threshold: int = 1) -> int:
        """getAgastScore(float, float, threshold, scale = 1) (:1710-1724): bilinear interpolation of four cached scores,
        in float, converted to uint8_t."""
        xf, yf = f32(xf), f32(yf)
        x = c_int(xf)
        rx1 = f32(xf - f32(x))
        rx = f32(f32(1.0) - rx1)
        y = c_int(yf)
        ry1 = f32(yf - f32(y))
        ry = f32(f32(1.0) - ry1)
        v = fmul(fmul(rx, ry), self.score(x, y, threshold))          # one expression: converted to uint8_t at the return
        v = fadd(v, fmul(fmul(rx1, ry), self.score(x + 1, y, threshold)))
        v = fadd(v, fmul(fmul(rx, ry1), self.score(x, y + 1, threshold)))
        v = fadd(v, fmul(fmul(rx1, ry1), self.score(x + 1, y + 1, threshold)))
        return c_int(v) & 0xFF


def subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2):
    """subpixel2D (:1535-1644) -> (max, delta_x, delta_y); the `delta_y = delta_x1` slip at :1629/:1634 is kept."""
    tmp1 = s_0_0 + s_0_2 - 2 * s_1_1 + s_2_0 + s_2_2
    coeff1 = 3 * (tmp1 + s_0_1 - ((s_1_0 + s_1_2) << 1) + s_2_1)
    coeff2 = 3 * (tmp1 - ((s_0_1 + s_2_1) << 1) + s_1_0 + s_1_2)
    tmp2 = s_0_2 - s_2_0
    tmp3 = s_0_0 + tmp2 - s_2_2
    tmp4 = tmp3 - 2 * tmp2
    coeff3 = -3 * (tmp3 + s_0_1 - s_2_1)
    coeff4 = -3 * (tmp4 + s_1_0 - s_1_2)
    coeff5 = (s_0_0 - s_0_2 - s_2_0 + s_2_2) << 2
    coeff6 = -(s_0_0 + s_0_2 - ((s_1_0 + s_0_1 + s_1_2 + s_2_1) << 1) - 5 * s_1_1 + s_2_0 + s_2_2) << 1
    H_det = 4 * coeff1 * coeff2 - coeff5 * coeff5
    if H_det == 0:
        return f32(float(f32(coeff6)) / 18.0), f32(0.0), f32(0.0)
    if not (H_det > 0 and coeff1 < 0):
        tmp_max = coeff3 + coeff4 + coeff5
        dx, dy = 1.0, 1.0
        tmp = -coeff3 + coeff4 - coeff5
        if tmp > tmp_max:
            tmp_max, dx, dy = tmp, -1.0, 1.0
        tmp = coeff3 - coeff4 - coeff5
        if tmp > tmp_max:
            tmp_max, dx, dy = tmp, 1.0, -1.0
        tmp = -coeff3 - coeff4 + coeff5
        if tmp > tmp_max:
            tmp_max, dx, dy = tmp, -1.0, -1.0
        return f32(float(f32(tmp_max + coeff1 + coeff2 + coeff6)) / 18.0), f32(dx), f32(dy)
    delta_x = f32(fdiv(2 * coeff2 * coeff3 - coeff4 * coeff5, -H_det))
    delta_y = f32(fdiv(2 * coeff1 * coeff4 - coeff3 * coeff5, -H_det))
    tx = tx_ = ty = ty_ = False
    if delta_x > 1.0:
        tx = True
    elif delta_x < -1.0:
        tx_ = True
    if delta_y > 1.0:
        ty = True
    if delta_y < -1.0:
        ty_ = True

    def clamp1(v):
        if v > 1.0:
            return f32(1.0)
        if v < -1.0:
            return f32(-1.0)
        return v

    def quad(dx, dy):  # (c1*dx*dx + c2*dy*dy + c3*dx + c4*dy + c5*dx*dy + c6) / 18.0: float products and sums, double division
        v = fmul(fmul(coeff1, dx), dx)
        v = fadd(v, fmul(fmul(coeff2, dy), dy))
        v = fadd(v, fmul(coeff3, dx))
        v = fadd(v, fmul(coeff4, dy))
        v = fadd(v, fmul(fmul(coeff5, dx), dy))
        v = fadd(v, coeff6)
        return f32(float(v) / 18.0)

    if tx or tx_ or ty or ty_:
        dx1 = dx2 = dy1 = dy2 = f32(0.0)
        if tx:
            dx1 = f32(1.0)
            dy1 = clamp1(f32(fdiv(-(coeff4 + coeff5), 2 * coeff2)))
        elif tx_:
            dx1 = f32(-1.0)
            dy1 = clamp1(f32(fdiv(-(coeff4 - coeff5), 2 * coeff2)))
        if ty:
            dy2 = f32(1.0)
            dx2 = clamp1(f32(fdiv(-(coeff3 + coeff5), 2 * coeff1)))
        elif ty_:
            dy2 = f32(-1.0)
            dx2 = clamp1(f32(fdiv(-(coeff3 - coeff5), 2 * coeff1)))
        max1, max2 = quad(dx1, dy1), quad(dx2, dy2)
        if max1 > max2:
            return max1, dx1, dx1
        return max2, dx2, dx2
    return quad(delta_x, delta_y), delta_x, delta_y


def _refine1d(s_05, s0, s05, k, lo, hi, div):
    """The common shape of refine1D / refine1D_1 / refine1D_2 (:1418-1533): k = the nine integer coefficients."""
    s_05, s0, s05 = f32(s_05), f32(s0), f32(s05)
    i_05 = c_int(1024.0 * float(s_05) + 0.5)
    i0 = c_int(1024.0 * float(s0) + 0.5)
    i05 = c_int(1024.0 * float(s05) + 0.5)
    a = k[0] * i_05 + k[1] * i0 + k[2] * i05
    if a >= 0:
        if s0 >= s_05 and s0 >= s05:
            return f32(1.0), s0
        if s_05 >= s0 and s_05 >= s05:
            return f32(lo), s_05
        if s05 >= s0 and s05 >= s_05:
            return f32(hi), s05
    b = k[3] * i_05 + k[4] * i0 + k[5] * i05
    ret = f32(fdiv(-b, 2 * a))
    if float(ret) < lo:
        ret = f32(lo)
    elif float(ret) > hi:
        ret = f32(hi)
    c = k[6] * i_05 + k[7] * i0 + k[8] * i05
    mx = fadd(fadd(c, fmul(fmul(a, ret), ret)), fmul(b, ret))  # one expression, stored into `max`
    return ret, f32(float(f32(mx)) / div)


def refine1d(s_05, s0, s05):
    return _refine1d(s_05, s0, s05, (16, -24, 8, -40, 54, -14, 24, -27, 6), 0.75, 1.5, 3072.0)


def refine1d_1(s_05, s0, s05):
    return _refine1d(s_05, s0, s05, (9, -18, 9, -21, 36, -15, 12, -16, 6), 0.6666666666666666666666666667, 1.3333333333333333333333333333, 2048.0)


def refine1d_2(s_05, s0, s05):
    return _refine1d(s_05, s0, s05, (2, -4, 2, -5, 8, -3, 3, -3, 1), 0.7, 1.5, 1024.0)


class ScaleSpace:
    """BriskScaleSpace (:561-704)."""

    BASIC_SIZE = f32(12.0)  # basicSize_ (:59)

    def __init__(self, img, octaves, halfsample, twothirdsample):
        self.n_layers = 1 if octaves == 0 else 2 * octaves
        L = [Layer(np.ascontiguousarray(img, np.uint8), 1.0, 0.0)]

        def derive(src, fn, factor):
            scale = f32(src.scale * f32(factor))
            return Layer(fn(src.img), scale, f32(0.5 * float(scale) - 0.5))

        if self.n_layers > 1:
            L.append(derive(L[0], twothirdsample, 1.5))
        for i in range(2, self.n_layers, 2):
            L.append(derive(L[i - 2], halfsample, 2))
            L.append(derive(L[i - 1], halfsample, 2))
        self.L = L

    # ---- isMax2D (:838-934)
    def is_max_2d(self, layer, x, y):
        raw = self.L[layer].raw
        center = raw(x, y)
        s_10 = raw(x - 1, y)
        if center < s_10:
            return False
        s10 = raw(x + 1, y)
        if center < s10:
            return False
        s0_1 = raw(x, y - 1)
        if center < s0_1:
            return False
        s01 = raw(x, y + 1)
        if center < s01:
            return False
        s_11 = raw(x - 1, y + 1)
        if center < s_11:
            return False
        s11 = raw(x + 1, y + 1)
        if center < s11:
            return False
        s1_1 = raw(x + 1, y - 1)
        if center < s1_1:
            return False
        s_1_1 = raw(x - 1, y - 1)
        if center < s_1_1:
            return False
        delta = []
        for v, d in ((s_1_1, (-1, -1)), (s0_1, (0, -1)), (s1_1, (1, -1)), (s_10, (-1, 0)), (s10, (1, 0)), (s_11, (-1, 1)), (s01, (0, 1)),
                     (s11, (1, 1))):
            if center == v:
                delta.append(d)
        if delta:
            smoothed = 4 * center + 2 * (s_10 + s10 + s0_1 + s01) + s_1_1 + s1_1 + s_11 + s11
            for dx, dy in delta:
                cx, cy = x + dx, y + dy
                other = (raw(cx - 1, cy - 1) + 2 * raw(cx, cy - 1) + raw(cx + 1, cy - 1) + 2 * raw(cx + 1, cy) + 4 * raw(cx, cy) +
                         2 * raw(cx - 1, cy) + raw(cx - 1, cy + 1) + 2 * raw(cx, cy + 1) + raw(cx + 1, cy + 1))
                if other > smoothed:
                    return False
        return True

    # ---- the walks over the neighbouring layers (:1106-1416)
    def _walk(self, other, x_1, x1, y_1, y1, threshold, tie_rule):
        """First row, middle rows, bottom row of getScoreMaxAbove / getScoreMaxBelow -> (ok, max, max_x, max_y).
        tie_rule: the smoothed-sum comparison getScoreMaxBelow makes on equal scores inside a middle row (:1316-1339)."""
        S, Sf = other.score, other.score_f
        max_x = c_int(fadd(x_1, 1))
        max_y = c_int(fadd(y_1, 1))
        mx = f32(Sf(x_1, y_1))
        if mx > threshold:
            return False, mx, max_x, max_y
        for x in range(c_int(fadd(x_1, 1)), c_int(x1) + 1):
            t = f32(Sf(f32(x), y_1))
            if t > threshold:
                return False, mx, max_x, max_y
            if t > mx:
                mx, max_x = t, x
        t = f32(Sf(x1, y_1))
        if t > threshold:
            return False, mx, max_x, max_y
        if t > mx:
            mx, max_x = t, c_int(x1)
        for y in range(c_int(fadd(y_1, 1)), c_int(y1) + 1):
            t = f32(Sf(x_1, f32(y)))
            if t > threshold:
                return False, mx, max_x, max_y
            if t > mx:
                mx, max_x, max_y = t, c_int(fadd(x_1, 1)), y
            for x in range(c_int(fadd(x_1, 1)), c_int(x1) + 1):
                t = f32(S(x, y))
                if t > threshold:
                    return False, mx, max_x, max_y
                if tie_rule and t == mx:
                    t1 = 2 * (S(x - 1, y) + S(x + 1, y) + S(x, y + 1) + S(x, y - 1)) + (S(x + 1, y + 1) + S(x - 1, y + 1) + S(x + 1, y - 1) + S(x - 1, y - 1))
                    t2 = 2 * (S(max_x - 1, max_y) + S(max_x + 1, max_y) + S(max_x, max_y + 1) + S(max_x, max_y - 1)) + (
                        S(max_x + 1, max_y + 1) + S(max_x - 1, max_y + 1) + S(max_x + 1, max_y - 1) + S(max_x - 1, max_y - 1))
                    if t1 > t2:
                        max_x, max_y = x, y
                if t > mx:
                    mx, max_x, max_y = t, x, y
            t = f32(Sf(x1, f32(y)))
            if t > threshold:
                return False, mx, max_x, max_y
            if t > mx:
                mx, max_x, max_y = t, c_int(x1), y
        t = f32(Sf(x_1, y1))
        if t > mx:
            mx, max_x, max_y = t, c_int(fadd(x_1, 1)), c_int(y1)
        for x in range(c_int(fadd(x_1, 1)), c_int(x1) + 1):
            t = f32(Sf(f32(x), y1))
            if t > mx:
                mx, max_x, max_y = t, x, c_int(y1)
        t = f32(Sf(x1, y1))
        if t > mx:
            mx, max_x, max_y = t, c_int(x1), c_int(y1)
        return True, mx, max_x, max_y

    @staticmethod
    def _patch(fn, x, y):
        """The nine scores in the order subpixel2D takes them: s_0_0, s_0_1, s_0_2, s_1_0, ... (first index x)."""
        return [fn(x + i, y + j) for i in (-1, 0, 1) for j in (-1, 0, 1)]

    @staticmethod
    def _saturate(dx, dy):
        refined = True
        if float(dx) > 1.0:
            dx, refined = f32(1.0), False
        if float(dx) < -1.0:
            dx, refined = f32(-1.0), False
        if float(dy) > 1.0:
            dy, refined = f32(1.0), False
        if float(dy) < -1.0:
            dy, refined = f32(-1.0), False
        return dx, dy, refined

    def score_max_above(self, layer, x_layer, y_layer, threshold):
        """getScoreMaxAbove (:1106-1240) -> (value, ismax, dx, dy)."""
        above = self.L[layer + 1]
        if layer % 2 == 0:
            q = lambda v: f32(float(f32(v)) / 6.0)
            x_1, x1 = q(4 * x_layer - 1 - 2), q(4 * x_layer - 1 + 2)
            y_1, y1 = q(4 * y_layer - 1 - 2), q(4 * y_layer - 1 + 2)
        else:
            q = lambda v: f32(f32(v) / f32(8.0))
            x_1, x1 = q(6 * x_layer - 1 - 3), q(6 * x_layer - 1 + 3)
            y_1, y1 = q(6 * y_layer - 1 - 3), q(6 * y_layer - 1 + 3)
        ok, mx, max_x, max_y = self._walk(above, x_1, x1, y_1, y1, threshold, tie_rule=False)
        if not ok:
            return f32(0), False, f32(0), f32(0)
        refined_max, dx_1, dy_1 = subpixel2d(*self._patch(above.score, max_x, max_y))
        real_x, real_y = f32(fadd(max_x, dx_1)), f32(fadd(max_y, dy_1))
        if layer % 2 == 0:
            dx = f32(fsub(fdiv(fadd(fmul(real_x, 6.0), 1.0), 4.0), x_layer))
            dy = f32(fsub(fdiv(fadd(fmul(real_y, 6.0), 1.0), 4.0), y_layer))
        else:
            dx = f32((float(real_x) * 8.0 + 1.0) / 6.0 - float(f32(x_layer)))
            dy = f32((float(real_y) * 8.0 + 1.0) / 6.0 - float(f32(y_layer)))
        dx, dy, refined = self._saturate(dx, dy)
        return (max(refined_max, mx) if refined else mx), True, dx, dy

    def score_max_below(self, layer, x_layer, y_layer, threshold):
        """getScoreMaxBelow (:1242-1416) -> (value, ismax, dx, dy)."""
        below = self.L[layer - 1]
        if layer % 2 == 0:
            q = lambda v: f32(float(f32(v)) / 6.0)
            x_1, x1 = q(8 * x_layer + 1 - 4), q(8 * x_layer + 1 + 4)
            y_1, y1 = q(8 * y_layer + 1 - 4), q(8 * y_layer + 1 + 4)
        else:
            q = lambda v: f32(float(f32(v)) / 4.0)
            x_1, x1 = q(6 * x_layer + 1 - 3), q(6 * x_layer + 1 + 3)
            y_1, y1 = q(6 * y_layer + 1 - 3), q(6 * y_layer + 1 + 3)
        ok, mx, max_x, max_y = self._walk(below, x_1, x1, y_1, y1, threshold, tie_rule=True)
        if not ok:
            return f32(0), False, f32(0), f32(0)
        refined_max, dx_1, dy_1 = subpixel2d(*self._patch(below.score, max_x, max_y))
        real_x, real_y = f32(fadd(max_x, dx_1)), f32(fadd(max_y, dy_1))
        if layer % 2 == 0:
            dx = f32((float(real_x) * 6.0 + 1.0) / 8.0 - float(f32(x_layer)))
            dy = f32((float(real_y) * 6.0 + 1.0) / 8.0 - float(f32(y_layer)))
        else:
            dx = f32((float(real_x) * 4.0 - 1.0) / 6.0 - float(f32(x_layer)))
            dy = f32((float(real_y) * 4.0 - 1.0) / 6.0 - float(f32(y_layer)))
        dx, dy, refined = self._saturate(dx, dy)
        return (max(refined_max, mx) if refined else mx), True, dx, dy

    # ---- refine3D (:937-1104) -> (score, x, y, scale, ismax)
    def refine3d(self, layer, x_layer, y_layer):
        this = self.L[layer]
        center = this.score(x_layer, y_layer, 1)
        max_above, ismax, dxa, dya = self.score_max_above(layer, x_layer, y_layer, center)
        if not ismax:
            return f32(0), None, None, None, False
        xl, yl = f32(x_layer), f32(y_layer)

        def place(r0, d_layer, r1, d_other, c):
            """(r0 * delta_layer + r1 * delta_other + float(c)) * scale + offset, in float."""
            v = fadd(fadd(fmul(r0, d_layer), fmul(r1, d_other)), c)
            return f32(fadd(fmul(v, this.scale), this.offset))

        if layer % 2 == 0:
            if layer == 0:
                p = self._patch(self.L[0].score58, x_layer, y_layer)
                _, dxb, dyb = subpixel2d(*p)
                # the running maximum is kept in a uchar and compared with ints: plain maximum of the nine
                max_below = f32(max(p))
            else:
                max_below, ismax, dxb, dyb = self.score_max_below(layer, x_layer, y_layer, center)
                if not ismax:
                    return f32(0), None, None, None, False
            max_layer, dxl, dyl = subpixel2d(*self._patch(this.score, x_layer, y_layer))
            s0 = max(f32(center), max_layer)
            scale, mx = (refine1d_2 if layer == 0 else refine1d)(max_below, s0, max_above)
            if float(scale) > 1.0:
                r0 = f32((1.5 - float(scale)) / .5)
                r1 = f32(1.0 - float(r0))
                x, y = place(r0, dxl, r1, dxa, xl), place(r0, dyl, r1, dya, yl)
            elif layer == 0:
                r0 = f32((float(scale) - 0.5) / 0.5)
                r_1 = f32(1.0 - float(r0))
                x = f32(fadd(fadd(fmul(r0, dxl), fmul(r_1, dxb)), xl))
                y = f32(fadd(fadd(fmul(r0, dyl), fmul(r_1, dyb)), yl))
            else:
                r0 = f32((float(scale) - 0.75) / 0.25)
                r_1 = f32(1.0 - float(r0))
                x, y = place(r0, dxl, r_1, dxb, xl), place(r0, dyl, r_1, dyb, yl)
        else:
            max_below, ismax, dxb, dyb = self.score_max_below(layer, x_layer, y_layer, center)
            if not ismax:
                return f32(0), None, None, None, False
            max_layer, dxl, dyl = subpixel2d(*self._patch(this.score, x_layer, y_layer))
            scale, mx = refine1d_1(max_below, max(f32(center), max_layer), max_above)
            if float(scale) > 1.0:
                r0 = f32(4.0 - float(scale) * 3.0)
                r1 = f32(1.0 - float(r0))
                x, y = place(r0, dxl, r1, dxa, xl), place(r0, dyl, r1, dya, yl)
            else:
                r0 = f32(float(scale) * 3.0 - 2.0)
                r_1 = f32(1.0 - float(r0))
                x, y = place(r0, dxl, r_1, dxb, xl), place(r0, dyl, r_1, dyb, yl)
        scale = f32(fmul(scale, this.scale))
        return mx, x, y, scale, True

    # ---- getKeypoints (:590-704)
    def get_keypoints(self, threshold):
        """-> list of (x, y, size, response, layer), in the order the reference appends them."""
        safe = threshold  # safetyFactor_ = 1
        points = [l.agast_points(safe) for l in self.L]
        out = []
        if self.n_layers == 1:
            l = self.L[0]
            for x, y in points[0]:
                if not self.is_max_2d(0, x, y):
                    continue
                mx, dx, dy = subpixel2d(*self._patch(l.score, x, y))
                out.append((f32(fadd(x, dx)), f32(fadd(y, dy)), self.BASIC_SIZE, mx, 0))
            return out
        for i, l in enumerate(self.L):
            if i == self.n_layers - 1:
                for x, y in points[i]:
                    if not self.is_max_2d(i, x, y):
                        continue
                    _, ismax, _, _ = self.score_max_below(i, x, y, l.score(x, y, safe))
                    if not ismax:
                        continue
                    mx, dx, dy = subpixel2d(*self._patch(l.score, x, y))
                    out.append((f32(fadd(fmul(fadd(x, dx), l.scale), l.offset)), f32(fadd(fmul(fadd(y, dy), l.scale), l.offset)),
                                f32(fmul(self.BASIC_SIZE, l.scale)), mx, i))
            else:
                for x, y in points[i]:
                    if not self.is_max_2d(i, x, y):
                        continue
                    score, kx, ky, scale, ismax = self.refine3d(i, x, y)
                    if not ismax:
                        continue
                    if score > f32(threshold):
                        out.append((kx, ky, f32(fmul(self.BASIC_SIZE, scale)), score, i))
        return out
