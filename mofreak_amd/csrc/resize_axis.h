// cv::resize(ROI side L -> 19, INTER_LINEAR, 8UC1) coefficient rows as COMPILE-TIME constants (OpenCV 2.4.x
// imgproc/src/imgwarp.cpp, called at MoFREAKUtilities.cpp:303-304).
//
// tables.cpp builds the same rows at run time for every ROI side (build_resize_axis, with std::floor / lrintf) and
// uploads them; the lane-per-keypoint MIP of the tile kernel (mip_lane.h) is compiled per ROI side and needs them as
// constants: source offsets become register numbers and byte selectors, weights become scalar operands.  The two
// constructions are compared entry by entry when the host tables are built (tables.cpp: a difference throws), so a
// compiler that evaluated this file differently from the run-time code could not go unnoticed.
#pragma once

#include <cstdint>

namespace mofreak {

constexpr int kAxisOut = 19;  // kPatch (tables.h)

struct ResizeAxisC {
    int ofs[kAxisOut];   // first source index (x axis: clamped as cv::resize does; y axis: clipped into the ROI)
    int ofs1[kAxisOut];  // second source index (x: ofs + 1, or ofs from xmax on; y: clip(ofs + 1))
    int c0[kAxisOut];    // 11-bit fixed-point weights; x axis from xmax on: {2048, 0}
    int c1[kAxisOut];
};

constexpr int axis_floor(double v)
{
    const int i = static_cast<int>(v);  // truncation
    return static_cast<double>(i) > v ? i - 1 : i;
}

// cvRound (round half to even) of a small non-negative float whose fraction is exactly representable
constexpr int axis_round_half_even(float v)
{
    const int i = static_cast<int>(v);
    const float r = v - static_cast<float>(i);
    if (r > 0.5f) return i + 1;
    if (r < 0.5f) return i;
    return (i & 1) ? i + 1 : i;
}

constexpr int axis_clip(int v, int n) { return v < 0 ? 0 : (v < n ? v : n - 1); }

constexpr ResizeAxisC make_resize_axis(int L, bool is_x)
{
    ResizeAxisC a{};
    const double inv_scale = static_cast<double>(kAxisOut) / L;
    const double scale = 1. / inv_scale;
    int dmax = kAxisOut;
    int s_arr[kAxisOut] = {};
    float f_arr[kAxisOut] = {};
    for (int d = 0; d < kAxisOut; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = axis_floor(static_cast<double>(f));
        f -= static_cast<float>(s);
        if (is_x) {
            if (s < 0) {
                f = 0;
                s = 0;
            }
            if (s + 1 >= L) {
                dmax = dmax < d ? dmax : d;
                if (s >= L - 1) {
                    f = 0;
                    s = L - 1;
                }
            }
        }
        s_arr[d] = s;
        f_arr[d] = f;
    }
    for (int d = 0; d < kAxisOut; ++d) {
        const int c0 = axis_round_half_even((1.f - f_arr[d]) * 2048);
        const int c1 = axis_round_half_even(f_arr[d] * 2048);
        if (is_x) {
            if (d < dmax) {
                a.ofs[d] = s_arr[d];
                a.ofs1[d] = s_arr[d] + 1;
                a.c0[d] = c0;
                a.c1[d] = c1;
            } else {  // HResizeLinear's tail: D[dx] = S[sx] * ONE
                a.ofs[d] = a.ofs1[d] = s_arr[d];
                a.c0[d] = 2048;
                a.c1[d] = 0;
            }
        } else {
            a.ofs[d] = axis_clip(s_arr[d], L);
            a.ofs1[d] = axis_clip(s_arr[d] + 1, L);
            a.c0[d] = c0;
            a.c1[d] = c1;
        }
    }
    return a;
}

}  // namespace mofreak
