// Device-side building blocks shared by the gfx950 kernels (kernels.hip, tile_kernel.hip).
// Compile with -ffp-contract=off: the float/double expressions restate reference expressions whose rounding
// is part of the result.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "device_types.h"

namespace mofreak {
namespace {

constexpr double kCvPi = 3.1415926535897932384626433832795;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS written by some lanes of a wave and read by others of the SAME wave: DS operations of one wave execute
// in order, so only the compiler has to be told not to move them across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ int absdiff_u8(uint32_t a, uint32_t b, int k)
{
    const int x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff;
    return x > y ? x - y : y - x;
}

// ------------------------------------------------------------------------------------------------
// FREAK pieces
// ------------------------------------------------------------------------------------------------

// floor(v / a) for 0 <= v, 0 < a, v / a <= 255 (a box SUM over its pixel count): one reciprocal, one fix-up.
__device__ __forceinline__ int div_box(int v, int a)
{
    int q = (int)((float)v * __builtin_amdgcn_rcpf((float)a));
    int r = v - q * a;  // (a can exceed 24 bits for the largest patterns of the gather path: full multiply)
    if (r < 0) {
        --q;
        r += a;
    }
    if (r >= a) ++q;
    return q;
}

// FREAK::meanIntensity, box branch (radius >= 0.5; the context refuses tables with a smaller sigma).
__device__ __forceinline__ int mean_intensity(const int32_t *__restrict__ integ, int pitch, float kx, float ky,
                                              const PatternPoint P)
{
    const float xf = P.x + kx;
    const float yf = P.y + ky;
    const float radius = P.sigma;
    const int x_left = (int)((double)(xf - radius) + 0.5);
    const int y_top = (int)((double)(yf - radius) + 0.5);
    const int x_right = (int)((double)(xf + radius) + 1.5);
    const int y_bottom = (int)((double)(yf + radius) + 1.5);
    const int32_t *top = integ + (int64_t)y_top * pitch + kIntegralColOffset;
    const int32_t *bot = integ + (int64_t)y_bottom * pitch + kIntegralColOffset;
    int ret_val = bot[x_right];
    ret_val -= bot[x_left];
    ret_val += top[x_left];
    ret_val -= top[x_right];
    return div_box(ret_val, (x_right - x_left) * (y_bottom - y_top)) & 0xff;
}

// thetaIdx from the integer direction sums (freak.cpp computeImpl):
//   angle = (float)(atan2((float)direction1,(float)direction0)*(180.0/CV_PI));  atan2(float,float) -> float
//   thetaIdx = int(FREAK_NB_ORIENTATION*angle*(1/360.0)+0.5); wrap into [0,256)
// The float atan2 is taken as the double result rounded once (the reference's x86 MSVC CRT does exactly that).
//
// No atan2 runs on the device.  thetaIdx is a monotone step function of the true angle of (direction0, direction1);
// the host finds every step's exact position through the float chain above (tables.cpp, theta_bounds) and stores
// its direction (cos, sin) in double.  Which side of a step an INTEGER direction lies on is the sign of a cross
// product, evaluated in double: the product's rounding error (~1e-12 at these magnitudes) is orders of magnitude
// below the smallest cross product an integer direction can have with a generic angle, so the result is exact and
// independent of any libm.
//   tb[0..127]   upper half plane: idx counts the steps with angle >= beta_k, beta_k ~ (k - 0.5) * 360/256 degrees
//   tb[128..254] lower half plane: idx = 256 - (steps with |angle| > mu_m), mu_m ~ (m + 0.5) * 360/256 degrees
// Which steps to test comes from a cheap single-precision angle (a degree-11 odd polynomial for atan on [0, 1], good
// to ~1e-5 rad, i.e. 4e-4 of a step): the count it suggests is within one of the true count, so testing the two steps
// around it decides exactly -- two independent table reads instead of a seven-deep chain of dependent ones.
__device__ __forceinline__ int theta_index(const ThetaBound *__restrict__ tb, int direction0, int direction1)
{
    if ((direction0 | direction1) == 0) return 0;  // atan2(0, 0) = 0
    const bool upper = direction1 >= 0;
    const float ax = fabsf((float)direction0), ay = fabsf((float)direction1);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float q = mn * __builtin_amdgcn_rcpf(mx), s = q * q;
    float r = -0.01172120f;
    r = __builtin_fmaf(r, s, 0.05265332f);
    r = __builtin_fmaf(r, s, -0.11643287f);
    r = __builtin_fmaf(r, s, 0.19354346f);
    r = __builtin_fmaf(r, s, -0.33262347f);
    r = __builtin_fmaf(r, s, 0.99997726f);
    r = r * q;                                   // atan(q), q in [0, 1]
    if (ay > ax) r = 1.57079632679f - r;
    if (direction0 < 0) r = 3.14159265359f - r;  // angle of (direction0, |direction1|), in [0, pi]
    const float pos = r * 40.7436654315f;         // in steps of 360/256 degrees
    const int n = upper ? 128 : 127;
    const ThetaBound *t = upper ? tb : tb + 128;
    const int c0 = min(max((int)(upper ? pos + 0.5f : pos), 1), n - 1);  // the suggested count: within one of the true one
    // upper half plane: step passed <=> cross = y * c - x * s >= 0 with y = direction1.  Lower half plane, y = -direction1:
    // passed <=> cross > 0; with both components negated instead the same expression is -cross, and "-cross >= 0" is
    // "not passed" (negation commutes with the roundings, and -0 >= 0 holds like 0 <= 0).
    const double y = (double)direction1, x = (double)(upper ? direction0 : -direction0);
    const ThetaBound b0 = t[c0 - 1], b1 = t[c0];   // steps number c0 and c0 + 1
    const int ge = ((y * b0.c - x * b0.s >= 0.0) ? 1 : 0) + ((y * b1.c - x * b1.s >= 0.0) ? 1 : 0);
    const int count = upper ? c0 - 1 + ge : c0 + 1 - ge;  // steps are ordered: passing step c0 + 1 implies passing step c0
    return upper ? count : (count == 0 ? 0 : kNbOrientation - count);
}

// ------------------------------------------------------------------------------------------------
// MIP pieces
// ------------------------------------------------------------------------------------------------

// motionInterchangePattern (MoFREAKUtilities.cpp:46-99) for all 8 patch centres (:308-316) at once:
// lane = 8*centre + offset computes one 9-byte strip SSD; the ballot IS the 8 motion bytes.
// The strip is 9 CONTIGUOUS bytes of the 19-byte-stride buffer (the reference walks patch.data with p++).
__device__ __forceinline__ uint64_t mip_bits(const uint8_t *cur19, const uint8_t *prev19, int mip_theta)
{
    const int lane = lane_id();
    const int c = lane >> 3, i = lane & 7;
    const int cx = (0xDDD99555u >> (4 * c)) & 15;         // centres x: 5,5,5,9,9,13,13,13
    const int cy = (0xD95D5D95u >> (4 * c)) & 15;         //         y: 5,9,13,5,13,5,9,13
    const int dx = (int)((0x14787410u >> (4 * i)) & 15) - 4;  // offsets dx: -4,-3,0,3,4,3,0,-3
    const int dy = (int)((0x10147874u >> (4 * i)) & 15) - 4;  //         dy: 0,3,4,3,0,-3,-4,-3
    const uint8_t *p = cur19 + (cy - 1) * kPatch + (cx - 1);
    const uint8_t *p2 = prev19 + (cy + dy - 1) * kPatch + (cx + dx - 1);
    int ssd = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int d = (int)p[k] - (int)p2[k];
        ssd += d * d;
    }
    return __ballot(ssd > mip_theta);
}

// One output pixel of cv::resize(8UC1 -> 19x19, INTER_LINEAR) from its four source pixels.
__device__ __forceinline__ uint8_t resize_px(int s00, int s01, int s10, int s11, const ResizeTap tx, const ResizeTap ty)
{
    const int t0 = s00 * tx.c0 + s01 * tx.c1;
    const int t1 = s10 * tx.c0 + s11 * tx.c1;
    return (uint8_t)(((((int)ty.c0 * (t0 >> 4)) >> 16) + (((int)ty.c1 * (t1 >> 4)) >> 16) + 2) >> 2);
}


}  // namespace
}  // namespace mofreak
