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

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

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

// (int)((double)a + 0.5) for a float 0.5 <= a < 2^22 with one float add: a + 0.5f is exact while it stays in a's
// binade; when it crosses into the next one the sum lies in [2^k, 2^k + 0.5), so rounding it to the coarser grid cannot
// reach another integer.  (Tile-path taps are > 1: the keypoint passed FREAK's border filter.)
__device__ __forceinline__ int round_half_up_pos(float a) { return (int)(a + 0.5f); }

// floor(v / a) for 0 <= v <= 255 * a, 0 < a <= 8192: (v + 0.5) / a lies at least 0.5 / a away from an integer, and the
// relative error of v_rcp_f32 (1 ulp) plus one rounding of the fma is below 2^-22, i.e. below 256 * 2^-22 = 6e-5 absolute.
__device__ __forceinline__ int div_box_small(int v, int a)
{
    const float r = __builtin_amdgcn_rcpf((float)a);
    return (int)__builtin_fmaf((float)v, r, 0.5f * r);
}

// The vertical step of the 8-bit bilinear resize, ((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2, with the
// weights pre-shifted by 12: t < 2^20 and b << 12 <= 2^23 are 24-bit operands, and the high half of their 48-bit
// product, (t & ~15) * (b << 12) >> 32, is (b * (t >> 4)) >> 16 exactly (all factors non-negative).
__device__ __forceinline__ int resize_y(uint32_t t0, uint32_t t1, uint32_t b0s, uint32_t b1s)
{
    uint32_t p0, p1;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(p0) : "v"(t0 & ~15u), "v"(b0s));
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(p1) : "v"(t1 & ~15u), "v"(b1s));
    return (int)((p0 + p1 + 2u) >> 2);
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

// FREAK::meanIntensity, box branch (radius >= 0.5; the context refuses tables with a smaller sigma), on the int32 integral
// of one pair in HBM (`integ` is the same in every lane: the corners are 32-bit element offsets from it).
// int(xf - radius + 0.5), int(xf + radius + 1.5): the reference adds 0.5 / 1.5 in double, i.e. exactly; the arguments are > 1
// (the keypoint passed FREAK's border filter), so round_half_up_pos gives the same integers.
__device__ __forceinline__ int mean_intensity(const int32_t *__restrict__ integ, int pitch, float kx, float ky,
                                              const PatternPoint P)
{
    const float xf = P.x + kx;
    const float yf = P.y + ky;
    const float radius = P.sigma;
    const int x_left = round_half_up_pos(xf - radius);
    const int y_top = round_half_up_pos(yf - radius);
    const int x_right = round_half_up_pos(xf + radius) + 1;
    const int y_bottom = round_half_up_pos(yf + radius) + 1;
    const uint32_t top = __umul24(y_top, pitch) + kIntegralColOffset, bot = __umul24(y_bottom, pitch) + kIntegralColOffset;  // rows, pitch < 2^24
    int ret_val = integ[bot + x_right];
    ret_val -= integ[bot + x_left];
    ret_val += integ[top + x_left];
    ret_val -= integ[top + x_right];
    const int area = (int)__umul24(x_right - x_left, y_bottom - y_top);
    // (wave-uniform choice: the fix-up form only where some lane's box is beyond div_box_small's range)
    return (__all(area <= 8192) ? div_box_small(ret_val, area) : div_box(ret_val, area)) & 0xff;
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

// The same bits from 19x19 buffers that start on a dword (the gather path's LDS scratch), per-lane constants hoisted by the
// caller: lane = 8*centre + offset; the two 9-byte strips sit at arbitrary byte offsets, so the covering aligned dwords are
// fetched and the strips shifted out; SSD = sum c^2 + sum p^2 - 2 sum c*p over the first eight bytes (packed u8 dot
// products) + the ninth byte's squared difference.
struct MipStripLane {
    int cw, pw;  // dword index of the first covering dword in the current / previous buffer
    int cs, ps;  // byte shifts
};
__device__ __forceinline__ MipStripLane mip_strip_lane()
{
    const int lane = lane_id();
    const int mc = lane >> 3, mi = lane & 7;
    const int mcx = (0xDDD99555u >> (4 * mc)) & 15, mcy = (0xD95D5D95u >> (4 * mc)) & 15;
    const int mdx = (int)((0x14787410u >> (4 * mi)) & 15) - 4, mdy = (int)((0x10147874u >> (4 * mi)) & 15) - 4;
    const int base_c = (mcy - 1) * kPatch + (mcx - 1), base_p = (mcy + mdy - 1) * kPatch + (mcx + mdx - 1);
    return MipStripLane{base_c >> 2, base_p >> 2, base_c & 3, base_p & 3};
}
__device__ __forceinline__ uint64_t mip_bits_strips(const uint32_t *cur19, const uint32_t *prev19, const MipStripLane m, int mip_theta)
{
    uint32_t cd[3], pd[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        cd[w] = cur19[m.cw + w];
        pd[w] = prev19[m.pw + w];
    }
    const uint32_t c0 = __builtin_amdgcn_alignbyte(cd[1], cd[0], m.cs), c1 = __builtin_amdgcn_alignbyte(cd[2], cd[1], m.cs);
    const uint32_t p0 = __builtin_amdgcn_alignbyte(pd[1], pd[0], m.ps), p1 = __builtin_amdgcn_alignbyte(pd[2], pd[1], m.ps);
    const int d8 = (int)((cd[2] >> (8 * m.cs)) & 0xffu) - (int)((pd[2] >> (8 * m.ps)) & 0xffu);
    const uint32_t sq = __builtin_amdgcn_udot4(c0, c0, __builtin_amdgcn_udot4(c1, c1, (uint32_t)__mul24(d8, d8), false), false) +
                        __builtin_amdgcn_udot4(p0, p0, __builtin_amdgcn_udot4(p1, p1, 0u, false), false);
    const uint32_t cross = __builtin_amdgcn_udot4(c0, p0, __builtin_amdgcn_udot4(c1, p1, 0u, false), false);
    return __ballot((int)(sq - 2u * cross) > mip_theta);
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
