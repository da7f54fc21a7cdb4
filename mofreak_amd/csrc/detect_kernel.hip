// Keypoint detector for gfx950: BRISK scale-space corners on the frame-difference image.
//
// Replaces, per frame pair, BriskFeatureDetector(30).detect(diff_img) (MoFREAKUtilities.cpp:420-423), i.e.
// BriskScaleSpace::constructPyramid + getKeypoints (brisk.cpp:572-704) with the OAST 9/16 detector and the
// AGAST 5/8 score behind them (oast9_16.cc:46, oast9_16_nms.cc:42, agast5_8_nms.cc:42).
//
// The reference walks its candidates one by one, asks for corner scores lazily and caches them; which scores are in the
// cache when a tie is broken decides the tie.  Here every step is data parallel and scores are computed where the reference
// computes them -- at corners, and in the cells refinement walks read -- with the reference's results bit for bit:
//
//   pyramid     difference image, then the 2/3 and 1/2 resamplers as closed forms of their SSE sequences (every output byte
//               depends on which part of the SIMD loop produced it: main blocks, the odd block, the scalar tail): all
//               layers of a 12-row band in one launch, lower layers read from LDS (det_pyramid_fused_kernel; odd widths,
//               very wide frames and octaves = 4: a kernel per layer)
//   corners     det_corner_kernel: a necessary four-point test on every pixel (packed 16-bit, four pixels a lane), the
//               survivors of a 64 x 64 tile compacted, the full score -- max over the 16 arcs of the arc's smallest
//               |difference|, minus one: what the bisection around the decision tree converges to -- only for them;
//               score plane (corner score or 0), a 64-bit hit mask per tile row, per-row counts
//   candidates  the corners in raster order per layer straight from the hit masks (det_scan_kernel, det_candidates_kernel),
//               classified by the strict part of isMax2D
//   refinement  det_window_kernel scores the cells a walk can read (the windows above and below, the own patch) per walker
//               from the image into a 64-byte record; det_walk_kernel walks on them exactly as refine3D does and
//               publishes what it asked for in the layer above
//   ties        isMax2D breaks ties on the reference's RAW score cache, which holds a score only where one has
//               been asked for before -- so the outcome depends on the processing order.  Reproduced exactly:
//               refinement marks the cells it asks for in the layer above ("touch" map), a maximum that reaches
//               its own 3x3 patch marks itself ("status" map), and a tie is decided from score * (detected |
//               touched from below | inside the patch of a raster-earlier maximum).  Layer 0's ties take their first
//               look chip-wide (det_tie_first_kernel); the rest -- and the ties that depend on each other -- are
//               resolved by one workgroup per pair, layer after layer (det_tie_kernel).
//   emission    ordered compaction in (layer, raster) order -- the order of the reference's keypoint vector, which
//               the rows of a .mofreak file inherit -- and every maximum and tie takes back the bytes it put into the
//               two maps, which are all zero between calls.
//
// Integer/byte work bound by vector issue and, in the tie kernels, by scattered line fetches; no MFMA.  Floating point mirrors the reference's expressions
// (float vs double literals) one operation at a time; compile with -ffp-contract=off.
#include <type_traits>

#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kDetThreads = 256;
// cand_spec bits (kWasTie: set by the candidate kernel; the others by the refinement run ahead of a tie's decision)
constexpr uint8_t kEmit = 1, kReached = 2, kWasTie = 0x80;

struct PairView {
    const DetGeom *g;
    const uint8_t *img;
    const uint8_t *score;
    uint8_t *score_rw;  // the same plane, for the refinement's writes
    uint8_t *touch;
    uint8_t *status;
};

// What a C `float` expression of brisk.cpp means.  The reference is a 32-bit Visual Studio 2010 project without an /arch
// option: x87 code, the FPU at 53-bit precision, intermediates of an expression kept in FPU registers and rounded to float
// only where they are assigned, cast, passed or returned (MOFREAK_FP_X87, the default: Fp<true>, intermediates in double);
// MOFREAK_FP_SSE rounds every float operation to float (Fp<false>, intermediates in float).  The refinement kernel is
// instantiated for both; operands are floats (or ints converted to float), results stay in R until they are stored.
template <bool X87>
struct Fp {
    typedef typename std::conditional<X87, double, float>::type R;
    static __device__ __forceinline__ R mul(R a, R b) { return a * b; }
    static __device__ __forceinline__ R add(R a, R b) { return a + b; }
    static __device__ __forceinline__ R sub(R a, R b) { return a - b; }
    static __device__ __forceinline__ R div(R a, R b) { return a / b; }
};

__device__ __forceinline__ PairView pair_view(const DetArgs &a, int p)
{
    PairView v;
    v.g = a.dg;
    const int64_t o = (int64_t)p * a.dg->plane_bytes;
    v.img = a.img + o;
    v.score = a.score + o;
    v.score_rw = a.score + o;
    v.touch = a.touch + o;
    v.status = a.status + o;
    return v;
}

// the batch's geometry into LDS (a per-thread layer index into the device copy is a memory round trip per field)
__device__ __forceinline__ void geom_to_lds(DetGeom *dst, const DetGeom *src)
{
    static_assert(sizeof(DetGeom) % 4 == 0, "copied as dwords");
    for (int i = threadIdx.x; i < (int)(sizeof(DetGeom) / 4); i += blockDim.x) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
}

__device__ __forceinline__ int avg_u8(int a, int b) { return (a + b + 1) >> 1; }  // _mm_avg_epu8

__device__ __forceinline__ unsigned long long load8(const uint8_t *p)
{
    unsigned long long q;
    __builtin_memcpy(&q, p, 8);  // one unaligned 8-byte load
    return q;
}

// ------------------------------------------------------------------ pyramid
// The batch's counters -- per-row corner counts of every pair -- are zeroed by the first kernel of the batch (and, before a call's
// first batch, the running keypoint total and the status word in front of them): nobody counts before det_corner_kernel.
__device__ __forceinline__ void zero_batch_counters(const DetArgs &a, int total_rows)
{
    const int64_t n = (int64_t)a.n_pairs * (total_rows + 1);
    const int64_t bid = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z), n_blocks = (int64_t)gridDim.x * gridDim.y * gridDim.z;
    for (int64_t i = bid * kDetThreads + threadIdx.x; i < n; i += n_blocks * kDetThreads) a.row_count[i] = 0;
    if (bid == 0 && threadIdx.x < 4 && a.first_pair == 0) (a.row_count - 4)[threadIdx.x] = 0;
}

__global__ __launch_bounds__(kDetThreads) void det_diff_kernel(DetArgs a)
{
    zero_batch_counters(a, a.dg->total_rows);
    const int p = blockIdx.y;
    const int W = a.dg->L[0].w, H = a.dg->L[0].h;
    const int per_row = (W + 15) / 16, t = blockIdx.x * kDetThreads + threadIdx.x;  // 16-pixel pieces, row after row
    const int y = t / per_row, x0 = (t - y * per_row) * 16;
    if (y >= H) return;
    const uint8_t *c = a.f.cur + (int64_t)p * a.f.pair_stride + (int64_t)y * a.f.row_stride + x0;
    const uint8_t *q = a.f.prev ? a.f.prev + (int64_t)p * a.f.pair_stride + (int64_t)y * a.f.row_stride + x0 : nullptr;
    uint8_t *d = a.img + (int64_t)p * a.dg->plane_bytes + a.dg->L[0].off + (int64_t)y * W + x0;
    // cv::absdiff (MoFREAKUtilities.cpp:413-414), sixteen pixels per thread; 16 bytes each way where everything is aligned
    const bool wide = x0 + 16 <= W && (((uintptr_t)c | (uintptr_t)d | (q ? (uintptr_t)q : 0)) & 15) == 0;
    if (wide) {
        const uint4 u = *reinterpret_cast<const uint4 *>(c), v = q ? *reinterpret_cast<const uint4 *>(q) : make_uint4(0, 0, 0, 0);
        auto word = [](uint32_t s, uint32_t t) {
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) r |= (uint32_t)absdiff_u8(s, t, k) << (8 * k);
            return r;
        };
        *reinterpret_cast<uint4 *>(d) = make_uint4(word(u.x, v.x), word(u.y, v.y), word(u.z, v.z), word(u.w, v.w));
        return;
    }
    const int n = min(16, W - x0);
    for (int k = 0; k < n; ++k) {
        const int u = c[k], v = q ? q[k] : 0;
        d[k] = (uint8_t)(u > v ? u - v : v - u);
    }
}

// BriskLayer::halfsample (brisk.cpp:1840-1972).  One output pixel:
__device__ __forceinline__ int half_pixel(const uint8_t *u, const uint8_t *l, int c, int hsize, int end)
{
    if (c < 16 * end)  // pairs of 16-byte blocks: rounding average of the two vertical rounding averages
        return avg_u8(avg_u8(u[2 * c], l[2 * c]), avg_u8(u[2 * c + 1], l[2 * c + 1]));
    if (c < 8 * hsize)  // the odd block: truncating mean of the vertical averages (:1929-1933)
        return (avg_u8(u[2 * c], l[2 * c]) + avg_u8(u[2 * c + 1], l[2 * c + 1])) / 2;
    // scalar tail (:1949-1956): columns k and k+1 behind the last whole block, not 2k and 2k+1
    const int k = c - 8 * hsize, b = 16 * hsize;
    return (u[b + k] + u[b + k + 1] + l[b + k] + l[b + k + 1]) / 4;
}

// Four output pixels per thread: inside the main blocks and the odd block (whose extents are multiples of 8 outputs)
// they come from 8 bytes of each of the two source rows -- two 8-byte loads, one 4-byte store; the tail and the last
// pixels of a row go one by one.
__global__ __launch_bounds__(kDetThreads) void det_half_kernel(DetArgs a, int src_l, int dst_l)
{
    const DetLayer S = a.dg->L[src_l], D = a.dg->L[dst_l];
    const int c = 4 * (blockIdx.x * kDetThreads + threadIdx.x), r = blockIdx.y, p = blockIdx.z;
    if (c >= D.w) return;
    const uint8_t *u = a.img + (int64_t)p * a.dg->plane_bytes + S.off + (int64_t)(2 * r) * S.w, *l = u + S.w;
    uint8_t *out = a.img + (int64_t)p * a.dg->plane_bytes + D.off + (int64_t)r * D.w + c;
    const int hsize = S.w / 16, end = hsize / 2;
    if (c + 4 <= 8 * hsize && c + 4 <= D.w) {
        const bool main_blocks = c < 16 * end;
        const unsigned long long qu = load8(u + 2 * c), ql = load8(l + 2 * c);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v0 = avg_u8((int)((qu >> (16 * k)) & 0xff), (int)((ql >> (16 * k)) & 0xff));
            const int v1 = avg_u8((int)((qu >> (16 * k + 8)) & 0xff), (int)((ql >> (16 * k + 8)) & 0xff));
            packed |= (uint32_t)(main_blocks ? avg_u8(v0, v1) : (v0 + v1) / 2) << (8 * k);
        }
        __builtin_memcpy(out, &packed, 4);
        return;
    }
    for (int k = 0; k < 4 && c + k < D.w; ++k) out[k] = (uint8_t)half_pixel(u, l, c + k, hsize, end);
}

// BriskLayer::twothirdsample (brisk.cpp:1974-2065).  One output pixel of output row r2 (mid = source row 3 * (r2 / 2) + 1,
// outer = the row above it for the upper output row of the pair, below it for the lower one):
__device__ __forceinline__ int twothird_pixel(const uint8_t *outer, const uint8_t *mid, int c, int hsize)
{
    if (c < 10 * hsize) {
        // shuffle masks of :1982-1984: outer column / "middle" column per output byte; the last pair reads 12, not 13
        const int i = c / 10, m = c - 10 * i;
        const int t2 = (int)((0xEC'B986'5320ull >> (4 * m)) & 15);  // {0,2,3,5,6,8,9,11,12,14}
        const int t1 = (int)((0xCC'AA77'4411ull >> (4 * m)) & 15);  // {1,1,4,4,7,7,10,10,12,12}
        const int x2 = 15 * i + t2, x1 = 15 * i + t1;
        const int v2 = avg_u8(avg_u8(outer[x2], mid[x2]), outer[x2]);
        const int v1 = avg_u8(avg_u8(outer[x1], mid[x1]), outer[x1]);
        return avg_u8(avg_u8(v2, v1), v2);
    }
    // scalar remainder (:2036-2052)
    const int k = c - 10 * hsize, j = 15 * hsize + 3 * (k >> 1);
    const int X2 = outer[j + 1], B2 = mid[j + 1];
    const int X = (k & 1) ? outer[j + 2] : outer[j], B = (k & 1) ? mid[j + 2] : mid[j];
    return ((4 * X + 2 * (X2 + B) + B2) / 9) & 0xff;
}

// A thread takes one block of the SSSE3 loop -- 15 source columns of two rows in, 10 bytes out: two 16-byte loads (the
// 16th byte belongs to the next block or the padded plane and is not used), the vertical step on all 15 columns, the
// horizontal step per output byte -- or, behind the blocks, one pixel of the scalar remainder.
__global__ __launch_bounds__(kDetThreads) void det_twothird_kernel(DetArgs a, int src_l, int dst_l)
{
    const DetLayer S = a.dg->L[src_l], D = a.dg->L[dst_l];
    const int t = blockIdx.x * kDetThreads + threadIdx.x, r2 = blockIdx.y, p = blockIdx.z;
    const int hsize = S.w / 15, rest = D.w - 10 * hsize;  // blocks, then `rest` remainder pixels
    if (t >= hsize + rest) return;
    const int r = r2 >> 1;
    const uint8_t *base = a.img + (int64_t)p * a.dg->plane_bytes + S.off;
    const uint8_t *mid = base + (int64_t)(3 * r + 1) * S.w;
    const uint8_t *outer = (r2 & 1) ? mid + S.w : mid - S.w;  // third row for the lower output row, first for the upper
    uint8_t *out = a.img + (int64_t)p * a.dg->plane_bytes + D.off + (int64_t)r2 * D.w;
    if (t >= hsize) {
        const int c = 10 * hsize + (t - hsize);
        out[c] = (uint8_t)twothird_pixel(outer, mid, c, hsize);
        return;
    }
    unsigned long long o[2], m[2];
    o[0] = load8(outer + 15 * t);
    o[1] = load8(outer + 15 * t + 8);
    m[0] = load8(mid + 15 * t);
    m[1] = load8(mid + 15 * t + 8);
    int v[15];  // _mm_avg_epu8(_mm_avg_epu8(outer, mid), outer) per column
#pragma unroll
    for (int x = 0; x < 15; ++x) {
        const int ov = (int)((o[x >> 3] >> (8 * (x & 7))) & 0xff), mv = (int)((m[x >> 3] >> (8 * (x & 7))) & 0xff);
        v[x] = avg_u8(avg_u8(ov, mv), ov);
    }
    constexpr int t2[10] = {0, 2, 3, 5, 6, 8, 9, 11, 12, 14}, t1[10] = {1, 1, 4, 4, 7, 7, 10, 10, 12, 12};  // the shuffle masks of :1982-1984
    unsigned long long lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const unsigned long long b = (unsigned long long)avg_u8(avg_u8(v[t2[k]], v[t1[k]]), v[t2[k]]);
        if (k < 8)
            lo |= b << (8 * k);
        else
            hi |= (uint32_t)b << (8 * (k - 8));
    }
    uint8_t *dst = out + 10 * t;
    __builtin_memcpy(dst, &lo, 8);
    const uint16_t hi16 = (uint16_t)hi;
    __builtin_memcpy(dst + 8, &hi16, 2);
}

// ---- the whole pyramid in one launch
// A workgroup takes a band of 12 rows of the frame pair and produces that band of every layer: |cur - prev| (12 rows),
// the 2/3 layer (8 rows), the halves (6, 4, 3, 2 rows).  Every resampler works inside a row (pairs / triples of rows), so
// full-width bands need no halo and every column quirk of the SIMD loops stays where it is; the lower layers of a band
// are read from LDS instead of from memory: the frames are read once and each layer is written once (8.1 MB per full-HD
// pair instead of 13.9 MB and one launch instead of six).  Used when the width is a multiple of 16 and the band fits 64 KB
// of LDS (frames up to 2800 pixels wide) and there are at most six layers; other shapes take the kernels above.
constexpr int kPyrBand = 12, kPyrLayers = 6;
struct PyrBand {
    int rows[kPyrLayers], pitch[kPyrLayers], base[kPyrLayers];  // rows of a band, LDS row pitch and offset per layer
    int bytes;
};
__host__ __device__ inline PyrBand pyr_band(const DetGeom &g)
{
    constexpr int r[kPyrLayers] = {12, 8, 6, 4, 3, 2};
    PyrBand b;
    int at = 0;
    for (int l = 0; l < kPyrLayers; ++l) {
        b.rows[l] = r[l];
        b.pitch[l] = l < g.n_layers ? ((g.L[l].w + 15) & ~15) + 16 : 0;  // 16 bytes behind a row: block loads may run past its end
        b.base[l] = at;
        if (l + 2 < g.n_layers || (l == 0 && g.n_layers > 1)) at += r[l] * b.pitch[l];  // kept in LDS only if a later layer reads it
    }
    b.bytes = at;
    return b;
}

// one block of the 2/3 resampler from two rows in LDS: 15 columns in, 10 bytes out (det_twothird_kernel's arithmetic)
__device__ __forceinline__ void twothird_block(const uint8_t *outer, const uint8_t *mid, uint8_t *dst_lds, uint8_t *dst)
{
    unsigned long long o[2], m[2];
    o[0] = load8(outer);
    o[1] = load8(outer + 8);
    m[0] = load8(mid);
    m[1] = load8(mid + 8);
    int v[15];  // _mm_avg_epu8(_mm_avg_epu8(outer, mid), outer) per column
#pragma unroll
    for (int x = 0; x < 15; ++x) {
        const int ov = (int)((o[x >> 3] >> (8 * (x & 7))) & 0xff), mv = (int)((m[x >> 3] >> (8 * (x & 7))) & 0xff);
        v[x] = avg_u8(avg_u8(ov, mv), ov);
    }
    constexpr int t2[10] = {0, 2, 3, 5, 6, 8, 9, 11, 12, 14}, t1[10] = {1, 1, 4, 4, 7, 7, 10, 10, 12, 12};  // the shuffle masks of :1982-1984
    unsigned long long lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const unsigned long long b = (unsigned long long)avg_u8(avg_u8(v[t2[k]], v[t1[k]]), v[t2[k]]);
        if (k < 8)
            lo |= b << (8 * k);
        else
            hi |= (uint32_t)b << (8 * (k - 8));
    }
    const uint16_t hi16 = (uint16_t)hi;
    __builtin_memcpy(dst, &lo, 8);
    __builtin_memcpy(dst + 8, &hi16, 2);
    if (dst_lds) {
        __builtin_memcpy(dst_lds, &lo, 8);
        __builtin_memcpy(dst_lds + 8, &hi16, 2);
    }
}

// a layer's band rows halved: four output pixels per item where the SIMD loop's blocks produced them, else pixel by pixel
__device__ __forceinline__ void half_band(const uint8_t *src, int src_pitch, int src_w, uint8_t *dst_lds, int dst_pitch, uint8_t *dst, int dst_w, int n_rows)
{
    const int hsize = src_w / 16, end = hsize / 2, per_row = (dst_w + 3) / 4;
    for (int it = threadIdx.x; it < n_rows * per_row; it += kDetThreads) {
        const int r = it / per_row, c = 4 * (it - r * per_row);
        const uint8_t *u = src + (2 * r) * src_pitch, *l = u + src_pitch;
        uint8_t o[4];
        if (c + 4 <= 8 * hsize && c + 4 <= dst_w) {
            const bool main_blocks = c < 16 * end;
            const unsigned long long qu = *reinterpret_cast<const unsigned long long *>(u + 2 * c), ql = *reinterpret_cast<const unsigned long long *>(l + 2 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v0 = avg_u8((int)((qu >> (16 * k)) & 0xff), (int)((ql >> (16 * k)) & 0xff));
                const int v1 = avg_u8((int)((qu >> (16 * k + 8)) & 0xff), (int)((ql >> (16 * k + 8)) & 0xff));
                o[k] = (uint8_t)(main_blocks ? avg_u8(v0, v1) : (v0 + v1) / 2);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = c + k < dst_w ? (uint8_t)half_pixel(u, l, c + k, hsize, end) : 0;
        }
        const uint32_t packed = (uint32_t)o[0] | (uint32_t)o[1] << 8 | (uint32_t)o[2] << 16 | (uint32_t)o[3] << 24;
        if (dst_lds) *reinterpret_cast<uint32_t *>(dst_lds + r * dst_pitch + c) = packed;  // (the pitch covers whole dwords)
        uint8_t *dg = dst + (int64_t)r * dst_w + c;
        if (c + 4 <= dst_w)
            __builtin_memcpy(dg, &packed, 4);
        else
            for (int k = 0; c + k < dst_w; ++k) dg[k] = o[k];
    }
}

__global__ __launch_bounds__(kDetThreads) void det_pyramid_fused_kernel(DetArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t band_lds[];
    zero_batch_counters(a, a.g.total_rows);
    const int p = blockIdx.y, band = blockIdx.x, n_layers = a.g.n_layers;
    const PyrBand B = pyr_band(a.g);  // from the argument block: scalar arithmetic
    uint8_t *plane = a.img + (int64_t)p * a.g.plane_bytes;
    // rows of this band that exist, per layer
    int nr[kPyrLayers];
#pragma unroll
    for (int l = 0; l < kPyrLayers; ++l) nr[l] = l < n_layers ? max(0, min(B.rows[l], a.g.L[l].h - band * B.rows[l])) : 0;
    const int W = a.g.L[0].w;
    {  // layer 0: cv::absdiff (MoFREAKUtilities.cpp:413-414), sixteen pixels per item
        const uint8_t *cur = a.f.cur + (int64_t)p * a.f.pair_stride, *prev = a.f.prev ? a.f.prev + (int64_t)p * a.f.pair_stride : nullptr;
        const int per_row = W / 16;
        uint8_t *l0 = plane + a.g.L[0].off + (int64_t)band * kPyrBand * W;
        for (int it = threadIdx.x; it < nr[0] * per_row; it += kDetThreads) {
            const int r = it / per_row, k = it - r * per_row;
            const int64_t src = (int64_t)(band * kPyrBand + r) * a.f.row_stride + 16 * k;
            uint4 u, v = make_uint4(0, 0, 0, 0);
            __builtin_memcpy(&u, cur + src, 16);
            if (prev) __builtin_memcpy(&v, prev + src, 16);
            auto word = [](uint32_t s, uint32_t t) {
                uint32_t d = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) d |= (uint32_t)absdiff_u8(s, t, q) << (8 * q);
                return d;
            };
            const uint4 d = make_uint4(word(u.x, v.x), word(u.y, v.y), word(u.z, v.z), word(u.w, v.w));
            if (n_layers > 1) *reinterpret_cast<uint4 *>(band_lds + B.base[0] + r * B.pitch[0] + 16 * k) = d;
            *reinterpret_cast<uint4 *>(l0 + (int64_t)r * W + 16 * k) = d;
        }
    }
    if (n_layers == 1) return;
    __syncthreads();
    {  // layer 1 = 2/3 of layer 0 (BriskLayer::twothirdsample, brisk.cpp:1974-2065): a block of 15 -> 10 columns or a remainder pixel per item
        const int w1 = a.g.L[1].w, hsize = W / 15, rest = w1 - 10 * hsize, per_row = hsize + rest;
        const bool keep = 3 < n_layers;  // layer 3 reads it
        uint8_t *l1 = plane + a.g.L[1].off + (int64_t)band * B.rows[1] * w1;
        for (int it = threadIdx.x; it < nr[1] * per_row; it += kDetThreads) {
            const int r2 = it / per_row, t = it - r2 * per_row, r = r2 >> 1;
            const uint8_t *mid = band_lds + B.base[0] + (3 * r + 1) * B.pitch[0];
            const uint8_t *outer = (r2 & 1) ? mid + B.pitch[0] : mid - B.pitch[0];  // third row for the lower output row, first for the upper
            uint8_t *dl = keep ? band_lds + B.base[1] + r2 * B.pitch[1] : nullptr, *dg = l1 + (int64_t)r2 * w1;
            if (t < hsize) {
                twothird_block(outer + 15 * t, mid + 15 * t, dl ? dl + 10 * t : nullptr, dg + 10 * t);
            } else {
                const int c = 10 * hsize + (t - hsize);
                const uint8_t b = (uint8_t)twothird_pixel(outer, mid, c, hsize);
                if (dl) dl[c] = b;
                dg[c] = b;
            }
        }
    }
    if (n_layers > 2)  // layer 2 = half of layer 0 (BriskLayer::halfsample, brisk.cpp:1840-1972)
        half_band(band_lds + B.base[0], B.pitch[0], W, 4 < n_layers ? band_lds + B.base[2] : nullptr, B.pitch[2],
                  plane + a.g.L[2].off + (int64_t)band * B.rows[2] * a.g.L[2].w, a.g.L[2].w, nr[2]);
    if (n_layers <= 3) return;
    __syncthreads();
    half_band(band_lds + B.base[1], B.pitch[1], a.g.L[1].w, 5 < n_layers ? band_lds + B.base[3] : nullptr, B.pitch[3],
              plane + a.g.L[3].off + (int64_t)band * B.rows[3] * a.g.L[3].w, a.g.L[3].w, nr[3]);
    if (n_layers > 4)
        half_band(band_lds + B.base[2], B.pitch[2], a.g.L[2].w, nullptr, 0, plane + a.g.L[4].off + (int64_t)band * B.rows[4] * a.g.L[4].w, a.g.L[4].w, nr[4]);
    if (n_layers <= 5) return;
    __syncthreads();
    half_band(band_lds + B.base[3], B.pitch[3], a.g.L[3].w, nullptr, 0, plane + a.g.L[5].off + (int64_t)band * B.rows[5] * a.g.L[5].w, a.g.L[5].w, nr[5]);
}

// ------------------------------------------------------------------ dense corner scores
// score = largest b in [1, 254] for which 9 contiguous ring pixels are all > c + b or all < c - b, 0 if none:
// what OastDetector9_16::cornerScore's bisection converges to (oast9_16_nms.cc:42-2116), for any start value <= it.
// over the 16 arcs of 9 contiguous ring pixels: the largest arc minimum and the smallest arc maximum of the raw values
// three-input minimum / maximum of non-negative values, spelled out: left to itself the compiler re-associates the chains
// below into more two-input operations than this count
__device__ __forceinline__ int min3i(int a, int b, int c)
{
    int r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    int r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void arc9_extremes(const int (&p)[16], int &max_of_min, int &min_of_max)
{
    // an arc of 9 = three runs of 3: 16 + 16 three-input operations per side, and 8 more to reduce the 16 arcs
    int lo3[16], hi3[16], lo9[16], hi9[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo3[k] = min3i(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        hi3[k] = max3i(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo9[k] = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        hi9[k] = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    }
    int lo5[5], hi5[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        lo5[k] = max3i(lo9[3 * k], lo9[3 * k + 1], lo9[3 * k + 2]);
        hi5[k] = min3i(hi9[3 * k], hi9[3 * k + 1], hi9[3 * k + 2]);
    }
    max_of_min = max(max3i(lo5[0], lo5[1], lo5[2]), max3i(lo5[3], lo5[4], lo9[15]));
    min_of_max = min(min3i(hi5[0], hi5[1], hi5[2]), min3i(hi5[3], hi5[4], hi9[15]));
}

constexpr int kScoreTileW = 64, kScoreTileH = 32, kScoreLdsW = 72;  // LDS rows start at image column x0 - 4: aligned dwords

// a tile of TILE_H rows of a layer's image with its 3-pixel ring halo -> LDS (rows of kScoreLdsW bytes, first column x0 - 4)
template <int TILE_H>
__device__ __forceinline__ void score_tile_load(uint8_t *tile, const uint8_t *img, const DetLayer &L, int x0, int y0)
{
    if ((L.w & 3) == 0) {  // rows start on dword boundaries (layer planes are 64-byte aligned): 18 aligned dwords per tile row
        for (int t = threadIdx.x; t < (TILE_H + 6) * (kScoreLdsW / 4); t += kDetThreads) {
            const int r = t / (kScoreLdsW / 4), k = t - r * (kScoreLdsW / 4);
            const int gx = x0 - 4 + 4 * k, gy = y0 - 3 + r;
            const bool in = gx >= 0 && gx < L.w && gy >= 0 && gy < L.h;
            const uint32_t v = in ? *reinterpret_cast<const uint32_t *>(img + (int64_t)gy * L.w + gx) : 0u;
            *reinterpret_cast<uint32_t *>(tile + r * kScoreLdsW + 4 * k) = v;
        }
    } else {
        for (int t = threadIdx.x; t < (TILE_H + 6) * kScoreLdsW; t += kDetThreads) {
            const int r = t / kScoreLdsW, c = t - r * kScoreLdsW;
            const int gx = x0 - 4 + c, gy = y0 - 3 + r;
            tile[r * kScoreLdsW + c] = (gx >= 0 && gx < L.w && gy >= 0 && gy < L.h) ? img[(int64_t)gy * L.w + gx] : 0;
        }
    }
}

// the OAST 9/16 score of the pixel at t (a byte of an image held with row pitch P): Bresenham circle of radius 3 in the
// order of OastDetector9_16::init_pattern (oast9_16.h:74-92)
template <int P>
__device__ __forceinline__ int ring_score(const uint8_t *t)
{
    const int c = t[0];
    int q[16];
    q[0] = t[-3];
    q[1] = t[-P - 3];
    q[2] = t[-2 * P - 2];
    q[3] = t[-3 * P - 1];
    q[4] = t[-3 * P];
    q[5] = t[-3 * P + 1];
    q[6] = t[-2 * P + 2];
    q[7] = t[-P + 3];
    q[8] = t[3];
    q[9] = t[P + 3];
    q[10] = t[2 * P + 2];
    q[11] = t[3 * P + 1];
    q[12] = t[3 * P];
    q[13] = t[3 * P - 1];
    q[14] = t[2 * P - 2];
    q[15] = t[P - 3];
    int arc_lo, arc_hi;
    arc9_extremes(q, arc_lo, arc_hi);
    const int vb = arc_lo - c, vd = c - arc_hi;  // brightest all-brighter arc margin, darkest all-darker arc margin
    return max(max(vb, vd) - 1, 0);
}

// Component entry point (mofreak_brisk_pyramid with scores_out): the score of EVERY pixel of a layer.  The detector itself
// scores only where the reference does (det_corner_kernel, det_refine_kernel).
__global__ __launch_bounds__(kDetThreads) void det_dense_score_kernel(DetArgs a, int layer)
{
    __shared__ __attribute__((aligned(4))) uint8_t tile[(kScoreTileH + 6) * kScoreLdsW];
    const DetLayer L = a.dg->L[layer];
    const int p = blockIdx.z, x0 = blockIdx.x * kScoreTileW, y0 = blockIdx.y * kScoreTileH;
    score_tile_load<kScoreTileH>(tile, a.img + (int64_t)p * a.dg->plane_bytes + L.off, L, x0, y0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *score = a.score + (int64_t)p * a.dg->plane_bytes + L.off;
#pragma unroll
    for (int it = 0; it < kScoreTileH / 4; ++it) {
        const int ry = wave + 4 * it, x = x0 + lane, y = y0 + ry;
        int s = 0;
        if (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) s = ring_score<kScoreLdsW>(tile + (ry + 3) * kScoreLdsW + lane + 4);
        if (x < L.w && y < L.h) score[(uint32_t)y * (uint32_t)L.w + (uint32_t)x] = (uint8_t)s;
    }
}

// ------------------------------------------------------------------ corners (what OastDetector9_16::detect + getAgastPoints leave behind)
// One workgroup per 64 x 64 tile of a layer, all layers in one launch.  The reference's decision tree answers "is there an
// arc of 9 ring pixels all brighter than c + t or all darker than c - t" with a handful of comparisons for most pixels;
// the data-parallel counterpart: (1) a necessary condition on the four compass points of the ring -- an arc of 9 holds at
// least two of them, so the second largest of their differences to the centre must exceed t (or the second smallest lie
// below -t) -- evaluated for four neighbouring pixels per lane on aligned dwords with packed 16-bit arithmetic (14
// vector operations per pixel; the kernel is bound by vector issue); it passes a few percent of a difference image's
// pixels; (2) the survivors of a tile are compacted and only they get the full score (the arc extremes), with full
// wavefronts.  Leaves, per tile: the score plane (score where >= threshold, 0 elsewhere: the reference's cache after
// getAgastPoints, brisk.cpp:1676-1690), a 64-bit hit mask per tile row, and the per-row corner counts.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// bit i: pixel i of the dword C (its left / right neighbours at distance 3 in L4 / R4, upper / lower in U / D) passes
__device__ __forceinline__ uint32_t compass_test4(uint32_t C, uint32_t L4, uint32_t R4, uint32_t U, uint32_t D, int t)
{
    const v2s T1 = {(short)(t + 1), (short)(t + 1)}, T = {(short)t, (short)t};
    uint32_t res = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t sel = h ? 0x0c030c02u : 0x0c010c00u;  // bytes (2, 3) / (0, 1), zero-extended to 16 bits each
        const v2s c = as_v2s(__builtin_amdgcn_perm(C, C, sel)), l = as_v2s(__builtin_amdgcn_perm(L4, L4, sel)), r = as_v2s(__builtin_amdgcn_perm(R4, R4, sel)),
                  u = as_v2s(__builtin_amdgcn_perm(U, U, sel)), d = as_v2s(__builtin_amdgcn_perm(D, D, sel));
        const v2s dl = l - c, dr = r - c, du = u - c, dd = d - c;
        const v2s mx1 = __builtin_elementwise_max(dl, dr), mn1 = __builtin_elementwise_min(dl, dr), mx2 = __builtin_elementwise_max(du, dd),
                  mn2 = __builtin_elementwise_min(du, dd);
        const v2s lo_of_max = __builtin_elementwise_min(mx1, mx2), hi_of_min = __builtin_elementwise_max(mn1, mn2);
        const v2s second_largest = __builtin_elementwise_max(lo_of_max, hi_of_min), second_smallest = __builtin_elementwise_min(lo_of_max, hi_of_min);
        // second_largest >= t + 1 (sign of the difference clear) or second_smallest + t < 0 (sign set), per 16-bit half
        const uint32_t m = (~as_u32(second_largest - T1) | as_u32(second_smallest + T)) & 0x80008000u;
        res |= ((m >> 15) & 1u) << (2 * h) | (m >> 31) << (2 * h + 1);
    }
    return res;
}

// the corner kernel's image tile in LDS: rows of 96 bytes from image column x0 - 16 (whole 16-byte pieces: six per row)
constexpr int kCornerLdsW = 96, kCornerLdsX = 16;

__global__ __launch_bounds__(kDetThreads) void det_corner_kernel(DetArgs a)
{
    constexpr int kPasses = kDetTileH / 16;  // 16 rows of 16 four-pixel groups per pass of the 256 threads
    __shared__ __attribute__((aligned(16))) uint8_t tile[(kDetTileH + 6) * kCornerLdsW];
    __shared__ __attribute__((aligned(16))) uint8_t out[kDetTileH * kDetTileW];
    __shared__ uint16_t list[kDetTileH * kDetTileW];
    __shared__ unsigned long long row_mask[kDetTileH];
    __shared__ int n_list;
    // Workgroups go to the eight XCDs in turn (each with an L2 of its own): an XCD takes a contiguous eighth of the pair's
    // tile list, so that neighbouring tiles -- which share their halo's cache lines -- meet in one L2 (the grid's x extent
    // is a multiple of 8: launch_det_corners).
    const int p = blockIdx.y, n_tiles = a.g.tile_start[a.g.n_layers], per_xcd = (n_tiles + 7) / 8;
    const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (t >= min(n_tiles, (int)((blockIdx.x & 7) + 1) * per_xcd)) return;
    int layer = 0;  // from the argument block's copy of the geometry with constant indices: scalar compares, no memory
#pragma unroll
    for (int k = 1; k < kDetMaxLayers; ++k) layer += (k < a.g.n_layers && t >= a.g.tile_start[k]) ? 1 : 0;
    const DetLayer L = a.dg->L[layer];
    const int tiles_x = a.dg->tiles_x[layer], tl = t - a.dg->tile_start[layer];
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x, x0 = tx * kDetTileW, y0 = ty * kDetTileH;
    const int64_t plane = (int64_t)p * a.dg->plane_bytes + L.off;
    {  // tile + 3-pixel halo -> LDS, sixteen bytes per item (positions outside the image read as zero and are never scored)
        const uint8_t *img = a.img + plane;
        constexpr int kPieces = kCornerLdsW / 16;
        for (int it = threadIdx.x; it < (kDetTileH + 6) * kPieces; it += kDetThreads) {
            const int r = it / kPieces, k = it - r * kPieces;
            const int gx = x0 - kCornerLdsX + 16 * k, gy = y0 - 3 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy >= 0 && gy < L.h) {
                const uint8_t *src = img + (int64_t)gy * L.w + gx;
                if (gx >= 0 && gx + 16 <= L.w) {
                    __builtin_memcpy(&v, src, 16);
                } else if (gx + 16 > 0 && gx < L.w) {  // a piece that straddles the image's edge: byte by byte
                    uint8_t b[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) b[q] = (gx + q >= 0 && gx + q < L.w) ? src[q] : (uint8_t)0;
                    __builtin_memcpy(&v, b, 16);
                }
            }
            *reinterpret_cast<uint4 *>(tile + r * kCornerLdsW + 16 * k) = v;
        }
    }
    static_assert(kDetTileH * kDetTileW == 16 * kDetThreads, "a thread clears 16 bytes of the tile's scores");
    reinterpret_cast<uint4 *>(out)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < kDetTileH) row_mask[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int thr = a.safe_threshold;
    // (1) the compass test: thread = (row r0 of a pass, group g of four pixels): a 16-bit mask of survivors per thread (four
    // passes of four pixels), one prefix sum over the wave and ONE reservation in the tile's list for all of them
    const int g = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    uint32_t colmask = 0;  // which of the group's four columns lie inside the scored region
#pragma unroll
    for (int i = 0; i < 4; ++i) colmask |= (x0 + 4 * g + i >= 3 && x0 + 4 * g + i < L.w - 3) ? 1u << i : 0u;
    uint32_t m16 = 0;
#pragma unroll
    for (int j = 0; j < kPasses; ++j) {
        const int ry = r0 + 16 * j, y = y0 + ry;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(tile + (ry + 3) * kCornerLdsW + kCornerLdsX) + g;  // the dword of columns 4g .. 4g + 3
        const uint32_t C = q[0], Cp = q[-1], Cn = q[1], U = q[-3 * (kCornerLdsW / 4)], D = q[3 * (kCornerLdsW / 4)];
        const uint32_t m4 = compass_test4(C, __builtin_amdgcn_alignbyte(C, Cp, 1), __builtin_amdgcn_alignbyte(Cn, C, 3), U, D, thr);
        m16 |= ((y >= 3 && y < L.h - 3) ? (m4 & colmask) : 0u) << (4 * j);
    }
    {
        const int cnt = __popc(m16);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int total = __shfl(incl, 63);
        if (total) {  // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&n_list, total);
            int k = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
            uint32_t m = m16;
            while (m) {  // a few bits per lane at most
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                list[k++] = (uint16_t)((r0 + 16 * (b >> 2)) << 6 | (4 * g + (b & 3)));
            }
        }
    }
    __syncthreads();
    // (2) the survivors' scores
    const int n = n_list;
    for (int i = threadIdx.x; i < n; i += kDetThreads) {
        const int e = list[i], ry = e >> 6, lx = e & 63;
        const int s = ring_score<kCornerLdsW>(tile + (ry + 3) * kCornerLdsW + lx + kCornerLdsX);
        if (s >= thr) {
            out[ry * kDetTileW + lx] = (uint8_t)s;
            atomicOr(&row_mask[ry], 1ull << lx);
        }
    }
    __syncthreads();
    // the tile's scores (zeros included: the plane is rewritten by every call), masks and counts
    {
        uint8_t *score = a.score + plane;
        const int ry = threadIdx.x >> 2, cx = (threadIdx.x & 3) * 16, x = x0 + cx, y = y0 + ry;
        if (y < L.h && x < L.w) {
            uint8_t *dst = score + (uint32_t)y * (uint32_t)L.w + (uint32_t)x;
            const uint4 v = reinterpret_cast<const uint4 *>(out)[threadIdx.x];
            if (x + 16 <= L.w) {
                __builtin_memcpy(dst, &v, 16);
            } else {
                for (int k = 0; k < L.w - x; ++k) dst[k] = out[ry * kDetTileW + cx + k];
            }
        }
    }
    if (threadIdx.x < kDetTileH && y0 + (int)threadIdx.x < L.h) {
        const int y = y0 + threadIdx.x;
        const unsigned long long m = row_mask[threadIdx.x];
        a.hit_mask[(int64_t)p * a.dg->mask_words + a.dg->mask_off[layer] + (int64_t)y * tiles_x + tx] = m;
        if (m) atomicAdd(&a.row_count[(int64_t)p * (a.dg->total_rows + 1) + L.row_base + y], __popcll(m));
    }
}

// exclusive scan of the per-row detection counts of one pair (all layers); one workgroup per pair
__global__ __launch_bounds__(kDetThreads) void det_scan_kernel(DetArgs a)
{
    __shared__ int part[kDetThreads];
    const int p = blockIdx.x, n = a.dg->total_rows;
    int32_t *rc = a.row_count + (int64_t)p * (n + 1);
    const int per = (n + kDetThreads - 1) / kDetThreads;
    const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += rc[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < kDetThreads; o <<= 1) {
        const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = rc[i];
        rc[i] = run;
        run += c;
    }
    __syncthreads();
    if (threadIdx.x == kDetThreads - 1) {
        const int total = part[kDetThreads - 1];
        rc[n] = total;
        if (total > a.cand_cap) atomicOr(a.status_word, 4);
    }
    __syncthreads();
    if (threadIdx.x <= a.dg->n_layers) {
        const int l = threadIdx.x;
        const int v = l < a.dg->n_layers ? rc[a.dg->L[l].row_base] : rc[n];
        a.layer_start[(int64_t)p * (kDetMaxLayers + 1) + l] = min(v, a.cand_cap);
    }
}

// The corners of a layer in raster order (the order of OastDetector9_16::detect), read off the hit masks.  A wave takes as
// many whole rows as fit its 64 lanes (a lane per 64-pixel word; the words of consecutive rows are consecutive in memory),
// or one row in stretches of 64 words; a prefix sum over the words' populations places every corner, then a lane per
// corner (the corners of a row cluster in few words: a lane per word would walk them one latency at a time) classifies
// it by the strict part of isMax2D (brisk.cpp:838-872) on the 3 x 3 scores around it.  A neighbour that is no corner
// holds 0 or a true score below the threshold in the score plane: never >= a corner's.
__global__ __launch_bounds__(kDetThreads) void det_candidates_kernel(DetArgs a)
{
    // the wave's row group is the same in every lane: kept in scalar registers, and so is everything that follows from it alone
    const int p = blockIdx.y, grp = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grp >= a.g.cand_group_start[a.g.n_layers]) return;
    // which layer: from the argument block's copy of the geometry with constant indices (scalar registers, no memory)
    int layer = 0;
#pragma unroll
    for (int k = 1; k < kDetMaxLayers; ++k) layer += (k < a.g.n_layers && grp >= a.g.cand_group_start[k]) ? 1 : 0;
    const DetLayer L = a.dg->L[layer];
    const int tiles_x = a.dg->tiles_x[layer], rpw = a.dg->cand_rows_per_wave[layer];
    const int y0 = (grp - a.dg->cand_group_start[layer]) * rpw;
    const int n_rows = min(rpw, L.h - y0), n_words = n_rows * tiles_x;  // rpw > 1: n_words <= 64
    const int32_t *rc = a.row_count + (int64_t)p * (a.dg->total_rows + 1) + L.row_base;
    if (rc[y0] == rc[y0 + n_rows]) return;  // no corner in these rows
    const int64_t plane = (int64_t)p * a.dg->plane_bytes + L.off;
    const uint8_t *sc = a.score + plane;
    const int64_t cbase = (int64_t)p * a.cand_cap;
    const unsigned long long *mrow = a.hit_mask + (int64_t)p * a.dg->mask_words + a.dg->mask_off[layer] + (int64_t)y0 * tiles_x;
    int carried = 0;  // one row in several stretches: its corners in the stretches before this one
    for (int w0 = 0; w0 < n_words; w0 += 64) {
        const int gw = w0 + lane;
        const bool in = gw < n_words;
        const int r = in ? gw / tiles_x : 0, wi = gw - r * tiles_x;  // the lane's word: row y0 + r, pixels 64 wi ..
        const unsigned long long m = in ? mrow[gw] : 0ull;
        const int row_first = rc[y0 + r];
        const int cnt = __popcll(m);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, 63);
        // corner j of the stretch (wave order = raster order) is candidate j + delta of the lane whose word holds it
        const int row_lane0 = rpw > 1 ? r * tiles_x : 0;  // the lane that holds the first word of this lane's row
        const int delta = row_first + carried - (__shfl(incl, row_lane0) - __shfl(cnt, row_lane0));
        const int where = (y0 + r) << 16 | wi;
        for (int j0 = 0; j0 < total; j0 += 64) {  // (wave-uniform trip count: the shuffles below read every lane)
            const int j = j0 + lane;
            int w = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int probe = __shfl(incl, w + step - 1);  // words 0 .. w + step - 1 hold <= j corners in all?
                if (probe <= j) w += step;
            }
            w = min(w, 63);
            const int before = __shfl(incl - cnt, w);
            const uint32_t mlo = (uint32_t)__shfl((int)(uint32_t)m, w), mhi = (uint32_t)__shfl((int)(uint32_t)(m >> 32), w);
            const int idx = j + __shfl(delta, w), wh = __shfl(where, w);
            // ... and is that word's (j - before)-th set bit, found by halving
            int k = j - before, pos = 0;
            uint32_t part = mlo;
            {
                const int c = __popc(mlo);
                if (k >= c) {
                    k -= c;
                    pos = 32;
                    part = mhi;
                }
            }
#pragma unroll
            for (int width = 16; width >= 1; width >>= 1) {
                const uint32_t lowmask = (1u << width) - 1u;
                const int c = __popc(part & lowmask);
                if (k >= c) {
                    k -= c;
                    pos += width;
                    part >>= width;
                }
                part &= lowmask;
            }
            const int x = 64 * (wh & 0xffff) + pos, y = wh >> 16;
            if (j < total && idx < a.cand_cap) {
                const uint8_t *srow = sc + (int64_t)y * L.w;
                uint32_t r0, r1, r2;  // scores x - 1 .. x + 2 of the three rows (a corner lies >= 3 pixels inside the layer)
                __builtin_memcpy(&r0, srow + x - 1 - L.w, 4);
                __builtin_memcpy(&r1, srow + x - 1, 4);
                __builtin_memcpy(&r2, srow + x - 1 + L.w, 4);
                const int s = (int)((r1 >> 8) & 0xff);
                int hi = 0, eq = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int v0 = (int)((r0 >> (8 * q)) & 0xff), v2 = (int)((r2 >> (8 * q)) & 0xff);
                    hi |= (v0 > s) | (v2 > s);
                    eq |= (v0 == s) | (v2 == s);
                }
                const int vl = (int)(r1 & 0xff), vr = (int)((r1 >> 16) & 0xff);
                hi |= (vl > s) | (vr > s);
                eq |= (vl == s) | (vr == s);
                const uint8_t flag = hi ? kDetNotMax : (eq ? kDetTie : kDetMax);
                a.cand_xy[cbase + idx] = (uint32_t)x | ((uint32_t)y << 16);
                a.cand_flag[cbase + idx] = flag;
                a.cand_emit[cbase + idx] = 0;
                a.cand_spec[cbase + idx] = flag == kDetTie ? kWasTie : 0;
                if (flag == kDetTie) a.status[plane + (int64_t)y * L.w + x] = kStPending;
            }
        }
        carried += total;
    }
}

// ------------------------------------------------------------------ refinement (thread per maximum)
// BriskLayer::getAgastScore(int, int, 1) (brisk.cpp:1685-1694) -- the corner score of an arbitrary cell, bisected from 1 --
// is computed here where the reference computes it: for the cells a refinement walk can ask for.  A walk reads at most
// 5 x 5 cells of a neighbouring layer, from the corner ((int)x_1 - 1, (int)y_1 - 1) of getScoreMaxAbove/Below's sampling
// square (patch and tie rings included), and its own 3 x 3 patch.  det_window_kernel scores those cells -- a thread per
// (walker, window), the 11 x 11 image bytes behind a window held in registers -- into a 64-byte record per walker;
// det_walk_kernel copies the record's windows into the thread's LDS and walks on them (the walk itself is a chain of
// data-dependent early exits: cell by cell on global memory it would pay a latency per step).
constexpr int kWinSide = 6, kWinRow = 8, kWinBytes = 48;   // a window in LDS: 6 rows of 8 bytes (5 x 5 cells are filled; a walk that asks for more is reported)
constexpr int kWinCells = 5;
constexpr int kPatchRows = kWinCells + 6;                   // image rows oy - 3 .. oy + 7, 16 bytes from column ox - 3
constexpr int kWinStride = 2 * kWinBytes + 4;               // 100 bytes = 25 dwords per thread: odd, neighbouring threads on different banks
// a walker's record, one 64-byte line: dwords 0..3 the 4 x 4 cells of the window above (a row each), 4..8 the first four
// cells of the five rows of the window below, 9..10 their fifth cells (a byte each), 11..15 the own 3 x 3 patch (nine bytes,
// first index x) and behind it, on layer 0, the nine 5/8 scores of the guessed layer below
constexpr int kRecBytes = 64, kRecBelowDw = 4, kRecBelowHiDw = 9, kRecOwnDw = 11;
struct Window {
    uint8_t *cells;
    int ox, oy, layer, side;  // side: cells per row and column that are filled (4 above, 5 below: window_place)
    unsigned long long asked;  // bit iy * kWinSide + ix: the walk asked for this cell (and the cell is inside the scored region)
    bool escaped;              // the walk left the window (cannot happen by construction; reported if it does)
};

// kPatchRows image rows of 16 bytes from (x0, y0) of a layer, in registers: byte c of row r = image (x0 + c, y0 + r).
// Addresses are clamped into the layer: what a clamp brings in stands for positions outside the image, which no
// in-region cell's ring touches (a start clamped at the left border is shifted back into place).
struct Patch {
    uint32_t w[kPatchRows][4];
};
__device__ __forceinline__ Patch patch_fetch(const uint8_t *img, const DetLayer &L, int x0, int y0)
{
    Patch q;
    const int xs = min(max(x0, 0), max(L.w - 1, 0));
    const int shift = xs - x0;  // 0 but for windows at the left border (1..3 there)
#pragma unroll
    for (int r = 0; r < kPatchRows; ++r) {  // unaligned 16-byte loads, all in flight together; the layer planes are padded by 64 bytes
        const int yc = min(max(y0 + r, 0), max(L.h - 1, 0));
        uint4 v;
        __builtin_memcpy(&v, img + L.off + (int64_t)yc * L.w + xs, 16);
        q.w[r][0] = v.x;
        q.w[r][1] = v.y;
        q.w[r][2] = v.z;
        q.w[r][3] = v.w;
    }
    if (shift > 0) {
        const int sh = 8 * min(shift, 3);
#pragma unroll
        for (int r = 0; r < kPatchRows; ++r) {
            q.w[r][3] = (q.w[r][3] << sh) | (q.w[r][2] >> (32 - sh));
            q.w[r][2] = (q.w[r][2] << sh) | (q.w[r][1] >> (32 - sh));
            q.w[r][1] = (q.w[r][1] << sh) | (q.w[r][0] >> (32 - sh));
            q.w[r][0] = q.w[r][0] << sh;
        }
    }
    return q;
}
__device__ __forceinline__ int patch_byte(const Patch &q, int r, int c) { return (int)((q.w[r][c >> 2] >> (8 * (c & 3))) & 0xff); }

// the OAST 9/16 score of the image byte at patch position (r, c) -- r, c compile-time constants after unrolling
__device__ __forceinline__ int patch_ring_score(const Patch &q, int r, int c)
{
    constexpr int dx[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    constexpr int dy[16] = {0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1};  // OastDetector9_16::init_pattern (oast9_16.h:74-92)
    int p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = patch_byte(q, r + dy[k], c + dx[k]);
    const int ctr = patch_byte(q, r, c);
    int arc_lo, arc_hi;
    arc9_extremes(p, arc_lo, arc_hi);
    return max(max(arc_lo - ctr, ctr - arc_hi) - 1, 0);
}

// AgastDetector5_8::cornerScore from b = 0 (agast5_8_nms.cc:42; brisk.cpp:1696-1703): 5 contiguous of the 8 neighbours of
// the image byte at patch position (r, c), neighbours in init_pattern order (agast5_8.h:66-76).  For a candidate, which
// lies at least 3 pixels inside the layer, none of the nine positions asked for touches the 2-pixel border where
// BriskLayer::getAgastScore_5_8 returns 0.
__device__ __forceinline__ int patch_score_5_8(const Patch &q, int r, int c)
{
    constexpr int dx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    const int v = patch_byte(q, r, c);
    int d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = patch_byte(q, r + dy[k], c + dx[k]) - v;
    int vb = -256, vd = -256;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int mn = d[k], mx = d[k];
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            mn = min(mn, d[(k + j) & 7]);
            mx = max(mx, d[(k + j) & 7]);
        }
        vb = max(vb, mn);
        vd = max(vd, -mx);
    }
    return max(max(vb, vd) - 1, 0);
}

template <bool MARK>
__device__ __forceinline__ int window_at(const PairView &v, Window &w, int x, int y)
{
    const int ix = x - w.ox, iy = y - w.oy;
    if ((unsigned)ix >= (unsigned)w.side || (unsigned)iy >= (unsigned)w.side) {
        w.escaped = true;
        return 0;
    }
    if (MARK) {
        const DetLayer &L = v.g->L[w.layer];
        if (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) w.asked |= 1ull << (iy * kWinSide + ix);
    }
    return w.cells[iy * kWinRow + ix];
}

// The cells a walk asked for in the layer above: their true scores go into that layer's score plane right away (a
// cell's score is what it is, whoever computes it and whenever; nobody reads a cell below the threshold unless a touch or
// status byte says the reference would have cached it).  The window's rows go out whole -- four scores per row, one
// unaligned 4-byte store -- for every row the walk asked a cell of: the cells beside the asked ones get their true scores
// too, which nobody minds (cells outside the scored region hold 0 in the window and in the plane).
__device__ __forceinline__ void publish_scores(const PairView &v, const Window &w)
{
    const DetLayer &L = v.g->L[w.layer];
#pragma unroll
    for (int iy = 0; iy < kWinCells - 1; ++iy) {  // (windows above are 4 x 4)
        if (!((w.asked >> (iy * kWinSide)) & 0x3f)) continue;
        const uint32_t row = *reinterpret_cast<const uint32_t *>(w.cells + iy * kWinRow);
        __builtin_memcpy(v.score_rw + L.off + (int64_t)(w.oy + iy) * L.w + w.ox, &row, 4);
    }
}

// ... and they become "cached" there -- touch -- once the walker is known to be a maximum.
__device__ __forceinline__ void apply_asked(const PairView &v, int layer_above, int ox, int oy, unsigned long long asked)
{
    const DetLayer &L = v.g->L[layer_above];
    while (asked) {
        const int k = __ffsll((long long)asked) - 1;
        asked &= asked - 1;
        v.touch[L.off + (int64_t)(oy + k / kWinSide) * L.w + ox + k % kWinSide] = 1;
    }
}

// BriskScaleSpace::subpixel2D (brisk.cpp:1535-1644); s = s_0_0, s_0_1, s_0_2, s_1_0, ... (first index x)
template <class FP>
__device__ __forceinline__ float subpixel2d(const FP fp, const int (&s)[9], float &delta_x, float &delta_y)
{
    const int s_0_0 = s[0], s_0_1 = s[1], s_0_2 = s[2], s_1_0 = s[3], s_1_1 = s[4], s_1_2 = s[5], s_2_0 = s[6], s_2_1 = s[7],
              s_2_2 = s[8];
    const int tmp1 = s_0_0 + s_0_2 - 2 * s_1_1 + s_2_0 + s_2_2;
    const int coeff1 = 3 * (tmp1 + s_0_1 - ((s_1_0 + s_1_2) * 2) + s_2_1);
    const int coeff2 = 3 * (tmp1 - ((s_0_1 + s_2_1) * 2) + s_1_0 + s_1_2);
    const int tmp2 = s_0_2 - s_2_0;
    const int tmp3 = (s_0_0 + tmp2 - s_2_2);
    const int tmp4 = tmp3 - 2 * tmp2;
    const int coeff3 = -3 * (tmp3 + s_0_1 - s_2_1);
    const int coeff4 = -3 * (tmp4 + s_1_0 - s_1_2);
    const int coeff5 = (s_0_0 - s_0_2 - s_2_0 + s_2_2) * 4;
    const int coeff6 = (-(s_0_0 + s_0_2 - ((s_1_0 + s_0_1 + s_1_2 + s_2_1) * 2) - 5 * s_1_1 + s_2_0 + s_2_2)) * 2;
    const int H_det = 4 * coeff1 * coeff2 - coeff5 * coeff5;
    if (H_det == 0) {
        delta_x = 0.0f;
        delta_y = 0.0f;
        return (float)((double)(float)coeff6 / 18.0);
    }
    if (!(H_det > 0 && coeff1 < 0)) {  // the maximum is at one of the four patch corners
        int tmp_max = coeff3 + coeff4 + coeff5;
        delta_x = 1.0f;
        delta_y = 1.0f;
        int tmp = -coeff3 + coeff4 - coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            delta_x = -1.0f;
            delta_y = 1.0f;
        }
        tmp = coeff3 - coeff4 - coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            delta_x = 1.0f;
            delta_y = -1.0f;
        }
        tmp = -coeff3 - coeff4 + coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            delta_x = -1.0f;
            delta_y = -1.0f;
        }
        return (float)((double)(float)(tmp_max + coeff1 + coeff2 + coeff6) / 18.0);
    }
    const float dx = (float)fp.div((float)(2 * coeff2 * coeff3 - coeff4 * coeff5), (float)(-H_det));
    const float dy = (float)fp.div((float)(2 * coeff1 * coeff4 - coeff3 * coeff5), (float)(-H_det));
    bool tx = false, tx_ = false, ty = false, ty_ = false;
    if ((double)dx > 1.0)
        tx = true;
    else if ((double)dx < -1.0)
        tx_ = true;
    if ((double)dy > 1.0) ty = true;
    if ((double)dy < -1.0) ty_ = true;
    const float c1 = (float)coeff1, c2 = (float)coeff2, c3 = (float)coeff3, c4 = (float)coeff4, c5 = (float)coeff5, c6 = (float)coeff6;
    // c1*x*x + c2*y*y + c3*x + c4*y + c5*x*y + c6: products and sums of floats, left to right (then divided in double)
    auto quad = [&](float x, float y) -> double {
        typename FP::R v = fp.mul(fp.mul(c1, x), x);
        v = fp.add(v, fp.mul(fp.mul(c2, y), y));
        v = fp.add(v, fp.mul(c3, x));
        v = fp.add(v, fp.mul(c4, y));
        v = fp.add(v, fp.mul(fp.mul(c5, x), y));
        return (double)fp.add(v, c6);
    };
    if (tx || tx_ || ty || ty_) {
        float dx1 = 0.0f, dx2 = 0.0f, dy1 = 0.0f, dy2 = 0.0f;
        if (tx) {
            dx1 = 1.0f;
            dy1 = (float)fp.div(-(float)(coeff4 + coeff5), (float)(2 * coeff2));
            if ((double)dy1 > 1.0) dy1 = 1.0f; else if ((double)dy1 < -1.0) dy1 = -1.0f;
        } else if (tx_) {
            dx1 = -1.0f;
            dy1 = (float)fp.div(-(float)(coeff4 - coeff5), (float)(2 * coeff2));
            if ((double)dy1 > 1.0) dy1 = 1.0f; else if ((double)dy1 < -1.0) dy1 = -1.0f;
        }
        if (ty) {
            dy2 = 1.0f;
            dx2 = (float)fp.div(-(float)(coeff3 + coeff5), (float)(2 * coeff1));
            if ((double)dx2 > 1.0) dx2 = 1.0f; else if ((double)dx2 < -1.0) dx2 = -1.0f;
        } else if (ty_) {
            dy2 = -1.0f;
            dx2 = (float)fp.div(-(float)(coeff3 - coeff5), (float)(2 * coeff1));
            if ((double)dx2 > 1.0) dx2 = 1.0f; else if ((double)dx2 < -1.0) dx2 = -1.0f;
        }
        const float max1 = (float)(quad(dx1, dy1) / 18.0);
        const float max2 = (float)(quad(dx2, dy2) / 18.0);
        if (max1 > max2) {
            delta_x = dx1;
            delta_y = dx1;  // sic (:1629)
            return max1;
        }
        delta_x = dx2;
        delta_y = dx2;  // sic (:1634)
        return max2;
    }
    delta_x = dx;
    delta_y = dy;
    return (float)(quad(dx, dy) / 18.0);
}

// refine1D (variant 0, :1418), refine1D_1 (1, :1459), refine1D_2 (2, :1499)
template <class FP>
__device__ __forceinline__ float refine1d(const FP fp, int variant, float s_05, float s0, float s05, float &max)
{
    const int i_05 = (int)(1024.0 * (double)s_05 + 0.5);
    const int i0 = (int)(1024.0 * (double)s0 + 0.5);
    const int i05 = (int)(1024.0 * (double)s05 + 0.5);
    int qa, qb, qc;
    double lo_d, hi_d;
    if (variant == 0) {
        qa = 16 * i_05 - 24 * i0 + 8 * i05;
        qb = -40 * i_05 + 54 * i0 - 14 * i05;
        qc = +24 * i_05 - 27 * i0 + 6 * i05;
        lo_d = 0.75;
        hi_d = 1.5;
    } else if (variant == 1) {
        qa = 9 * i_05 - 18 * i0 + 9 * i05;
        qb = -21 * i_05 + 36 * i0 - 15 * i05;
        qc = +12 * i_05 - 16 * i0 + 6 * i05;
        lo_d = 0.6666666666666666666666666667;
        hi_d = 1.33333333333333333333333333;
    } else {
        qa = 2 * i_05 - 4 * i0 + 2 * i05;
        qb = -5 * i_05 + 8 * i0 - 3 * i05;
        qc = +3 * i_05 - 3 * i0 + 1 * i05;
        lo_d = 0.7;
        hi_d = 1.5;
    }
    if (qa >= 0) {  // second derivative must be negative
        if (s0 >= s_05 && s0 >= s05) {
            max = s0;
            return 1.0f;
        }
        if (s_05 >= s0 && s_05 >= s05) {
            max = s_05;
            return (float)lo_d;
        }
        if (s05 >= s0 && s05 >= s_05) {
            max = s05;
            return (float)(variant == 1 ? 1.3333333333333333333333333333 : 1.5);
        }
    }
    float ret_val = (float)fp.div(-(float)qb, (float)(2 * qa));
    if ((double)ret_val < lo_d)
        ret_val = (float)lo_d;
    else if ((double)ret_val > hi_d)
        ret_val = (float)hi_d;
    float m = (float)fp.add(fp.add((float)qc, fp.mul(fp.mul((float)qa, ret_val), ret_val)), fp.mul((float)qb, ret_val));
    if (variant == 2)
        m = m / 1024.0f;
    else
        m = (float)((double)m / (variant == 0 ? 3072.0 : 2048.0));
    max = m;
    return ret_val;
}

// the sampling square of getScoreMaxAbove (ABOVE, brisk.cpp:1123-1133) / getScoreMaxBelow (:1269-1281) in the neighbouring layer
template <bool ABOVE>
__device__ __forceinline__ void walk_square(int layer, int x_layer, int y_layer, float &x_1, float &x1, float &y_1, float &y1)
{
    const bool octave = (layer & 1) == 0;
    if (ABOVE) {
        if (octave) {  // double division (:1123-1126)
            x_1 = (float)((double)(float)(4 * x_layer - 1 - 2) / 6.0);
            x1 = (float)((double)(float)(4 * x_layer - 1 + 2) / 6.0);
            y_1 = (float)((double)(float)(4 * y_layer - 1 - 2) / 6.0);
            y1 = (float)((double)(float)(4 * y_layer - 1 + 2) / 6.0);
        } else {  // float division (:1130-1133)
            x_1 = (float)(6 * x_layer - 1 - 3) / 8.0f;
            x1 = (float)(6 * x_layer - 1 + 3) / 8.0f;
            y_1 = (float)(6 * y_layer - 1 - 3) / 8.0f;
            y1 = (float)(6 * y_layer - 1 + 3) / 8.0f;
        }
    } else {
        if (octave) {
            x_1 = (float)((double)(float)(8 * x_layer + 1 - 4) / 6.0);
            x1 = (float)((double)(float)(8 * x_layer + 1 + 4) / 6.0);
            y_1 = (float)((double)(float)(8 * y_layer + 1 - 4) / 6.0);
            y1 = (float)((double)(float)(8 * y_layer + 1 + 4) / 6.0);
        } else {
            x_1 = (float)((double)(float)(6 * x_layer + 1 - 3) / 4.0);
            x1 = (float)((double)(float)(6 * x_layer + 1 + 3) / 4.0);
            y_1 = (float)((double)(float)(6 * y_layer + 1 - 3) / 4.0);
            y1 = (float)((double)(float)(6 * y_layer + 1 + 3) / 4.0);
        }
    }
}

// where the walk's window sits: it reaches from (int)x_1 - 1 (patch around max_x = (int)x1 when that equals (int)x_1)
// to (int)x1 + 1.  The sampling square is 2/3 or 3/4 of a cell wide in the layer above ((int)x1 <= (int)x_1 + 1: four
// cells) and 4/3 or 3/2 cells wide in the layer below ((int)x1 <= (int)x_1 + 2: five cells); the window holds six
template <bool ABOVE>
__device__ __forceinline__ void window_place(Window &win, int layer, int x_layer, int y_layer)
{
    float x_1, x1, y_1, y1;
    walk_square<ABOVE>(layer, x_layer, y_layer, x_1, x1, y_1, y1);
    win.ox = (int)x_1 - 1;
    win.oy = (int)y_1 - 1;
    win.layer = ABOVE ? layer + 1 : layer - 1;
    win.side = ABOVE ? kWinCells - 1 : kWinCells;
    win.asked = 0;
    win.escaped = false;
}

// getScoreMaxAbove (ABOVE, brisk.cpp:1106-1249) / getScoreMaxBelow (:1251-1416); the window is in place and filled
template <bool ABOVE, class FP>
__device__ __forceinline__ float neighbour_layer_max(const PairView &v, Window &win, int layer, int x_layer, int y_layer, int threshold, bool &ismax, float &dx,
                                                     float &dy)
{
    ismax = false;
    const bool octave = (layer & 1) == 0;
    float x_1, x1, y_1, y1;
    walk_square<ABOVE>(layer, x_layer, y_layer, x_1, x1, y_1, y1);
    const float thr = (float)threshold;
    const FP fp{};
    const int xa = (int)fp.add(x_1, 1.0f), xb = (int)x1, ya = (int)fp.add(y_1, 1.0f), yb = (int)y1;
    auto S = [&](int x, int y) { return window_at<ABOVE>(v, win, x, y); };  // getAgastScore(int, int, 1)
    auto Q = [&](int x, int y) { return window_at<false>(v, win, x, y); };  // same, for the layer below (no bookkeeping)
    auto F = [&](float xf, float yf) {                                      // getAgastScore(float, float, 1): bilinear through uint8_t
        const int x = (int)xf;
        const float rx1 = xf - (float)x;
        const float rx = 1.0f - rx1;
        const int y = (int)yf;
        const float ry1 = yf - (float)y;
        const float ry = 1.0f - ry1;
        const float s00 = (float)S(x, y), s10 = (float)S(x + 1, y), s01 = (float)S(x, y + 1), s11 = (float)S(x + 1, y + 1);
        const typename FP::R r = fp.add(fp.add(fp.add(fp.mul(fp.mul(rx, ry), s00), fp.mul(fp.mul(rx1, ry), s10)), fp.mul(fp.mul(rx, ry1), s01)),
                                        fp.mul(fp.mul(rx1, ry1), s11));  // one expression, converted to uint8_t
        return (int)r & 0xff;
    };

    // first row
    int max_x = xa, max_y = ya;
    float tmp_max;
    float max = (float)F(x_1, y_1);
    if (max > thr) return 0;
    for (int x = xa; x <= xb; x++) {
        tmp_max = (float)F((float)x, y_1);
        if (tmp_max > thr) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = x;
        }
    }
    tmp_max = (float)F(x1, y_1);
    if (tmp_max > thr) return 0;
    if (tmp_max > max) {
        max = tmp_max;
        max_x = xb;
    }
    // middle rows
    for (int y = ya; y <= yb; y++) {
        tmp_max = (float)F(x_1, (float)y);
        if (tmp_max > thr) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = xa;
            max_y = y;
        }
        for (int x = xa; x <= xb; x++) {
            tmp_max = (float)S(x, y);
            if (tmp_max > thr) return 0;
            if (!ABOVE && tmp_max == max) {  // :1321-1344 (below only)
                const int t1 = 2 * (Q(x - 1, y) + Q(x + 1, y) + Q(x, y + 1) +
                                    Q(x, y - 1)) +
                               (Q(x + 1, y + 1) + Q(x - 1, y + 1) +
                                Q(x + 1, y - 1) + Q(x - 1, y - 1));
                const int t2 = 2 * (Q(max_x - 1, max_y) + Q(max_x + 1, max_y) +
                                    Q(max_x, max_y + 1) + Q(max_x, max_y - 1)) +
                               (Q(max_x + 1, max_y + 1) + Q(max_x - 1, max_y + 1) +
                                Q(max_x + 1, max_y - 1) + Q(max_x - 1, max_y - 1));
                if (t1 > t2) {
                    max_x = x;
                    max_y = y;
                }
            }
            if (tmp_max > max) {
                max = tmp_max;
                max_x = x;
                max_y = y;
            }
        }
        tmp_max = (float)F(x1, (float)y);
        if (tmp_max > thr) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = xb;
            max_y = y;
        }
    }
    // bottom row: no early exit
    tmp_max = (float)F(x_1, y1);
    if (tmp_max > max) {
        max = tmp_max;
        max_x = xa;
        max_y = yb;
    }
    for (int x = xa; x <= xb; x++) {
        tmp_max = (float)F((float)x, y1);
        if (tmp_max > max) {
            max = tmp_max;
            max_x = x;
            max_y = yb;
        }
    }
    tmp_max = (float)F(x1, y1);
    if (tmp_max > max) {
        max = tmp_max;
        max_x = xb;
        max_y = yb;
    }

    float dx_1, dy_1;
    int patch[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) patch[k] = S(max_x + k / 3 - 1, max_y + k % 3 - 1);  // s_0_0, s_0_1, s_0_2, s_1_0, ... (first index x)
    const float refined_max = subpixel2d(fp, patch, dx_1, dy_1);
    const float real_x = (float)fp.add((float)max_x, dx_1);
    const float real_y = (float)fp.add((float)max_y, dy_1);
    bool returnrefined = true;
    if (ABOVE) {
        if (octave) {  // float arithmetic (:1228-1229)
            dx = (float)fp.sub(fp.div(fp.add(fp.mul(real_x, 6.0f), 1.0f), 4.0f), (float)x_layer);
            dy = (float)fp.sub(fp.div(fp.add(fp.mul(real_y, 6.0f), 1.0f), 4.0f), (float)y_layer);
        } else {  // double arithmetic (:1232-1233)
            dx = (float)(((double)real_x * 8.0 + 1.0) / 6.0 - (double)(float)x_layer);
            dy = (float)(((double)real_y * 8.0 + 1.0) / 6.0 - (double)(float)y_layer);
        }
    } else {
        if (octave) {
            dx = (float)(((double)real_x * 6.0 + 1.0) / 8.0 - (double)(float)x_layer);
            dy = (float)(((double)real_y * 6.0 + 1.0) / 8.0 - (double)(float)y_layer);
        } else {
            dx = (float)(((double)real_x * 4.0 - 1.0) / 6.0 - (double)(float)x_layer);
            dy = (float)(((double)real_y * 4.0 - 1.0) / 6.0 - (double)(float)y_layer);
        }
    }
    if (dx > 1.0f) { dx = 1.0f; returnrefined = false; }
    if (dx < -1.0f) { dx = -1.0f; returnrefined = false; }
    if (dy > 1.0f) { dy = 1.0f; returnrefined = false; }
    if (dy < -1.0f) { dy = -1.0f; returnrefined = false; }
    ismax = true;
    if (returnrefined) return refined_max < max ? max : refined_max;  // std::max(refined_max, max)
    return max;
}

struct Refined {
    bool emit, reached;  // reached: the walk got as far as the 3x3 patch on its own layer (those cells are cached from then on)
    bool escaped;
    DetResult r;
    unsigned long long asked;  // cells of the layer above the walk asked for, relative to (ox, oy)
    int ox, oy;
};

// What getKeypoints does with one 2-D maximum (brisk.cpp:609-702), refine3D included (:937-1103).
template <class FP>
__device__ __forceinline__ Refined refine_maximum(const PairView &v, uint8_t *lds_cells, const int (&own_patch)[9], const int (&s58)[9], int layer, int px, int py,
                                                  int threshold)
{
    // One walk above, one below, one own patch -- in the reference's order (above, below, patch), each at a single
    // call site so that everything inlines and no argument goes through the stack.  The windows' cells (lds_cells), the
    // own patch (first index x: getAgastScore(int, int, 1) on the own layer, :1685-1694) and the 5/8 scores come from the
    // walker's record (det_window_kernel).
    const float basicSize = 12.0f;
    const FP fp{};
    const DetGeom &g = *v.g;
    const DetLayer &L = g.L[layer];
    Refined out;
    out.emit = false;
    out.reached = false;
    out.r = DetResult{0.f, 0.f, 0.f, 0.f};
    out.asked = 0;
    out.ox = out.oy = 0;
    out.escaped = false;
    const bool single = g.n_layers == 1, last = layer == g.n_layers - 1, octave = (layer & 1) == 0;
    Window wa, wb;
    wa.cells = lds_cells;
    wb.cells = lds_cells + kWinBytes;
    if (!last) window_place<true>(wa, layer, px, py);
    if (layer > 0) window_place<false>(wb, layer, px, py);
    const int center = own_patch[4];
    bool ismax = true;
    float max_above = 0.f, max_below = 0.f;
    float delta_x_above = 0.f, delta_y_above = 0.f, delta_x_below = 0.f, delta_y_below = 0.f, delta_x_layer, delta_y_layer;
    if (!last) {  // refine3D: getScoreMaxAbove first (:945-950)
        max_above = neighbour_layer_max<true, FP>(v, wa, layer, px, py, center, ismax, delta_x_above, delta_y_above);
        out.asked = wa.asked;
        out.ox = wa.ox;
        out.oy = wa.oy;
        out.escaped = wa.escaped;
        publish_scores(v, wa);
        if (!ismax) return out;
    }
    if (layer > 0) {  // getScoreMaxBelow: the last layer (:651-657), octaves above 0 (:991-996), intra layers (:1049-1053)
        max_below = neighbour_layer_max<false, FP>(v, wb, layer, px, py, center, ismax, delta_x_below, delta_y_below);
        out.escaped |= wb.escaped;
        if (!ismax) return out;
    } else if (!single) {  // layer 0: guess the missing layer below with the 5/8 mask (:959-989)
        int mb = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) mb = max(mb, s58[k]);
        (void)subpixel2d(fp, s58, delta_x_below, delta_y_below);
        max_below = (float)mb;
    }
    const float max_layer = subpixel2d(fp, own_patch, delta_x_layer, delta_y_layer);
    out.reached = true;
    // the own 3 x 3 patch is in the reference's cache from here on (status says so once the candidate is a maximum): its scores
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int x = px + k / 3 - 1, y = py + k % 3 - 1;
        if (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) v.score_rw[L.off + (int64_t)y * L.w + x] = (uint8_t)own_patch[k];
    }
    if (single) {  // :609-638
        out.emit = true;
        out.r = DetResult{(float)fp.add((float)px, delta_x_layer), (float)fp.add((float)py, delta_y_layer), basicSize, max_layer};
        return out;
    }
    if (last) {  // :659-678
        out.emit = true;
        out.r = DetResult{(float)fp.add(fp.mul(fp.add((float)px, delta_x_layer), L.scale), L.offset),
                          (float)fp.add(fp.mul(fp.add((float)py, delta_y_layer), L.scale), L.offset), basicSize * L.scale, max_layer};
        return out;
    }
    const float s0 = ((float)center < max_layer) ? max_layer : (float)center;  // std::max(float(center), max_layer)
    float best;
    float scale = refine1d(fp, octave ? (layer == 0 ? 2 : 0) : 1, max_below, s0, max_above, best);
    float r0, r1;
    bool up;
    if (octave) {
        up = (double)scale > 1.0;
        if (up)
            r0 = (float)((1.5 - (double)scale) / .5);  // :1019
        else if (layer == 0)
            r0 = (float)(((double)scale - 0.5) / 0.5);  // :1029
        else
            r0 = (float)(((double)scale - 0.75) / 0.25);  // :1036
    } else {
        up = (double)scale > 1.0;
        r0 = up ? (float)(4.0 - (double)scale * 3.0) : (float)((double)scale * 3.0 - 2.0);  // :1076, :1085
    }
    r1 = (float)(1.0 - (double)r0);
    const float ox = up ? delta_x_above : delta_x_below, oy = up ? delta_y_above : delta_y_below;
    // (r0 * delta_layer + r1 * delta_other + float(c)) [* scale + offset]: one expression each in the reference
    typename FP::R xe = fp.add(fp.add(fp.mul(r0, delta_x_layer), fp.mul(r1, ox)), (float)px);
    typename FP::R ye = fp.add(fp.add(fp.mul(r0, delta_y_layer), fp.mul(r1, oy)), (float)py);
    if (up || layer != 0) {  // layer 0 interpolating towards the guessed layer below stays in image coordinates (:1031-1032)
        xe = fp.add(fp.mul(xe, L.scale), L.offset);
        ye = fp.add(fp.mul(ye, L.scale), L.offset);
    }
    const float x = (float)xe, y = (float)ye;
    scale *= L.scale;
    if (best > (float)threshold) {  // :698
        out.emit = true;
        out.r = DetResult{x, y, basicSize * scale, best};
    }
    return out;
}


// Refinement of candidate i.  It reads nothing but its record, so it does not depend on whether the candidate's tie (if it
// has one) is already decided: SPECULATIVE runs it ahead of the decision and parks what the decision will publish --
// result, "reached its patch", and the cells it asked for in the layer above.
template <class FP>
__device__ __forceinline__ void finish_candidate(const DetArgs &a, const PairView &v, uint8_t *lds_cells, const int (&own_patch)[9], const int (&s58)[9], int p,
                                                 int i, int layer, int x, int y, bool SPECULATIVE)
{
    const Refined r = refine_maximum<FP>(v, lds_cells, own_patch, s58, layer, x, y, a.threshold);
    const int64_t ci = (int64_t)p * a.cand_cap + i;
    a.cand_res[ci] = r.r;
    if (r.escaped) atomicOr(a.status_word, 16);
    a.cand_asked[ci] = r.asked;  // (also what the clean-up behind the emission takes back from the touch map)
    a.cand_win[ci] = (uint32_t)r.ox | (uint32_t)r.oy << 16;
    if (SPECULATIVE) {
        a.cand_emit[ci] = 0;
        a.cand_spec[ci] = (uint8_t)(kWasTie | (r.emit ? kEmit : 0) | (r.reached ? kReached : 0));
    } else {
        a.cand_emit[ci] = r.emit ? 1 : 0;
        const DetLayer &L = v.g->L[layer];
        v.status[L.off + (int64_t)y * L.w + x] = r.reached ? kStReached : kStDone;
        if (r.asked) apply_asked(v, layer + 1, r.ox, r.oy, r.asked);
    }
}

// maxima without ties: independent of everything else; ties: refined ahead of their decision
constexpr int kRefineChunk = 512;  // candidates per workgroup

// the SIDE x SIDE cells of window w from the image bytes behind it: the first four cells of a row in lo, the fifth in hi
template <int SIDE>
__device__ __forceinline__ void window_scores(const Patch &q, const DetLayer &L, const Window &w, uint32_t (&lo)[SIDE], uint32_t (&hi)[SIDE])
{
#pragma unroll
    for (int iy = 0; iy < SIDE; ++iy) {
        const int y = w.oy + iy;
        const bool row_in = y >= 3 && y < L.h - 3;
        lo[iy] = 0;
        hi[iy] = 0;
#pragma unroll
        for (int ix = 0; ix < SIDE; ++ix) {
            const int x = w.ox + ix;
            const int sc = (row_in && x >= 3 && x < L.w - 3) ? patch_ring_score(q, iy + 3, ix + 3) : 0;
            if (ix < 4)
                lo[iy] |= (uint32_t)sc << (8 * ix);
            else
                hi[iy] = (uint32_t)sc;
        }
    }
}

// The cells behind the walks.  A workgroup takes a chunk of candidates, gathers its maxima and ties ("walkers": a quarter
// or so of the candidates; the list goes to global memory for det_walk_kernel) and hands out (walker, window) items --
// window 0: the 5 x 5 cells of the layer above, 1: of the layer below, 2: the own 3 x 3 patch and, on layer 0, the 5/8
// scores that stand in for the layer below (:959-989).  A thread fetches the 11 x 11 image bytes behind its window into
// registers (eleven 16-byte loads in flight together) and scores the cells from there: no LDS, few registers, many waves.
__global__ __launch_bounds__(kDetThreads) void det_window_kernel(DetArgs a)
{
    __shared__ int todo[kRefineChunk], wave_cnt[16], n_todo, ls_s[kDetMaxLayers + 1];
    __shared__ DetGeom geom_s;
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n_layers = a.dg->n_layers, n = ls[n_layers], c0 = blockIdx.x * kRefineChunk;
    if (c0 >= n) return;
    if (threadIdx.x <= kDetMaxLayers) ls_s[threadIdx.x] = (int)threadIdx.x <= n_layers ? ls[threadIdx.x] : 0x7fffffff;
    geom_to_lds(&geom_s, a.dg);
    const int64_t cb = (int64_t)p * a.cand_cap;
    {  // the chunk's walkers, in candidate order: both halves of the chunk in one go; and the ties among them, with their
       // positions, as a list of their own for det_tie_kernel
        static_assert(kRefineChunk == 2 * kDetThreads, "a thread takes two candidates of the chunk");
        const int i_a = c0 + threadIdx.x, i_b = i_a + kDetThreads;
        const uint8_t f_a = a.cand_flag[cb + min(i_a, n - 1)], f_b = a.cand_flag[cb + min(i_b, n - 1)];
        const uint32_t xy_a = a.cand_xy[cb + min(i_a, n - 1)], xy_b = a.cand_xy[cb + min(i_b, n - 1)];
        const bool take_a = i_a < n && f_a != kDetNotMax, take_b = i_b < n && f_b != kDetNotMax;
        const bool tie_a = i_a < n && f_a == kDetTie, tie_b = i_b < n && f_b == kDetTie;
        const unsigned long long m_a = __ballot(take_a), m_b = __ballot(take_b), t_a = __ballot(tie_a), t_b = __ballot(tie_b);
        if (lane == 0) {
            wave_cnt[wave] = __popcll(m_a);
            wave_cnt[4 + wave] = __popcll(m_b);
            wave_cnt[8 + wave] = __popcll(t_a);
            wave_cnt[12 + wave] = __popcll(t_b);
        }
        __syncthreads();
        int before_a = 0, before_b = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        int tbefore_a = 0, tbefore_b = wave_cnt[8] + wave_cnt[9] + wave_cnt[10] + wave_cnt[11];
        for (int w = 0; w < wave; ++w) {
            before_a += wave_cnt[w];
            before_b += wave_cnt[4 + w];
            tbefore_a += wave_cnt[8 + w];
            tbefore_b += wave_cnt[12 + w];
        }
        const unsigned long long below = (1ull << lane) - 1;
        if (take_a) {
            const int k = before_a + __popcll(m_a & below);
            todo[k] = i_a;
            a.walk_list[cb + c0 + k] = i_a;
        }
        if (take_b) {
            const int k = before_b + __popcll(m_b & below);
            todo[k] = i_b;
            a.walk_list[cb + c0 + k] = i_b;
        }
        if (tie_a) a.tie_list[cb + c0 + tbefore_a + __popcll(t_a & below)] = DetTie{i_a, xy_a};
        if (tie_b) a.tie_list[cb + c0 + tbefore_b + __popcll(t_b & below)] = DetTie{i_b, xy_b};
        if (threadIdx.x == 0) {
            n_todo = before_b + wave_cnt[4] + wave_cnt[5] + wave_cnt[6] + wave_cnt[7];
            a.tie_count[(int64_t)p * a.walk_chunks + blockIdx.x] = tbefore_b + wave_cnt[12] + wave_cnt[13] + wave_cnt[14] + wave_cnt[15];
        }
        __syncthreads();
    }
    const int nt = n_todo;
    if (threadIdx.x == 0) a.walk_count[(int64_t)p * a.walk_chunks + blockIdx.x] = nt;
    PairView v = pair_view(a, p);
    v.g = &geom_s;
    const bool single = n_layers == 1;
    for (int it = threadIdx.x; it < 3 * nt; it += kDetThreads) {
        const int part = it / nt, k = it - part * nt;  // part-major: a wave's threads mostly share the window kind
        const int i = todo[k];
        const uint32_t xy = a.cand_xy[cb + i];
        int layer = 0;
#pragma unroll
        for (int l = 1; l < kDetMaxLayers; ++l) layer += i >= ls_s[l] ? 1 : 0;
        const int px = (int)(xy & 0xffff), py = (int)(xy >> 16);
        uint8_t *rec = a.cand_cells + (cb + c0 + k) * kRecBytes;
        if (part == 2) {
            // the candidate's own surroundings: 11 x 11 bytes centred on it (a candidate lies >= 3 pixels inside: px - 5 >= -2)
            const DetLayer &L = v.g->L[layer];
            const Patch q = patch_fetch(v.img, L, px - 5, py - 5);
            uint32_t o[3] = {0, 0, 0}, f[3] = {0, 0, 0};
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int dx = c / 3 - 1, dy = c % 3 - 1, x = px + dx, y = py + dy;
                const int sc = (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) ? patch_ring_score(q, 5 + dy, 5 + dx) : 0;
                o[c >> 2] |= (uint32_t)sc << (8 * (c & 3));
            }
            if (layer == 0 && !single) {
#pragma unroll
                for (int c = 0; c < 9; ++c) f[c >> 2] |= (uint32_t)patch_score_5_8(q, 5 + c % 3 - 1, 5 + c / 3 - 1) << (8 * (c & 3));
            }
            uint32_t *d = reinterpret_cast<uint32_t *>(rec) + kRecOwnDw;  // bytes 0..8: the patch, 9..17: the 5/8 scores
            d[0] = o[0];
            d[1] = o[1];
            d[2] = (o[2] & 0xffu) | f[0] << 8;
            d[3] = f[0] >> 24 | f[1] << 8;
            d[4] = f[1] >> 24 | (f[2] & 0xffu) << 8;
            continue;
        }
        const bool above = part == 0;
        if (above ? layer == n_layers - 1 : layer == 0) continue;  // no such layer (layer 0's guessed layer below: window 2)
        Window w;
        if (above)
            window_place<true>(w, layer, px, py);
        else
            window_place<false>(w, layer, px, py);
        const DetLayer &L = v.g->L[w.layer];
        const Patch q = patch_fetch(v.img, L, w.ox - 3, w.oy - 3);
        uint32_t *d = reinterpret_cast<uint32_t *>(rec);
        if (above) {
            uint32_t lo[kWinCells - 1], hi[kWinCells - 1];
            window_scores<kWinCells - 1>(q, L, w, lo, hi);
            *reinterpret_cast<uint4 *>(d) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        } else {
            uint32_t lo[kWinCells], hi[kWinCells];
            window_scores<kWinCells>(q, L, w, lo, hi);
#pragma unroll
            for (int iy = 0; iy < kWinCells; ++iy) d[kRecBelowDw + iy] = lo[iy];
            d[kRecBelowHiDw] = hi[0] | hi[1] << 8 | hi[2] << 16 | hi[3] << 24;
            d[kRecBelowHiDw + 1] = hi[4];
        }
    }
}

// The walks: a thread per walker of the chunk, its record's windows copied into its own LDS.
template <bool X87>
__global__ __launch_bounds__(kDetThreads) void det_walk_kernel(DetArgs a)
{
    __shared__ __attribute__((aligned(4))) uint8_t windows[kDetThreads * kWinStride];
    __shared__ int ls_s[kDetMaxLayers + 1];
    __shared__ DetGeom geom_s;
    const int p = blockIdx.y, c0 = blockIdx.x * kRefineChunk;
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n_layers = a.dg->n_layers, n = ls[n_layers];
    if (c0 >= n) return;
    if (threadIdx.x <= kDetMaxLayers) ls_s[threadIdx.x] = (int)threadIdx.x <= n_layers ? ls[threadIdx.x] : 0x7fffffff;
    geom_to_lds(&geom_s, a.dg);
    __syncthreads();
    const int nt = a.walk_count[(int64_t)p * a.walk_chunks + blockIdx.x];
    const int64_t cb = (int64_t)p * a.cand_cap;
    PairView v = pair_view(a, p);
    v.g = &geom_s;
    uint8_t *cells = windows + threadIdx.x * kWinStride;
    for (int k = threadIdx.x; k < nt; k += kDetThreads) {
        const int i = a.walk_list[cb + c0 + k];
        const uint32_t xy = a.cand_xy[cb + i];
        const uint8_t flag = a.cand_flag[cb + i];
        const uint4 *rec = reinterpret_cast<const uint4 *>(a.cand_cells + (cb + c0 + k) * kRecBytes);
        uint4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = rec[j];  // the record: one line
        int layer = 0;
#pragma unroll
        for (int l = 1; l < kDetMaxLayers; ++l) layer += i >= ls_s[l] ? 1 : 0;
        // record -> window rows of 8 bytes (above: four cells a row, below: four and one; the rest of a window is never read: a
        // walk that leaves the filled cells is reported)
        uint32_t *cw = reinterpret_cast<uint32_t *>(cells);
        const uint32_t words[16] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w, r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
        for (int iy = 0; iy < kWinCells - 1; ++iy) cw[2 * iy] = words[iy];
#pragma unroll
        for (int iy = 0; iy < kWinCells; ++iy) {
            cw[kWinBytes / 4 + 2 * iy] = words[kRecBelowDw + iy];
            cw[kWinBytes / 4 + 2 * iy + 1] = (words[kRecBelowHiDw + (iy >> 2)] >> (8 * (iy & 3))) & 0xffu;
        }
        int own_patch[9], s58[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            own_patch[c] = (int)((words[kRecOwnDw + (c >> 2)] >> (8 * (c & 3))) & 0xff);
            s58[c] = (int)((words[kRecOwnDw + ((9 + c) >> 2)] >> (8 * ((9 + c) & 3))) & 0xff);
        }
        finish_candidate<Fp<X87>>(a, v, cells, own_patch, s58, p, i, layer, (int)(xy & 0xffff), (int)(xy >> 16), flag != kDetMax);
    }
}

// ---- ties
// The smoothing part of isMax2D (brisk.cpp:874-933) on the reference's score cache as candidate (px, py) would find
// it (see the header comment): a cell holds its score if it is a detected corner, if the layer below asked for it,
// or if it lies in the 3x3 patch of a maximum that was processed earlier (raster order) and got as far as its patch;
// otherwise it still holds zero.  All loads are issued up front (they are independent); the logic runs on registers.
struct TieStep {
    bool ready, is_max;
};

// One bulk load, both answers.  Ready: no undecided tie that precedes this one in raster order could still change a
// cell it reads -- such a tie matters only through cells of its 3x3 patch that lie in this candidate's 5x5 window AND
// whose cached value is not settled already (settled: score 0, a detected corner, or asked for from the layer below).
// And, if ready, whether it survives the smoothed comparison.  Only the 24 cells before (px, py) in raster order can
// hold a maximum that was processed earlier.
// A candidate lies at least 3 pixels inside its layer (the score is zero in the border), so every row segment read
// here starts inside the layer: fourteen 8-byte loads, issued together (the two bytes some run past the end of a row
// stay inside the padded plane and are masked out).
struct TieRows {
    unsigned long long st[4], sc[5], tc[5];
};
__device__ __forceinline__ TieRows tie_load(const PairView &v, const DetLayer &L, int px, int py)
{
    const int64_t at = L.off + (int64_t)py * L.w + px;
    TieRows r;
#pragma unroll
    for (int dy = -3; dy <= 0; ++dy) r.st[dy + 3] = load8(v.status + at + dy * L.w - 3);
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        r.sc[dy + 2] = load8(v.score + at + dy * L.w - 2);
        r.tc[dy + 2] = load8(v.touch + at + dy * L.w - 2);
    }
    return r;
}
#ifdef MOFREAK_DEBUG_BOUNDS  // cell by cell, as the text above goes: the bounds-checking build holds the fast form below against it
__device__ __forceinline__ TieStep tie_decide_plain(const TieRows &rows, const DetLayer &L, int safe_threshold, int px, int py)
{
    const unsigned long long(&st_row)[4] = rows.st, (&sc_row)[5] = rows.sc, (&tc_row)[5] = rows.tc;
    uint8_t st[7][7];
    int sc[5][5];
    uint8_t tc[5][5];
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy)
#pragma unroll
        for (int dx = -3; dx <= 3; ++dx) {
            const bool before = dy < 0 || (dy == 0 && dx < 0);  // the candidate is >= 3 pixels inside: these cells exist
            st[dy + 3][dx + 3] = before ? (uint8_t)(st_row[before ? dy + 3 : 0] >> (8 * (dx + 3))) : (uint8_t)kStNone;
        }
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int qx = px + dx, qy = py + dy;
            const bool in = qx >= 3 && qy >= 3 && qx < L.w - 3 && qy < L.h - 3;
            sc[dy + 2][dx + 2] = in ? (int)((sc_row[dy + 2] >> (8 * (dx + 2))) & 0xff) : 0;
            tc[dy + 2][dx + 2] = in ? (uint8_t)(tc_row[dy + 2] >> (8 * (dx + 2))) : (uint8_t)0;
        }
    int r[5][5];
    bool waits = false;
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            const int s = sc[y][x];
            const bool settled = s == 0 || s >= safe_threshold || tc[y][x] != 0;
            bool filled = s >= safe_threshold || tc[y][x] != 0, pending_near = false;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {  // window (y+dy, x+dx) = cell + (dx-1, dy-1)
                    filled |= st[y + dy][x + dx] == kStReached;
                    pending_near |= st[y + dy][x + dx] == kStPending;
                }
            r[y][x] = filled ? s : 0;
            waits |= !settled && pending_near;
        }
    const int center = r[2][2];
    auto smooth = [&](int cx, int cy) {  // 1 2 1 / 2 4 2 / 1 2 1 around (cx, cy) in window coordinates
        return r[cy - 1][cx - 1] + 2 * r[cy - 1][cx] + r[cy - 1][cx + 1] + 2 * r[cy][cx - 1] + 4 * r[cy][cx] + 2 * r[cy][cx + 1] + r[cy + 1][cx - 1] +
               2 * r[cy + 1][cx] + r[cy + 1][cx + 1];
    };
    const int smoothedcenter = smooth(2, 2);
    bool is_max = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            if (r[2 + dy][2 + dx] == center && smooth(2 + dx, 2 + dy) > smoothedcenter) is_max = false;
        }
    return TieStep{!waits, is_max};
}
#endif
// The same decision on whole rows: the 25 cells' nine status comparisons each are a few hundred operations cell by cell, and a
// workgroup per pair has the ties of a pair to itself.  Bytes stay in their 8-byte rows: a flag is bit 7 of its byte.
__device__ __forceinline__ TieStep tie_decide_rows(const TieRows &rows, const DetLayer &L, int safe_threshold, int px, int py)
{
    typedef unsigned long long u64;
    constexpr u64 k80 = 0x8080808080808080ull, k7f = 0x7f7f7f7f7f7f7f7full, k01 = 0x0101010101010101ull;
    auto nonzero = [&](u64 x) -> u64 { return (((x & k7f) + k7f) | x) & k80; };          // flag: byte != 0
    auto equals = [&](u64 x, int c) -> u64 { return ~nonzero(x ^ (k01 * (u64)c)) & k80; };  // flag: byte == c
    auto three = [&](u64 m) -> u64 { return m | (m >> 8) | (m >> 16); };                  // flag of byte x: any of bytes x, x + 1, x + 2
    // status: rows dy = -3 .. 0, byte b <-> dx = b - 3; only cells before the candidate in raster order count
    u64 reached[4], pending[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 valid = i < 3 ? 0x0080808080808080ull : 0x0000000000808080ull;
        reached[i] = three(equals(rows.st[i], kStReached) & valid);
        pending[i] = three(equals(rows.st[i], kStPending) & valid);
    }
    // cell row y of the 5 x 5 window (dy = y - 2) has the status rows dy - 1 .. dy + 1 around it: rows y, y + 1, y + 2 of the
    // seven, of which 4 .. 6 lie behind the candidate
    const u64 reached_near[5] = {reached[0] | reached[1] | reached[2], reached[1] | reached[2] | reached[3], reached[2] | reached[3], reached[3], 0ull};
    const u64 pending_near[5] = {pending[0] | pending[1] | pending[2], pending[1] | pending[2] | pending[3], pending[2] | pending[3], pending[3], 0ull};
    // scores and touch marks: rows dy = -2 .. 2, byte b <-> dx = b - 2; cells outside the scored region count as empty
    u64 cols = 0;
#pragma unroll
    for (int x = 0; x < 5; ++x) cols |= (px + x - 2 >= 3 && px + x - 2 < L.w - 3) ? 0xffull << (8 * x) : 0ull;
    const u64 t7 = k01 * (u64)(safe_threshold & 0x7f);
    const bool t_high = (safe_threshold & 0x80) != 0;
    u64 waits = 0, r_row[5];
#pragma unroll
    for (int y = 0; y < 5; ++y) {
        const u64 in = (py + y - 2 >= 3 && py + y - 2 < L.h - 3) ? cols : 0ull;
        const u64 s = rows.sc[y] & in, t = rows.tc[y] & in;
        const u64 low_ge = (((s & k7f) | k80) - t7) & k80;                 // low seven bits >= the threshold's
        const u64 ge = t_high ? (s & low_ge) : ((s | low_ge) & k80);       // flag: score >= safe_threshold
        const u64 touched = nonzero(t);
        const u64 settled = (~nonzero(s) & k80) | ge | touched;
        const u64 filled = (ge | touched | reached_near[y]) & 0x0000008080808080ull;
        waits |= ~settled & pending_near[y] & 0x0000008080808080ull;
        r_row[y] = s & ((filled - (filled >> 7)) | filled);               // the score where the cache holds it, else 0
    }
    int r[5][5];
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
        for (int x = 0; x < 5; ++x) r[y][x] = (int)((r_row[y] >> (8 * x)) & 0xff);
    // 1 2 1 / 2 4 2 / 1 2 1 around the nine inner cells, rows first
    int h[5][3];
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
        for (int x = 0; x < 3; ++x) h[y][x] = r[y][x] + 2 * r[y][x + 1] + r[y][x + 2];
    auto smooth = [&](int cx, int cy) { return h[cy - 1][cx - 1] + 2 * h[cy][cx - 1] + h[cy + 1][cx - 1]; };
    const int center = r[2][2], smoothedcenter = smooth(2, 2);
    bool is_max = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            if (r[2 + dy][2 + dx] == center && smooth(2 + dx, 2 + dy) > smoothedcenter) is_max = false;
        }
    return TieStep{waits == 0, is_max};
}

// (the bounds-checking build decides every tie both ways and reports a difference: status bit 32)
__device__ __forceinline__ TieStep tie_decide(const DetArgs &a, const TieRows &rows, const DetLayer &L, int px, int py)
{
    const TieStep s = tie_decide_rows(rows, L, a.safe_threshold, px, py);
#ifdef MOFREAK_DEBUG_BOUNDS
    const TieStep q = tie_decide_plain(rows, L, a.safe_threshold, px, py);
    if (q.ready != s.ready || q.is_max != s.is_max) atomicOr(a.status_word, 32);
#endif
    return s;
}
__device__ __forceinline__ TieStep tie_step(const DetArgs &a, const PairView &v, const DetLayer &L, int px, int py)
{
    return tie_decide(a, tie_load(v, L, px, py), L, px, py);
}

// What a tie needs besides its neighbourhood, fetched together with it (nothing here depends on the decision)
struct TieCand {
    uint32_t xy, win;
    uint8_t spec;
    unsigned long long asked;
};
__device__ __forceinline__ TieCand tie_cand(const DetArgs &a, int64_t ci, uint32_t xy)
{
    return TieCand{xy, a.cand_win[ci], a.cand_spec[ci], a.cand_asked[ci]};
}

// The status map is the one thing ties of a layer tell each other: a byte per pixel that goes from pending to its
// final value once.
__device__ __forceinline__ void tie_apply(const DetArgs &a, const PairView &v, const DetLayer &L, int64_t ci, const TieCand &c, int layer, int px, int py,
                                          bool is_max)
{
    uint8_t *st = v.status + L.off + (int64_t)py * L.w + px;
    if (is_max) {  // publish what the refinement kernel parked
        a.cand_flag[ci] = kDetMax;
        a.cand_emit[ci] = (c.spec & kEmit) ? 1 : 0;
        if (c.asked) apply_asked(v, layer + 1, (int)(c.win & 0xffff), (int)(c.win >> 16), c.asked);
        __hip_atomic_store(st, (c.spec & kReached) ? kStReached : kStDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        a.cand_flag[ci] = kDetNotMax;
        __hip_atomic_store(st, (uint8_t)kStDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Layer 0's ties at first sight, chip-wide: nothing a layer-0 tie reads is still to come once the walks are through (its
// layer has no layer below that could ask for cells in it), so every tie of layer 0 can take its first look at the same
// time -- a workgroup per chunk of candidates, straight off the chunk's tie list -- instead of one CU per pair fetching their
// 17 scattered lines each.  A tie that is ready decides, publishes and is crossed off its list (the candidate index
// complemented); the others are left to det_tie_kernel's chains.  Ties that decide here side by side: as there (a ready tie
// depends on no pending one; a status byte changes once; whoever reads the old value merely waits).
__device__ __forceinline__ int tie_candidate(const DetTie &e) { return e.cand < 0 ? ~e.cand : e.cand; }

__global__ __launch_bounds__(kDetThreads) void det_tie_first_kernel(DetArgs a)
{
    const int p = blockIdx.y;
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n0 = ls[1];  // layer 0's candidates: [0, n0)
    const int64_t cb = (int64_t)p * a.cand_cap;
    const PairView v = pair_view(a, p);
    const DetLayer L = a.dg->L[0];
    for (int c = blockIdx.x; c * kRefineChunk < n0; c += gridDim.x) {
        const int nt = a.tie_count[(int64_t)p * a.walk_chunks + c];
        DetTie *list = a.tie_list + cb + (int64_t)c * kRefineChunk;
        for (int k = threadIdx.x; k < nt; k += kDetThreads) {
            const DetTie e = list[k];
            if (e.cand >= n0) continue;  // (the chunk that runs on into layer 1)
            const int px = (int)(e.xy & 0xffff), py = (int)(e.xy >> 16);
            const TieCand cnd = tie_cand(a, cb + e.cand, e.xy);
            const TieStep step = tie_step(a, v, L, px, py);
            if (step.ready) {
                tie_apply(a, v, L, cb + e.cand, cnd, 0, px, py, step.is_max);
                list[k].cand = ~e.cand;
            }
        }
    }
}

// The ties of a pair: one workgroup per pair.  det_window_kernel left the ties of every chunk of candidates as a list
// (candidate, position); the chunks' lists, one behind the other, are the pair's ties in candidate order -- layer after
// layer, raster order inside a layer.  Prologue: the chunk counts are summed up, the list is copied into LDS (what does not
// fit is read from global memory where it lies), the layers' ranges in it are looked up; layer 0's ties have had their first
// sight (det_tie_first_kernel): the ones it left are put on layer 0's waiting list on the way.  Then layer after layer (a layer
// reads what the maxima of the layer below -- its ties included -- asked for in it):  (1) First sight: every tie looks at its
// neighbourhood once, two ties per thread at a time with all their loads in flight together; the ones that are ready --
// the great majority -- decide and publish on the spot (a ready tie depends on no tie that is still pending, so two of
// them never need each other's outcome, and a status byte changes once, from pending to final: whoever reads the old
// value merely waits); the others go on the waiting list.  (2) Chains: what first sight left waiting are ties that
// depend on each other: a thread per link (or several), each spinning until the links before it have published; the
// earliest pending tie of a layer is always ready, so the spinning ends.  Everything the threads tell each other stays
// inside the workgroup -- one CU, one vector cache -- so workgroup-scope ordering is all it takes, and all the waves
// involved are resident.
constexpr int kTieThreads = 512, kTieWaves = kTieThreads / 64;
constexpr int kEmitChunk = 1024;  // candidates per chunk of the ordered emission
#ifdef MOFREAK_DEBUG_BOUNDS
constexpr int kTieListCap = 64, kDetWaitCap = 8, kTieChunksMax = 16;  // the debug build overflows all three on every tie-heavy image: the fallbacks get tested
#else
constexpr int kTieListCap = 6144, kDetWaitCap = 2048;  // ties per pair / waiting ties per pair and layer held in LDS
constexpr int kTieChunksMax = kTieThreads;             // chunks of candidates per pair the prologue takes a thread each for (more: a run of chunks each)
#endif
static_assert(kTieChunksMax <= kTieThreads && (kTieChunksMax & (kTieChunksMax - 1)) == 0, "a thread per run of chunks; the look-up halves a power of two");

__global__ __launch_bounds__(kTieThreads) void det_tie_kernel(DetArgs a)
{
    __shared__ DetTie tie_s[kTieListCap];
    __shared__ int wait_idx[kDetWaitCap], chunk_off[kTieChunksMax + 1], wave_tot[kTieWaves], n_wait_s[kDetMaxLayers], ls_s[kDetMaxLayers + 1],
        tie_lo_s[kDetMaxLayers + 1];
    __shared__ DetGeom geom_s;
    const int p = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n_layers = a.dg->n_layers;
    const int n_chunks = (ls[n_layers] + kRefineChunk - 1) / kRefineChunk;  // (det_window_kernel wrote the counts of exactly these)
    const int64_t cb = (int64_t)p * a.cand_cap;
    // ---- prologue
    // (a thread per chunk; a pair with more chunks than threads -- a raised candidate capacity and an image that fills it --
    // has a thread sum a run of `per` consecutive chunks, and a look-up walks the run it lands in)
    const int per = (n_chunks + kTieChunksMax - 1) / kTieChunksMax;
    const int32_t *chunk_cnt = a.tie_count + (int64_t)p * a.walk_chunks;
    int own_cnt = 0;
    for (int c = threadIdx.x * per; c < min((int)(threadIdx.x + 1) * per, n_chunks); ++c) own_cnt += chunk_cnt[c];
    if (threadIdx.x <= kDetMaxLayers) ls_s[threadIdx.x] = (int)threadIdx.x <= n_layers ? ls[threadIdx.x] : 0x7fffffff;
    if (threadIdx.x < kDetMaxLayers) n_wait_s[threadIdx.x] = 0;
    geom_to_lds(&geom_s, a.dg);
    int incl = own_cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, n_ties = 0;
#pragma unroll
    for (int w = 0; w < kTieWaves; ++w) {
        before += w < wave ? wave_tot[w] : 0;
        n_ties += wave_tot[w];
    }
    if (threadIdx.x < kTieChunksMax) chunk_off[threadIdx.x] = before + incl - own_cnt;  // (runs past the last chunk: the total)
    if (threadIdx.x == 0) chunk_off[kTieChunksMax] = n_ties;
    __syncthreads();
    // tie k of the pair: in which chunk's list, and there
    auto from_global = [&](int k) -> DetTie {
        int c = 0;
#pragma unroll
        for (int step = kTieChunksMax / 2; step >= 1; step >>= 1)
            if (chunk_off[c + step] <= k) c += step;  // the last run that starts at or before k (empty ones in front of it start there too)
        int off = chunk_off[c];
        if (per > 1) {
            c *= per;
            for (int cnt = chunk_cnt[c]; off + cnt <= k; cnt = chunk_cnt[c]) {
                off += cnt;
                ++c;
            }
        }
        return a.tie_list[cb + (int64_t)c * kRefineChunk + (k - off)];
    };
    auto tie_at = [&](int k) -> DetTie { return k < kTieListCap ? tie_s[k] : from_global(k); };
    {
        const int n_lds = min(n_ties, kTieListCap);
        constexpr int kU = 4;  // loads in flight per thread
        for (int k0 = threadIdx.x; k0 < n_lds; k0 += kU * kTieThreads) {
            DetTie e[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) e[u] = from_global(min(k0 + u * kTieThreads, n_lds - 1));
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (k0 + u * kTieThreads < n_lds) {
                    tie_s[k0 + u * kTieThreads] = e[u];
                    if (e[u].cand >= 0 && e[u].cand < ls_s[1]) {  // layer 0, not ready at first sight
                        const int w = atomicAdd(&n_wait_s[0], 1);
                        if (w < kDetWaitCap) wait_idx[w] = k0 + u * kTieThreads;
                    }
                }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && n_ties > kTieListCap) n_wait_s[0] = kDetWaitCap + 1;  // not all of them seen here: every tie of layer 0 is looked at again
    if (threadIdx.x <= n_layers) {  // where a layer's ties begin: the first tie whose candidate is not below the layer's first
        const int first = ls_s[threadIdx.x];
        int lo = 0, hi = n_ties;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tie_candidate(tie_at(mid)) < first)
                lo = mid + 1;
            else
                hi = mid;
        }
        tie_lo_s[threadIdx.x] = lo;
    }
    __syncthreads();
    PairView v = pair_view(a, p);
    v.g = &geom_s;
    // ---- layer after layer
    for (int layer = 0; layer < n_layers; ++layer) {
        const int lo = tie_lo_s[layer], hi = tie_lo_s[layer + 1];
        if (hi <= lo) continue;
        const DetLayer L = geom_s.L[layer];
        // (1) first sight, ties k and k + kTieThreads of a thread's stretch together
        auto settle = [&](int k, const DetTie e, const TieCand &c, const TieStep step) {
            const int px = (int)(e.xy & 0xffff), py = (int)(e.xy >> 16);
            if (step.ready)
                tie_apply(a, v, L, cb + e.cand, c, layer, px, py, step.is_max);
            else {
                const int w = atomicAdd(&n_wait_s[layer], 1);
                if (w < kDetWaitCap) wait_idx[w] = k;
            }
        };
        for (int k0 = lo + (int)threadIdx.x; k0 < hi && layer > 0; k0 += 2 * kTieThreads) {  // (layer 0: det_tie_first_kernel)
            const int k1 = k0 + kTieThreads;
            const bool two = k1 < hi;
            const DetTie e0 = tie_at(k0), e1 = tie_at(two ? k1 : k0);
            const TieCand c0 = tie_cand(a, cb + e0.cand, e0.xy), c1 = tie_cand(a, cb + e1.cand, e1.xy);
            const TieRows r0 = tie_load(v, L, (int)(e0.xy & 0xffff), (int)(e0.xy >> 16)), r1 = tie_load(v, L, (int)(e1.xy & 0xffff), (int)(e1.xy >> 16));
            const TieStep s0 = tie_decide(a, r0, L, (int)(e0.xy & 0xffff), (int)(e0.xy >> 16));
            const TieStep s1 = tie_decide(a, r1, L, (int)(e1.xy & 0xffff), (int)(e1.xy >> 16));
            settle(k0, e0, c0, s0);
            if (two) settle(k1, e1, c1, s1);
        }
        __syncthreads();  // the waiting list is complete; what first sight published is visible
        // (2) chains
        const int n_wait = n_wait_s[layer];
        if (n_wait != 0) {
            const bool listed = n_wait <= kDetWaitCap;  // list overflow: every tie of the layer, each by the thread that saw it first
            const int n_items = listed ? n_wait : hi - lo;
            auto item = [&](int j) -> int { return listed ? wait_idx[j] : lo + j; };
            // a thread's first tie stays in registers between passes (there are rarely more than a few hundred per pair and
            // layer): a pass is then one round of neighbourhood loads
            const bool has_own = (int)threadIdx.x < n_items;
            DetTie own_e = tie_at(has_own ? item(threadIdx.x) : lo);
            own_e.cand = tie_candidate(own_e);
            const TieCand own = tie_cand(a, cb + own_e.cand, own_e.xy);
            bool own_waits = has_own && a.cand_flag[cb + own_e.cand] == kDetTie;  // (all of them unless the list overflowed)
            // The earliest pending tie of a layer is always ready, so a pass decides at least one tie and n_items + 1 passes
            // are enough for the longest possible chain.  Should that invariant ever break (a status byte left pending by a
            // candidate nobody lists), the thread gives up and says so (status bit 32 -> MOFREAK_ERR_HIP) instead of hanging
            // the device.
            for (int pass = 0;; ++pass) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // the pass reads what has been published by now
                bool waits = false;
                if (own_waits) {
                    const int px = (int)(own.xy & 0xffff), py = (int)(own.xy >> 16);
                    const TieStep step = tie_step(a, v, L, px, py);
                    if (step.ready) {
                        tie_apply(a, v, L, cb + own_e.cand, own, layer, px, py, step.is_max);
                        own_waits = false;
                    } else {
                        waits = true;
                    }
                }
                for (int j = threadIdx.x + kTieThreads; j < n_items; j += kTieThreads) {
                    DetTie e = tie_at(item(j));
                    e.cand = tie_candidate(e);
                    if (a.cand_flag[cb + e.cand] != kDetTie) continue;  // written by this thread only (or before this launch)
                    const TieCand c = tie_cand(a, cb + e.cand, e.xy);
                    const int px = (int)(e.xy & 0xffff), py = (int)(e.xy >> 16);
                    const TieStep step = tie_step(a, v, L, px, py);
                    if (step.ready)
                        tie_apply(a, v, L, cb + e.cand, c, layer, px, py, step.is_max);
                    else
                        waits = true;
                }
                if (!waits) break;
                if (pass > n_items) {
                    atomicOr(a.status_word, 32);
                    break;
                }
            }
        }
        __syncthreads();  // the layer is through (its threads' last decisions included); the waiting list is free again
    }
    // ---- how many of each chunk of kEmitChunk candidates are emitted, and of the pair (det_emit_scan / scatter place the chunks):
    //      the flags are final now; a wave per chunk, sixteen flag bytes (0 or 1) per lane
    static_assert(kEmitChunk == 64 * 16, "a wave reads a chunk's flags in one go");
    const int n = ls_s[n_layers];
    int mine = 0;
    for (int c0 = wave; c0 * kEmitChunk < n; c0 += 4 * kTieWaves) {
        uint4 f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (the flag array is padded: a load may run past n, never past the buffer)
            const int i = min((c0 + u * kTieWaves) * kEmitChunk + lane * 16, n);
            __builtin_memcpy(&f[u], a.cand_emit + cb + i, 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * kTieWaves, i = c * kEmitChunk + lane * 16;
            if (c * kEmitChunk >= n) break;  // wave-uniform
            const int valid = min(max(n - i, 0), 16);  // bytes of this lane's sixteen that are candidates
            const uint32_t w[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
            uint32_t cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vb = min(max(valid - 4 * q, 0), 4);
                const uint32_t m = vb >= 4 ? 0xffffffffu : (1u << (8 * vb)) - 1u;
                cnt = __builtin_amdgcn_udot4(w[q] & m, 0x01010101u, cnt, false);
            }
            const int total = wave_sum((int)cnt);
            if (lane == 0) a.emit_chunks[(int64_t)p * a.emit_chunk_cap + c] = total;
            mine += total;
        }
    }
    if (lane == 0) wave_tot[wave] = mine;  // (the prologue's slots)
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < kTieWaves; ++w) total += wave_tot[w];
        a.emit_count[p] = total;
    }
}

// ------------------------------------------------------------------ ordered emission
// The candidates of a pair in chunks of kEmitChunk: det_tie_kernel has counted how many of each chunk are emitted (and,
// summed up, of the pair); det_emit_scatter_kernel places a chunk behind the chunks before it.
// one workgroup: CSR offsets of the batch's pairs, continuing the running total of the call
__global__ __launch_bounds__(kDetThreads) void det_emit_scan_kernel(DetArgs a, int64_t *running)
{
    __shared__ long long part[kDetThreads];
    const int n = a.n_pairs;
    const int per = (n + kDetThreads - 1) / kDetThreads;
    const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += a.emit_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < kDetThreads; o <<= 1) {
        const long long t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    const long long base = *running;
    long long run = base + part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) {
        a.emit_offsets[i] = run;
        a.out_offsets[a.first_pair + i] = run;
        run += a.emit_count[i];
    }
    __syncthreads();
    if (threadIdx.x == kDetThreads - 1) {
        const long long end = base + part[kDetThreads - 1];
        a.emit_offsets[n] = end;
        a.out_offsets[a.first_pair + n] = end;
        *running = end;
        if (end > a.out_capacity) atomicOr(a.status_word, 8);
    }
}

__global__ __launch_bounds__(kDetThreads) void det_emit_scatter_kernel(DetArgs a)
{
    __shared__ int wave_cnt[4], chunk_base, ls_s[kDetMaxLayers + 1], w_s[kDetMaxLayers];
    __shared__ long long off_s[kDetMaxLayers];
    const int p = blockIdx.y, c = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n_layers = a.dg->n_layers, n = ls[n_layers];
    if (c * kEmitChunk >= n) return;
    if (wave == 1 && lane <= kDetMaxLayers) {  // the pair's layer boundaries and the layers' geometry: one round trip for everybody
        ls_s[lane] = lane <= n_layers ? ls[lane] : 0x7fffffff;
        if (lane < kDetMaxLayers) {
            w_s[lane] = a.dg->L[lane].w;
            off_s[lane] = a.dg->L[lane].off;
        }
    }
    if (wave == 0) {  // emitted candidates in the chunks before this one
        int before = 0;
        for (int k = lane; k < c; k += 64) before += a.emit_chunks[(int64_t)p * a.emit_chunk_cap + k];
        before = wave_sum(before);
        if (lane == 0) chunk_base = before;
    }
    __syncthreads();
    long long run = a.emit_offsets[p] + chunk_base;
    const int64_t plane = (int64_t)p * a.dg->plane_bytes;
    for (int i0 = c * kEmitChunk; i0 < min(n, (c + 1) * kEmitChunk); i0 += kDetThreads) {
        const int i = i0 + threadIdx.x;
        const int64_t ci = (int64_t)p * a.cand_cap + i;
        const bool e = i < n && a.cand_emit[ci];
        if (i < n) {
            // Everything that read the two bookkeeping maps of the tie logic is done: take back what this candidate put
            // there -- its status byte and the touch bytes of the cells it asked for in the layer above -- so that both
            // maps are all zero again when the next call starts (nobody clears them wholesale).
            const uint8_t fl = a.cand_flag[ci], sp = a.cand_spec[ci];
            if (fl == kDetMax || (sp & kWasTie)) {  // maxima and ties own a status byte; only maxima have touched cells of the layer above
                const uint32_t xy = a.cand_xy[ci];
                const unsigned long long asked = fl == kDetMax ? a.cand_asked[ci] : 0ull;
                const uint32_t win = a.cand_win[ci];  // (meaningful only with asked != 0)
                int layer = 0;
#pragma unroll
                for (int l = 1; l < kDetMaxLayers; ++l) layer += i >= ls_s[l] ? 1 : 0;
                a.status[plane + off_s[layer] + (int64_t)(xy >> 16) * w_s[layer] + (xy & 0xffff)] = kStNone;
                if (asked) {
                    // whole window rows at a time (eight bytes: two more than the window is wide -- every byte of the map
                    // ends up zero anyway, the planes are padded, and nobody reads the map any more in this call)
                    uint8_t *ub = a.touch + plane + off_s[layer + 1] + (int64_t)(win >> 16) * w_s[layer + 1] + (win & 0xffff);
                    const int uw = w_s[layer + 1];
                    const unsigned long long zero = 0ull;
#pragma unroll
                    for (int iy = 0; iy < kWinCells; ++iy)
                        if ((asked >> (iy * kWinSide)) & 0x3f) __builtin_memcpy(ub + iy * uw, &zero, 8);
                }
            }
        }
        const unsigned long long m = __ballot(e);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wave_cnt[w];
            all += wave_cnt[w];
        }
        if (e) {
            const long long o = run + before + __popcll(m & ((1ull << lane) - 1));
            if (o < a.out_capacity) {
                const DetResult r = a.cand_res[ci];
                a.out_kps[o] = mofreak_keypoint{r.x, r.y, r.size};
                if (a.out_response) a.out_response[o] = r.response;
                if (a.out_layer) {
                    int layer = 0;
#pragma unroll
                    for (int l = 1; l < kDetMaxLayers; ++l) layer += i >= ls_s[l] ? 1 : 0;
                    a.out_layer[o] = layer;
                }
            }
        }
        run += all;
        __syncthreads();
    }
}

}  // namespace

int launch_det_pyramid(const DetArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DetGeom &g = a.g;
    if (g.n_layers <= kPyrLayers && g.L[0].w > 0 && g.L[0].h > 0 && (g.L[0].w & 15) == 0) {
        const PyrBand B = pyr_band(g);
        if (B.bytes <= 64 * 1024) {  // the whole pyramid band by band in one launch
            hipLaunchKernelGGL(det_pyramid_fused_kernel, dim3((g.L[0].h + kPyrBand - 1) / kPyrBand, a.n_pairs), dim3(kDetThreads), (size_t)B.bytes, s, a);
            return (int)hipGetLastError();
        }
    }
    if (g.L[0].w > 0 && g.L[0].h > 0)
        hipLaunchKernelGGL(det_diff_kernel, dim3((unsigned)(((int64_t)((g.L[0].w + 15) / 16) * g.L[0].h + kDetThreads - 1) / kDetThreads), a.n_pairs), dim3(kDetThreads), 0, s, a);
    // BriskScaleSpace::constructPyramid (brisk.cpp:572-588): layer 1 = 2/3 of layer 0, layer i >= 2 = half of layer i-2
    for (int l = 1; l < g.n_layers; ++l) {
        if (g.L[l].w == 0 || g.L[l].h == 0) continue;
        if (l == 1) {  // a thread per 15-column block of the source row pair + one per remainder pixel
            const int hsize = g.L[0].w / 15, items = hsize + (g.L[1].w - 10 * hsize);
            hipLaunchKernelGGL(det_twothird_kernel, dim3((items + kDetThreads - 1) / kDetThreads, g.L[l].h, a.n_pairs), dim3(kDetThreads), 0, s, a, 0, 1);
        } else {  // a thread per four output pixels
            hipLaunchKernelGGL(det_half_kernel, dim3((g.L[l].w + 4 * kDetThreads - 1) / (4 * kDetThreads), g.L[l].h, a.n_pairs), dim3(kDetThreads), 0, s, a, l - 2, l);
        }
    }
    return (int)hipGetLastError();
}

int launch_det_scores(const DetArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int l = 0; l < a.g.n_layers; ++l) {
        const dim3 grid((a.g.L[l].w + kScoreTileW - 1) / kScoreTileW, (a.g.L[l].h + kScoreTileH - 1) / kScoreTileH, a.n_pairs);
        if (grid.x == 0 || grid.y == 0) continue;
        hipLaunchKernelGGL(det_dense_score_kernel, grid, dim3(kDetThreads), 0, s, a, l);
    }
    return (int)hipGetLastError();
}

#ifdef MOFREAK_DEBUG_BOUNDS
// (bounds-checking build) The two bookkeeping maps must be all zero when a call starts -- every call takes back the bytes it
// set, a new buffer is cleared: status bit 64 if a byte is not (a stale byte bends a tie decision once in a long while).
__global__ __launch_bounds__(kDetThreads) void det_maps_clean_check_kernel(DetArgs a)
{
    const int64_t n16 = (int64_t)a.n_pairs * a.dg->plane_bytes / 16;
    unsigned int any = 0;
    for (int64_t i = (int64_t)blockIdx.x * kDetThreads + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kDetThreads) {
        const uint4 t = reinterpret_cast<const uint4 *>(a.touch)[i], u = reinterpret_cast<const uint4 *>(a.status)[i];
        any |= t.x | t.y | t.z | t.w | u.x | u.y | u.z | u.w;
    }
    if (any) atomicOr(a.status_word, 64);
}
#endif

int launch_det_corners(const DetArgs &a, void *stream)
{
#ifdef MOFREAK_DEBUG_BOUNDS
    if (a.n_pairs > 0) hipLaunchKernelGGL(det_maps_clean_check_kernel, dim3(1024), dim3(kDetThreads), 0, static_cast<hipStream_t>(stream), a);  // (after the pyramid: the status word is zero)
#endif
    const int tiles = a.g.tile_start[a.g.n_layers];  // every layer's tiles in one launch, an eighth of the list per XCD
    if (tiles > 0 && a.n_pairs > 0)
        hipLaunchKernelGGL(det_corner_kernel, dim3(((tiles + 7) / 8) * 8, a.n_pairs), dim3(kDetThreads), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int launch_det_keypoints(const DetArgs &a, int64_t *running, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(det_scan_kernel, dim3(a.n_pairs), dim3(kDetThreads), 0, s, a);
    hipLaunchKernelGGL(det_candidates_kernel, dim3((a.g.cand_group_start[a.g.n_layers] + 3) / 4, a.n_pairs), dim3(kDetThreads), 0, s, a);
    const dim3 rgrid((a.cand_cap + kRefineChunk - 1) / kRefineChunk, a.n_pairs);
    hipLaunchKernelGGL(det_window_kernel, rgrid, dim3(kDetThreads), 0, s, a);
    if (a.fp_x87)
        hipLaunchKernelGGL(det_walk_kernel<true>, rgrid, dim3(kDetThreads), 0, s, a);
    else
        hipLaunchKernelGGL(det_walk_kernel<false>, rgrid, dim3(kDetThreads), 0, s, a);
    // ties: layer by layer (a layer's ties read what the maxima of the layer below asked for in it)
    hipLaunchKernelGGL(det_tie_first_kernel, dim3(std::min(a.walk_chunks, 64), a.n_pairs), dim3(kDetThreads), 0, s, a);  // layer 0 at first sight
    hipLaunchKernelGGL(det_tie_kernel, dim3(a.n_pairs), dim3(kTieThreads), 0, s, a);  // every layer's ties, layer after layer
    const dim3 egrid((a.cand_cap + kEmitChunk - 1) / kEmitChunk, a.n_pairs);
    hipLaunchKernelGGL(det_emit_scan_kernel, dim3(1), dim3(kDetThreads), 0, s, a, running);
    hipLaunchKernelGGL(det_emit_scatter_kernel, egrid, dim3(kDetThreads), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace mofreak
