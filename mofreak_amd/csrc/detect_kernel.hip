// Keypoint detector for gfx950: BRISK scale-space corners on the frame-difference image.
//
// Replaces, per frame pair, BriskFeatureDetector(30).detect(diff_img) (MoFREAKUtilities.cpp:420-423), i.e.
// BriskScaleSpace::constructPyramid + getKeypoints (brisk.cpp:572-704) with the OAST 9/16 detector and the
// AGAST 5/8 score behind them (oast9_16.cc:46, oast9_16_nms.cc:42, agast5_8_nms.cc:42).
//
// The reference walks its candidates one by one, asks for corner scores lazily and caches them; this design turns
// that inside out so that every step is data parallel, and keeps the reference's results bit for bit:
//
//   pyramid     difference image, then the 2/3 and 1/2 resamplers as closed forms of their SSE sequences
//               (every output byte depends on which part of the SIMD loop produced it: main blocks, the odd
//               block, the scalar tail), one thread per output pixel
//   scores      the corner score of EVERY pixel of every layer in one pass: the bisection around the decision
//               tree evaluates "largest b for which some arc of 9 ring pixels is all brighter than c+b or all
//               darker than c-b", which is max over arcs of the arc's smallest |difference|, minus one
//   candidates  pixels with score >= threshold in raster order per layer (counted per row while scoring, scanned,
//               scattered by one wave per row), classified by the strict part of isMax2D
//   refinement  one thread per surviving candidate walks the layer above / below exactly as refine3D does, on
//               the dense score maps
//   ties        isMax2D breaks ties on the reference's RAW score cache, which holds a score only where one has
//               been asked for before -- so the outcome depends on the processing order.  Reproduced exactly:
//               refinement marks the cells it asks for in the layer above ("touch" map), a maximum that reaches
//               its own 3x3 patch marks itself ("status" map), and a tie is decided from score * (detected |
//               touched from below | inside the patch of a raster-earlier maximum).  Ties that could depend on
//               each other are resolved in rounds by one workgroup per pair, layer by layer.
//   emission    ordered compaction in (layer, raster) order -- the order of the reference's keypoint vector, which
//               the rows of a .mofreak file inherit.
//
// Integer/byte work bound by LDS and VALU issue, no MFMA.  Floating point mirrors the reference's expressions
// (float vs double literals) one operation at a time; compile with -ffp-contract=off.
#include <type_traits>

#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kDetThreads = 256;

struct PairView {
    const DetGeom *g;
    const uint8_t *img;
    const uint8_t *score;
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
    v.touch = a.touch + o;
    v.status = a.status + o;
    return v;
}

__device__ __forceinline__ int avg_u8(int a, int b) { return (a + b + 1) >> 1; }  // _mm_avg_epu8

__device__ __forceinline__ unsigned long long load8(const uint8_t *p)
{
    unsigned long long q;
    __builtin_memcpy(&q, p, 8);  // one unaligned 8-byte load
    return q;
}

// ------------------------------------------------------------------ pyramid
__global__ __launch_bounds__(kDetThreads) void det_diff_kernel(DetArgs a)
{
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

constexpr int kScoreTileW = kDetTileW, kScoreTileH = kDetTileH, kScoreLdsW = 72;  // LDS rows start at image column x0 - 4: aligned dwords

// a tile of a layer's image with its 3-pixel ring halo -> LDS (rows of kScoreLdsW bytes, first column x0 - 4)
__device__ __forceinline__ void score_tile_load(uint8_t *tile, const uint8_t *img, const DetLayer &L, int x0, int y0)
{
    if ((L.w & 3) == 0) {  // rows start on dword boundaries (layer planes are 64-byte aligned): 18 aligned dwords per tile row
        for (int t = threadIdx.x; t < (kScoreTileH + 6) * (kScoreLdsW / 4); t += kDetThreads) {
            const int r = t / (kScoreLdsW / 4), k = t - r * (kScoreLdsW / 4);
            const int gx = x0 - 4 + 4 * k, gy = y0 - 3 + r;
            const bool in = gx >= 0 && gx < L.w && gy >= 0 && gy < L.h;
            const uint32_t v = in ? *reinterpret_cast<const uint32_t *>(img + (int64_t)gy * L.w + gx) : 0u;
            *reinterpret_cast<uint32_t *>(tile + r * kScoreLdsW + 4 * k) = v;
        }
    } else {
        for (int t = threadIdx.x; t < (kScoreTileH + 6) * kScoreLdsW; t += kDetThreads) {
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
    score_tile_load(tile, a.img + (int64_t)p * a.dg->plane_bytes + L.off, L, x0, y0);
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
// One workgroup per 64 x 32 tile of a layer, all layers in one launch.  The reference's decision tree answers "is there an
// arc of 9 ring pixels all brighter than c + t or all darker than c - t" with a handful of comparisons for most pixels;
// the data-parallel counterpart: (1) a necessary condition on the four compass points of the ring -- an arc of 9 holds at
// least two of them, so two must be brighter (darker) -- passes a few percent of a difference image's pixels; (2) the
// survivors of a tile are compacted and only they get the full score (the arc extremes), with full wavefronts.  Leaves,
// per tile: the score plane (score where >= threshold, 0 elsewhere: the reference's cache after getAgastPoints,
// brisk.cpp:1676-1690), a 64-bit hit mask per tile row, and the per-row corner counts.
__global__ __launch_bounds__(kDetThreads) void det_corner_kernel(DetArgs a)
{
    __shared__ __attribute__((aligned(4))) uint8_t tile[(kScoreTileH + 6) * kScoreLdsW];
    __shared__ __attribute__((aligned(8))) uint8_t out[kScoreTileH * kScoreTileW];
    __shared__ uint16_t list[kScoreTileH * kScoreTileW];
    __shared__ unsigned long long row_mask[kScoreTileH];
    __shared__ int n_list;
    const int p = blockIdx.y, t = blockIdx.x;
    int layer = 0;  // from the argument block's copy of the geometry with constant indices: scalar compares, no memory
#pragma unroll
    for (int k = 1; k < kDetMaxLayers; ++k) layer += (k < a.g.n_layers && t >= a.g.tile_start[k]) ? 1 : 0;
    const DetLayer L = a.dg->L[layer];
    const int tiles_x = a.dg->tiles_x[layer], tl = t - a.dg->tile_start[layer];
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x, x0 = tx * kScoreTileW, y0 = ty * kScoreTileH;
    const int64_t plane = (int64_t)p * a.dg->plane_bytes + L.off;
    score_tile_load(tile, a.img + plane, L, x0, y0);
    for (int k = threadIdx.x; k < kScoreTileH * kScoreTileW / 4; k += kDetThreads) reinterpret_cast<uint32_t *>(out)[k] = 0u;
    if (threadIdx.x < kScoreTileH) row_mask[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int thr = a.safe_threshold;
#pragma unroll
    for (int it = 0; it < kScoreTileH / 4; ++it) {
        const int ry = wave + 4 * it, x = x0 + lane, y = y0 + ry;
        const uint8_t *q = tile + (ry + 3) * kScoreLdsW + lane + 4;
        const int c = q[0], hi = c + thr, lo = c - thr;
        const int l = q[-3], u = q[-3 * kScoreLdsW], r = q[3], d = q[3 * kScoreLdsW];
        const int nb = (l > hi) + (u > hi) + (r > hi) + (d > hi), nd = (l < lo) + (u < lo) + (r < lo) + (d < lo);
        const bool pass = (nb >= 2 || nd >= 2) && x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3;
        const unsigned long long m = __ballot(pass);
        if (m) {  // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&n_list, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (pass) list[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)(ry << 6 | lane);
        }
    }
    __syncthreads();
    const int n = n_list;
    for (int i = threadIdx.x; i < n; i += kDetThreads) {
        const int e = list[i], ry = e >> 6, lx = e & 63;
        const int s = ring_score<kScoreLdsW>(tile + (ry + 3) * kScoreLdsW + lx + 4);
        if (s >= thr) {
            out[ry * kScoreTileW + lx] = (uint8_t)s;
            atomicOr(&row_mask[ry], 1ull << lx);
        }
    }
    __syncthreads();
    // the tile's scores (zeros included: the plane is rewritten by every call), masks and counts
    uint8_t *score = a.score + plane;
    if ((L.w & 3) == 0) {
        for (int k = threadIdx.x; k < kScoreTileH * kScoreTileW / 4; k += kDetThreads) {
            const int ry = k >> 4, cx = (k & 15) * 4, x = x0 + cx, y = y0 + ry;
            if (x < L.w && y < L.h) *reinterpret_cast<uint32_t *>(score + (uint32_t)y * (uint32_t)L.w + (uint32_t)x) = reinterpret_cast<const uint32_t *>(out)[k];
        }
    } else {
        for (int k = threadIdx.x; k < kScoreTileH * kScoreTileW; k += kDetThreads) {
            const int ry = k >> 6, x = x0 + (k & 63), y = y0 + ry;
            if (x < L.w && y < L.h) score[(uint32_t)y * (uint32_t)L.w + (uint32_t)x] = out[k];
        }
    }
    if (threadIdx.x < kScoreTileH && y0 + (int)threadIdx.x < L.h) {
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
        a.emit_count[p] = 0;  // summed up by det_emit_count_kernel
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

// One wave per layer row: the row's corners in x order (the order of OastDetector9_16::detect), read off the row's hit
// masks -- a lane per 64-pixel word, a prefix sum over the words' populations, then every lane walks the few bits of
// its own word -- each classified by the strict part of isMax2D (brisk.cpp:838-872) on the 3 x 3 scores around it.
// A neighbour that is no corner holds 0 or a true score below the threshold in the score plane: never >= a corner's.
__global__ __launch_bounds__(kDetThreads) void det_candidates_kernel(DetArgs a)
{
    // the wave's row is the same in every lane: kept in a scalar register, and so is everything that follows from it alone
    const int p = blockIdx.y, row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.g.total_rows) return;
    // which layer the row belongs to: from the argument block's copy of the geometry with constant indices (scalar
    // registers, no memory) -- the same search on the device copy is a chain of dependent loads
    int layer = 0;
#pragma unroll
    for (int k = 1; k < kDetMaxLayers; ++k) layer += (k < a.g.n_layers && row >= a.g.L[k].row_base) ? 1 : 0;
    const DetLayer L = a.dg->L[layer];
    const int y = row - L.row_base;
    if (y < 3 || y >= L.h - 3) return;
    const int32_t *rc = a.row_count + (int64_t)p * (a.dg->total_rows + 1);
    int base = rc[row];
    if (rc[row + 1] == base) return;
    const int64_t plane = (int64_t)p * a.dg->plane_bytes + L.off;
    const uint8_t *srow = a.score + plane + (int64_t)y * L.w;
    const int64_t cbase = (int64_t)p * a.cand_cap;
    const int tiles_x = a.dg->tiles_x[layer];
    const unsigned long long *mrow = a.hit_mask + (int64_t)p * a.dg->mask_words + a.dg->mask_off[layer] + (int64_t)y * tiles_x;
    const int tie_base = a.layer_start[(int64_t)p * (kDetMaxLayers + 1) + layer];
    for (int w0 = 0; w0 < tiles_x; w0 += 64) {
        const int wi = w0 + lane;
        unsigned long long m = wi < tiles_x ? mrow[wi] : 0ull;
        const int cnt = __popcll(m);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        int idx = base + incl - cnt;
        base += __shfl(incl, 63);
        while (m) {
            const int x = 64 * wi + (__ffsll((long long)m) - 1);
            m &= m - 1;
            if (idx < a.cand_cap) {
                uint32_t r0, r1, r2;  // scores x - 1 .. x + 2 of the three rows (a corner lies >= 3 pixels inside the layer)
                __builtin_memcpy(&r0, srow + x - 1 - L.w, 4);
                __builtin_memcpy(&r1, srow + x - 1, 4);
                __builtin_memcpy(&r2, srow + x - 1 + L.w, 4);
                const int s = (int)((r1 >> 8) & 0xff);
                int hi = 0, eq = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int v0 = (int)((r0 >> (8 * k)) & 0xff), v2 = (int)((r2 >> (8 * k)) & 0xff);
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
                a.cand_asked[cbase + idx] = 0ull;
                if (flag == kDetTie) {
                    a.status[plane + (int64_t)y * L.w + x] = kStPending;
                    const int k = atomicAdd(&a.tie_count[(int64_t)p * kDetMaxLayers + layer], 1);
                    a.tie_list[cbase + tie_base + k] = idx;
                }
            }
            ++idx;
        }
    }
}

// ------------------------------------------------------------------ refinement (thread per maximum)
// BriskLayer::getAgastScore(int, int, 1) (brisk.cpp:1685-1694) on the dense map.  MARK: record that the reference
// would have asked for (and therefore cached) this cell -- only cells of the layer ABOVE the walker matter later.
template <bool MARK>
__device__ __forceinline__ int score_at(const PairView &v, int layer, int x, int y)
{
    const DetLayer &L = v.g->L[layer];
    const bool in = x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3;
    const int64_t o = L.off + (in ? (int64_t)y * L.w + x : 0);  // unconditional load: independent loads overlap
    const int s = v.score[o];
    return in ? s : 0;
}

// The part of a neighbouring layer's score map one refinement walk can reach -- at most 5 x 5 cells from the corner
// ((int)x_1 - 1, (int)y_1 - 1) of getScoreMaxAbove/Below's sampling square, patch and tie rings included -- fetched with six
// independent 8-byte loads into the thread's own 48 bytes of LDS.  The walk itself is a chain of data-dependent early
// exits: on global memory every step would pay a full memory latency.
constexpr int kWinSide = 6, kWinRow = 8, kWinBytes = 48, kWinStride = 108;  // two windows (above, below) of 6 rows of 8 bytes per thread; 27 dwords apart: neighbouring threads hit different banks
struct Window {
    uint8_t *cells;
    int ox, oy, layer;
    unsigned long long asked;  // bit iy * kWinSide + ix: the walk asked for this cell (and the cell is inside the scored region)
    bool escaped;              // the walk left the window (cannot happen by construction; reported if it does)
};

struct WindowRows {
    unsigned long long r[kWinSide];
};
__device__ __forceinline__ WindowRows window_fetch(const PairView &v, const Window &w)
{
    const DetLayer &L = v.g->L[w.layer];
    // which of the six columns lie inside the scored region (3-pixel border): a byte mask over one 8-byte row
    unsigned long long colmask = 0;
#pragma unroll
    for (int k = 0; k < kWinSide; ++k) {
        const int x = w.ox + k;
        if (x >= 3 && x < L.w - 3) colmask |= 0xffull << (8 * k);
    }
    const int oxc = min(max(w.ox, 0), max(L.w - 1, 0));  // ox >= 0 by construction; the clamp only keeps the address sane
    if (oxc != w.ox) colmask = 0;
    WindowRows rows;
#pragma unroll
    for (int r = 0; r < kWinSide; ++r) {  // one unaligned 8-byte load per row, all six in flight together; the plane is
        const int y = w.oy + r;           // padded, so the two bytes past the window never leave the allocation
        const int yc = min(max(y, 0), max(L.h - 1, 0));
        unsigned long long q;
        __builtin_memcpy(&q, v.score + L.off + (int64_t)yc * L.w + oxc, 8);
        rows.r[r] = (y >= 3 && y < L.h - 3) ? (q & colmask) : 0ull;
    }
    return rows;
}
__device__ __forceinline__ void window_store(const Window &w, const WindowRows &rows)
{
#pragma unroll
    for (int r = 0; r < kWinSide; ++r) {
        *reinterpret_cast<uint32_t *>(w.cells + r * kWinRow) = (uint32_t)rows.r[r];
        *reinterpret_cast<uint32_t *>(w.cells + r * kWinRow + 4) = (uint32_t)(rows.r[r] >> 32);
    }
}

template <bool MARK>
__device__ __forceinline__ int window_at(const PairView &v, Window &w, int x, int y)
{
    const int ix = x - w.ox, iy = y - w.oy;
    if ((unsigned)ix >= (unsigned)kWinSide || (unsigned)iy >= (unsigned)kWinSide) {
        w.escaped = true;
        return score_at<false>(v, w.layer, x, y);
    }
    if (MARK) {
        const DetLayer &L = v.g->L[w.layer];
        if (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) w.asked |= 1ull << (iy * kWinSide + ix);
    }
    return w.cells[iy * kWinRow + ix];
}

// The cells a walk asked for in the layer above become "cached" there -- once the walker is known to be a maximum.
__device__ __forceinline__ void apply_asked(const PairView &v, int layer_above, int ox, int oy, unsigned long long asked)
{
    const DetLayer &L = v.g->L[layer_above];
    while (asked) {
        const int k = __ffsll((long long)asked) - 1;
        asked &= asked - 1;
        v.touch[L.off + (int64_t)(oy + k / kWinSide) * L.w + ox + k % kWinSide] = 1;
    }
}

// AgastDetector5_8::cornerScore from b = 0 (agast5_8_nms.cc:42; brisk.cpp:1696-1703): 5 contiguous of the 8 neighbours,
// on a 5x5 image block held in registers (rows py-2 .. py+2, bytes = columns px-2 ..): the score at (px + ax, py + ay),
// neighbours in init_pattern order (agast5_8.h:66-76).  For a candidate, which lies at least 3 pixels inside the layer,
// none of the nine positions asked for touches the 2-pixel border where BriskLayer::getAgastScore_5_8 returns 0.
__device__ __forceinline__ int score_5_8_block(const unsigned long long (&im)[5], int ax, int ay)
{
    auto at = [&](int dx, int dy) { return (int)((im[ay + dy + 2] >> (8 * (ax + dx + 2))) & 0xff); };
    const int c = at(0, 0);
    int d[8];
    d[0] = at(-1, 0) - c;
    d[1] = at(-1, -1) - c;
    d[2] = at(0, -1) - c;
    d[3] = at(1, -1) - c;
    d[4] = at(1, 0) - c;
    d[5] = at(1, 1) - c;
    d[6] = at(0, 1) - c;
    d[7] = at(-1, 1) - c;
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
// to (int)x1 + 1 <= (int)x_1 + 3: five cells; the window holds six
template <bool ABOVE>
__device__ __forceinline__ void window_place(Window &win, int layer, int x_layer, int y_layer)
{
    float x_1, x1, y_1, y1;
    walk_square<ABOVE>(layer, x_layer, y_layer, x_1, x1, y_1, y1);
    win.ox = (int)x_1 - 1;
    win.oy = (int)y_1 - 1;
    win.layer = ABOVE ? layer + 1 : layer - 1;
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
__device__ __forceinline__ Refined refine_maximum(const PairView &v, uint8_t *lds_cells, int layer, int px, int py, int threshold)
{
    // One walk above, one below, one own patch -- in the reference's order (above, below, patch), each at a single
    // call site so that everything inlines and no argument goes through the stack.
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
    // Everything the refinement reads, requested at once (the walks are chains of data-dependent early exits: fetched
    // step by step, each would pay a memory latency): the windows of the two neighbouring layers, the candidate's own
    // 3x3 patch -- its centre is the candidate's score -- and, on layer 0, the 5x5 image block behind the 5/8 scores.
    // A candidate lies at least 3 pixels inside its layer, so these row segments start inside it (the bytes that run
    // past a row's end stay inside the padded plane and are not used).
    Window wa, wb;
    wa.cells = lds_cells;
    wb.cells = lds_cells + kWinBytes;
    WindowRows ra, rb;
    if (!last) {
        window_place<true>(wa, layer, px, py);
        ra = window_fetch(v, wa);
    }
    if (layer > 0) {
        window_place<false>(wb, layer, px, py);
        rb = window_fetch(v, wb);
    }
    unsigned long long own[3], im[5];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) own[dy + 1] = load8(v.score + L.off + (int64_t)(py + dy) * L.w + px - 1);
    const bool guess_below = layer == 0 && !single;
    if (guess_below) {
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) im[dy + 2] = load8(v.img + L.off + (int64_t)(py + dy) * L.w + px - 2);
    }
    if (!last) window_store(wa, ra);
    if (layer > 0) window_store(wb, rb);
    const int center = (int)((own[1] >> 8) & 0xff);
    bool ismax = true;
    float max_above = 0.f, max_below = 0.f;
    float delta_x_above = 0.f, delta_y_above = 0.f, delta_x_below = 0.f, delta_y_below = 0.f, delta_x_layer, delta_y_layer;
    if (!last) {  // refine3D: getScoreMaxAbove first (:945-950)
        max_above = neighbour_layer_max<true, FP>(v, wa, layer, px, py, center, ismax, delta_x_above, delta_y_above);
        out.asked = wa.asked;
        out.ox = wa.ox;
        out.oy = wa.oy;
        out.escaped = wa.escaped;
        if (!ismax) return out;
    }
    if (layer > 0) {  // getScoreMaxBelow: the last layer (:651-657), octaves above 0 (:991-996), intra layers (:1049-1053)
        max_below = neighbour_layer_max<false, FP>(v, wb, layer, px, py, center, ismax, delta_x_below, delta_y_below);
        out.escaped |= wb.escaped;
        if (!ismax) return out;
    } else if (!single) {  // layer 0: guess the missing layer below with the 5/8 mask (:959-989)
        int s[9];
        int mb = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            s[k] = score_5_8_block(im, k / 3 - 1, k % 3 - 1);
            mb = max(mb, s[k]);
        }
        (void)subpixel2d(fp, s, delta_x_below, delta_y_below);
        max_below = (float)mb;
    }
    int own_patch[9];  // s_0_0, s_0_1, s_0_2, s_1_0, ... (first index x): getAgastScore(int, int, 1) on the own layer (:1685-1694)
#pragma unroll
    for (int k = 0; k < 9; ++k) own_patch[k] = (int)((own[k % 3] >> (8 * (k / 3))) & 0xff);
    const float max_layer = subpixel2d(fp, own_patch, delta_x_layer, delta_y_layer);
    out.reached = true;
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

__device__ __forceinline__ int layer_of(const int32_t *layer_start, int n_layers, int i)
{
    int l = 0;
    while (l + 1 < n_layers && i >= layer_start[l + 1]) ++l;
    return l;
}

// cand_emit bits
constexpr uint8_t kEmit = 1, kReached = 2;

// Refinement of candidate i.  It reads nothing but the dense score maps, so it does not depend on whether the
// candidate's tie (if it has one) is already decided: SPECULATIVE runs it ahead of the decision and parks what the
// decision will publish -- result, "reached its patch", and the cells it asked for in the layer above.
template <bool SPECULATIVE, class FP>
__device__ __forceinline__ void finish_candidate(const DetArgs &a, const PairView &v, uint8_t *lds_cells, int p, int i, int layer, int x, int y)
{
    const Refined r = refine_maximum<FP>(v, lds_cells, layer, x, y, a.threshold);
    const int64_t ci = (int64_t)p * a.cand_cap + i;
    a.cand_res[ci] = r.r;
    if (r.escaped) atomicOr(a.status_word, 16);
    if (SPECULATIVE) {
        a.cand_emit[ci] = 0;
        a.cand_spec[ci] = (uint8_t)((r.emit ? kEmit : 0) | (r.reached ? kReached : 0));
        a.cand_asked[ci] = r.asked;
        a.cand_win[ci] = (uint32_t)r.ox | (uint32_t)r.oy << 16;
    } else {
        a.cand_emit[ci] = r.emit ? 1 : 0;
        const DetLayer &L = a.dg->L[layer];
        v.status[L.off + (int64_t)y * L.w + x] = r.reached ? kStReached : kStDone;
        if (r.asked) apply_asked(v, layer + 1, r.ox, r.oy, r.asked);
    }
}

// maxima without ties: independent of everything else; ties: refined ahead of their decision
constexpr int kRefineChunk = 512;  // candidates per workgroup

template <bool X87>
__global__ __launch_bounds__(kDetThreads) void det_refine_kernel(DetArgs a)
{
    __shared__ __attribute__((aligned(4))) uint8_t windows[kDetThreads * kWinStride];
    __shared__ int todo[kRefineChunk], wave_cnt[4], n_todo;
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n = ls[a.dg->n_layers], c0 = blockIdx.x * kRefineChunk;
    if (c0 >= n) return;
    // Only a quarter or so of the candidates are maxima or ties: the chunk's ones are gathered first (order does not
    // matter here), so that the walks below run with full waves.
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    for (int i0 = c0; i0 < min(n, c0 + kRefineChunk); i0 += kDetThreads) {
        const int i = i0 + threadIdx.x;
        const bool take = i < n && a.cand_flag[(int64_t)p * a.cand_cap + i] != kDetNotMax;
        const unsigned long long m = __ballot(take);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = n_todo;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        if (take) todo[before + __popcll(m & ((1ull << lane) - 1))] = i;
        __syncthreads();
        if (threadIdx.x == 0) n_todo += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const PairView v = pair_view(a, p);
    for (int k = threadIdx.x; k < n_todo; k += kDetThreads) {
        const int i = todo[k];
        const int64_t ci = (int64_t)p * a.cand_cap + i;
        const uint32_t xy = a.cand_xy[ci];
        const int layer = layer_of(ls, a.dg->n_layers, i), x = (int)(xy & 0xffff), y = (int)(xy >> 16);
        if (a.cand_flag[ci] == kDetMax)
            finish_candidate<false, Fp<X87>>(a, v, windows + threadIdx.x * kWinStride, p, i, layer, x, y);
        else
            finish_candidate<true, Fp<X87>>(a, v, windows + threadIdx.x * kWinStride, p, i, layer, x, y);
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
__device__ __forceinline__ TieStep tie_step(const PairView &v, const DetLayer &L, int safe_threshold, int px, int py)
{
    const int64_t at = L.off + (int64_t)py * L.w + px;
    unsigned long long st_row[4], sc_row[5], tc_row[5];
#pragma unroll
    for (int dy = -3; dy <= 0; ++dy) st_row[dy + 3] = load8(v.status + at + dy * L.w - 3);
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        sc_row[dy + 2] = load8(v.score + at + dy * L.w - 2);
        tc_row[dy + 2] = load8(v.touch + at + dy * L.w - 2);
    }
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

// What a tie needs besides its neighbourhood, fetched together with it (nothing here depends on the decision)
struct TieCand {
    uint32_t xy, win;
    uint8_t flag, spec;
    unsigned long long asked;
};
__device__ __forceinline__ TieCand tie_cand(const DetArgs &a, int64_t ci)
{
    return TieCand{a.cand_xy[ci], a.cand_win[ci], a.cand_flag[ci], a.cand_spec[ci], a.cand_asked[ci]};
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

// Ties of one layer, all pairs at once.  First sight: every tie looks at its neighbourhood once; the ones that are
// ready -- the great majority -- decide and publish on the spot (a ready tie depends on no tie that is still pending,
// so two of them never need each other's outcome, and a status byte changes once, from pending to final: whoever reads
// the old value merely waits); the others go on the pair's waiting list.
constexpr int kTieThreads = 512, kTieGroups = 8;
#ifdef MOFREAK_DEBUG_BOUNDS
constexpr int kDetWaitCap = 8;  // the debug build overflows the list on every tie-heavy image: the scanning fallback gets tested
#else
constexpr int kDetWaitCap = 4096;  // waiting ties per pair and layer the chain kernel takes from a list (more: it scans the layer)
#endif

__global__ __launch_bounds__(kTieThreads) void det_tie_first_kernel(DetArgs a, int layer, int32_t *waiting)
{
    const int p = blockIdx.y;
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int lo = ls[layer], hi = ls[layer + 1];
    const PairView v = pair_view(a, p);
    const int64_t cb = (int64_t)p * a.cand_cap;
    const DetLayer L = a.dg->L[layer];
    int32_t *list = a.wait_list + (int64_t)p * kDetWaitCap;
    for (int i = lo + blockIdx.x * kTieThreads + threadIdx.x; i < hi; i += gridDim.x * kTieThreads) {
        const TieCand c = tie_cand(a, cb + i);
        if (c.flag != kDetTie) continue;
        const int px = (int)(c.xy & 0xffff), py = (int)(c.xy >> 16);
        const TieStep step = tie_step(v, L, a.safe_threshold, px, py);
        if (step.ready)
            tie_apply(a, v, L, cb + i, c, layer, px, py, step.is_max);
        else {
            const int k = atomicAdd(&waiting[p], 1);
            if (k < kDetWaitCap) list[k] = i;
        }
    }
}

// What first sight left waiting: chains of ties that depend on each other.  One workgroup per pair, a thread per link
// (or several), each spinning until the links before it have published; the earliest pending tie of a layer is
// always ready, so the spinning ends.  Everything the threads tell each other stays inside the workgroup -- one CU,
// one vector cache -- so workgroup-scope ordering is all it takes, and all the waves involved are resident.
__global__ __launch_bounds__(kTieThreads) void det_tie_chain_kernel(DetArgs a, int layer, const int32_t *waiting)
{
    const int p = blockIdx.x;
    const int n_wait = waiting[p];
    if (n_wait == 0) return;
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int lo = ls[layer], hi = ls[layer + 1];
    const PairView v = pair_view(a, p);
    const int64_t cb = (int64_t)p * a.cand_cap;
    const DetLayer L = a.dg->L[layer];
    const int32_t *list = a.wait_list + (int64_t)p * kDetWaitCap;
    const bool listed = n_wait <= kDetWaitCap;  // list overflow: every candidate of the layer
    const int n_items = listed ? n_wait : hi - lo;
    // a thread's first two ties stay in registers between passes (there are rarely more than a few hundred per pair and
    // layer): a pass is then one round of neighbourhood loads, not list entry -> record -> neighbourhood
    constexpr int kOwn = 2;
    TieCand own[kOwn];
    int own_i[kOwn];
    bool own_waits[kOwn];
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
        const int k = threadIdx.x + j * kTieThreads;
        own_i[j] = k < n_items ? (listed ? list[k] : lo + k) : lo;
    }
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
        own[j] = tie_cand(a, cb + own_i[j]);
        own_waits[j] = (int)threadIdx.x + j * kTieThreads < n_items && own[j].flag == kDetTie;
    }
    // The earliest pending tie of a layer is always ready, so a pass decides at least one tie and n_items + 1 passes
    // are enough for the longest possible chain.  Should that invariant ever break (a status byte left pending by a
    // candidate nobody lists), the thread gives up and says so (status bit 32 -> MOFREAK_ERR_HIP) instead of hanging
    // the device.
    for (int pass = 0;; ++pass) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // the pass reads what has been published by now
        bool waits = false;
#pragma unroll
        for (int j = 0; j < kOwn; ++j) {
            if (!own_waits[j]) continue;
            const int px = (int)(own[j].xy & 0xffff), py = (int)(own[j].xy >> 16);
            const TieStep step = tie_step(v, L, a.safe_threshold, px, py);
            if (step.ready) {
                tie_apply(a, v, L, cb + own_i[j], own[j], layer, px, py, step.is_max);
                own_waits[j] = false;
            } else {
                waits = true;
            }
        }
        for (int k = threadIdx.x + kOwn * kTieThreads; k < n_items; k += kTieThreads) {
            const int i = listed ? list[k] : lo + k;
            const TieCand c = tie_cand(a, cb + i);
            if (c.flag != kDetTie) continue;  // written by this thread only
            const int px = (int)(c.xy & 0xffff), py = (int)(c.xy >> 16);
            const TieStep step = tie_step(v, L, a.safe_threshold, px, py);
            if (step.ready)
                tie_apply(a, v, L, cb + i, c, layer, px, py, step.is_max);
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

// ------------------------------------------------------------------ ordered emission
// The candidates of a pair in chunks of kEmitChunk, one workgroup per chunk: how many of each chunk are emitted
// (and, summed up, of the pair); det_emit_scatter_kernel places a chunk behind the chunks before it.
constexpr int kEmitChunk = 1024;

__global__ __launch_bounds__(kDetThreads) void det_emit_count_kernel(DetArgs a)
{
    __shared__ int total;
    const int p = blockIdx.y, c = blockIdx.x;
    const int n = a.layer_start[(int64_t)p * (kDetMaxLayers + 1) + a.dg->n_layers];
    if (c * kEmitChunk >= n) return;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = c * kEmitChunk + threadIdx.x; i < min(n, (c + 1) * kEmitChunk); i += kDetThreads) cnt += a.cand_emit[(int64_t)p * a.cand_cap + i];
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.emit_chunks[(int64_t)p * a.emit_chunk_cap + c] = total;
        atomicAdd(&a.emit_count[p], total);  // zeroed by det_scan_kernel
    }
}

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
    __shared__ int wave_cnt[4], chunk_base;
    const int p = blockIdx.y, c = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *ls = a.layer_start + (int64_t)p * (kDetMaxLayers + 1);
    const int n = ls[a.dg->n_layers];
    if (c * kEmitChunk >= n) return;
    if (wave == 0) {  // emitted candidates in the chunks before this one
        int before = 0;
        for (int k = lane; k < c; k += 64) before += a.emit_chunks[(int64_t)p * a.emit_chunk_cap + k];
        before = wave_sum(before);
        if (lane == 0) chunk_base = before;
    }
    __syncthreads();
    long long run = a.emit_offsets[p] + chunk_base;
    for (int i0 = c * kEmitChunk; i0 < min(n, (c + 1) * kEmitChunk); i0 += kDetThreads) {
        const int i = i0 + threadIdx.x;
        const int64_t ci = (int64_t)p * a.cand_cap + i;
        const bool e = i < n && a.cand_emit[ci];
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
                if (a.out_layer) a.out_layer[o] = layer_of(ls, a.dg->n_layers, i);
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
        hipLaunchKernelGGL(det_score_kernel, grid, dim3(kDetThreads), 0, s, a, l);
    }
    return (int)hipGetLastError();
}

int launch_det_keypoints(const DetArgs &a, int64_t *running, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(det_scan_kernel, dim3(a.n_pairs), dim3(kDetThreads), 0, s, a);
    hipLaunchKernelGGL(det_candidates_kernel, dim3((a.g.total_rows + 3) / 4, a.n_pairs), dim3(kDetThreads), 0, s, a);
    if (a.fp_x87)
        hipLaunchKernelGGL(det_refine_kernel<true>, dim3((a.cand_cap + kRefineChunk - 1) / kRefineChunk, a.n_pairs), dim3(kDetThreads), 0, s, a);
    else
        hipLaunchKernelGGL(det_refine_kernel<false>, dim3((a.cand_cap + kRefineChunk - 1) / kRefineChunk, a.n_pairs), dim3(kDetThreads), 0, s, a);
    // ties: layer by layer (a layer's ties read what the maxima of the layer below asked for in it)
    for (int l = 0; l < a.g.n_layers; ++l) {
        int32_t *waiting = a.tie_waiting + (int64_t)l * a.n_pairs;
        hipLaunchKernelGGL(det_tie_first_kernel, dim3(kTieGroups, a.n_pairs), dim3(kTieThreads), 0, s, a, l, waiting);
        hipLaunchKernelGGL(det_tie_chain_kernel, dim3(a.n_pairs), dim3(kTieThreads), 0, s, a, l, waiting);
    }
    const dim3 egrid((a.cand_cap + kEmitChunk - 1) / kEmitChunk, a.n_pairs);
    hipLaunchKernelGGL(det_emit_count_kernel, egrid, dim3(kDetThreads), 0, s, a);
    hipLaunchKernelGGL(det_emit_scan_kernel, dim3(1), dim3(kDetThreads), 0, s, a, running);
    hipLaunchKernelGGL(det_emit_scatter_kernel, egrid, dim3(kDetThreads), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace mofreak
