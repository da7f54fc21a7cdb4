// Fused tile kernel of the MoFREAK path for gfx950 (CDNA4): one 512-thread workgroup owns a 96x64-pixel tile of one
// frame pair and describes every keypoint whose pixel falls in it, entirely out of LDS.  Two workgroups share a CU
// (75 KB of LDS each), so one workgroup's barriers and global-memory latencies are covered by the other's work.
//
//   stage 0  gray tiles: the tile + 8-px rim of `current` and `previous` (u8, 112x80 each) -> LDS
//   stage 1  MIP (MoFREAKUtilities.cpp:288-325, 46-99), one wave per keypoint, no workgroup barrier: the ~280 pixels of
//            the two 19x19 resamples that motionInterchangePattern actually reads (cv::resize fixed-point bilinear,
//            host-built sample table per ROI side held in registers) -> the wave's own LDS buffer; then
//            lane = 8*centre + offset, strip SSD, __ballot = the 8 motion bytes
//   stage 2  integral image of |current - previous| over tile + halo, kept MODULO 2^16 (u16, two pixels per dword):
//            the row pass scans inside a wave (v_sad_u8 inside a lane's 16 pixels, DPP row_shr across the 16 lanes
//            of a region row), the column pass adds packed pairs (v_pk_add_u16).  A box sum is exact modulo 2^16 as
//            long as the box holds at most 257 pixels (257 * 255 < 2^16); larger boxes are summed in horizontal
//            slices of at most 257 pixels each.  Box sums are translation-invariant, so the tile-local integral gives
//            the same box means as cv::integral of the whole frame -- which never exists in HBM.
//            The halo is sized per call from the largest FREAK pattern among the call's tile-path keypoints
//            (binning pass, device-resident word): 24, 32, 40 or 48 pixels.
//   stage 3  FREAK (cv::FREAK::compute on the difference image, :427-428), one wave per group of four keypoints, no
//            workgroup barrier: 43 box means per keypoint (172 tasks over three 64-lane passes), orientation with
//            16 lanes per keypoint (DPP row reduction), rotated means, lane = descriptor bit, __ballot = the 8
//            appearance bytes, one 16-byte store per descriptor
//
// HBM traffic is the two frames (halo re-reads are served by L2 / Infinity Cache) + keypoints in + descriptors out.
// Keypoints whose FREAK pattern does not fit the 48-px halo (patternSizes[scale] > 48, i.e. size >= ~14.9) or whose
// ROI does not fit the rim are left to the gather path (describe_kernel over a global integral) by the binning pass.
#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kTileThreads = 512;                    // 8 waves; two workgroups per CU = 4 waves per SIMD
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kBatch = 96;                           // keypoints described per pass over a tile's list
constexpr int kGroup = 4;                            // keypoints one wave describes together in stage 3
constexpr int kMinHalo = 24;                         // smallest integral halo (patternSizes[0] = 23)
constexpr int kIPitch = kTileRW + 8;                 // LDS integral pitch (u16); logical column c at physical c+7
constexpr int kIColOff = 7;
constexpr int kIPitchDw = kIPitch / 2;
constexpr int kIntegralBytes = (kTileRH + 1) * kIPitch * 2;
constexpr int kRowGroupIters = (kTileRH / 4 + kTileWaves - 1) / kTileWaves;   // 4-row groups per wave in the row pass
constexpr int kColBlockRows = 16;
constexpr int kMaxColBlocks = kTileRH / kColBlockRows;                         // 10
constexpr int kMaxDcols = kTileRW / 2;                                         // 96 dword columns (pixel pairs)
constexpr int kColIters = (kMaxColBlocks * kMaxDcols + kTileThreads - 1) / kTileThreads;
constexpr int kVStride = 44;                         // bytes per keypoint in a wave's box-mean buffer (11 dwords: odd)
constexpr int kMipIters = 5;                         // 64-lane passes over the <= 320 sampled 19x19 positions
constexpr int kBoxIters = (kGroup * kNbPoints + 63) / 64;                       // 3
constexpr int kBigPoints = 12;                       // points of the two outer rings: the boxes that may need slices
constexpr int kGrayTasks = 2 * kTileCH * (kTileCW / 16);
constexpr int kGrayIters = (kGrayTasks + kTileThreads - 1) / kTileThreads;
constexpr int kP19Wave = 2 * 2 * kP19Pad;            // a wave's MIP buffers: two keypoints x (cur19, prev19)

// ---- LDS carve (bytes); every offset is a multiple of 16
constexpr int kOffIntegral = 0;
// stages 0-1 use the (not yet built) integral area: gray tiles and the waves' 19x19 buffers
constexpr int kOffCur = 0;
constexpr int kOffPrev = kOffCur + kTileCW * kTileCH;
constexpr int kOffP19 = kOffPrev + kTileCW * kTileCH;
constexpr int kOffScratch = (kIntegralBytes + 15) / 16 * 16;
constexpr int kScratchBytes = kMaxColBlocks * kMaxDcols * 4;   // column-block totals; stage-1 records; stage-3 box means
constexpr int kOffTheta = kOffScratch + kScratchBytes;
constexpr int kOffKf = kOffTheta + kThetaBounds * (int)sizeof(ThetaBound);   // the batch's FREAK records
constexpr int kOffMot = kOffKf + kBatch * 16;                                // motion bytes kept for the fused store
constexpr int kOffStamps = kOffMot + kBatch * 8;                             // diagnostic build only: 32 x u64
constexpr int kTileLdsBytes = kOffStamps + 256;
static_assert(kOffP19 % 16 == 0 && kP19Wave % 16 == 0 && kOffScratch % 16 == 0 && kOffTheta % 16 == 0, "LDS carve alignment");
static_assert(kOffP19 + kTileWaves * kP19Wave <= kIntegralBytes, "stage-1 buffers fit the integral area");
static_assert(2 * kTileLdsBytes <= 160 * 1024, "two workgroups per CU");
static_assert(kTileRW % 16 == 0 && kTileRH % kColBlockRows == 0 && kTileRW / 16 <= 16, "region blocking");
static_assert((kIPitch * 2) % 16 == 0, "integral rows start on 16 bytes");
static_assert(kTileWaves * kGroup * kVStride <= kScratchBytes, "stage-3 box means fit the scratch area");
static_assert(kBatch % (kGroup * kTileWaves) == 0, "whole groups per wave in a full batch");

struct KpFreak {   // stage 3 per-keypoint record
    float kx, ky;
    int32_t g;
    int16_t idx, theta;
};
struct KpMip {     // stage 1 per-keypoint record
    int32_t g;
    uint16_t roi_off;
    uint8_t L, pad;
};
static_assert(kBatch * (int)sizeof(KpMip) <= kScratchBytes, "stage-1 records fit the scratch area");
static_assert(sizeof(KpFreak) == 16, "record size used by the LDS carve");

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) uint8_t lds_vu8;   // byte loads the optimiser must not fuse

struct __attribute__((aligned(8))) Px16 {
    uint32_t w[4];
};

// 16 pixels of one row starting at image column gx (zero outside the image).  gx is a multiple of 8.
__device__ __forceinline__ Px16 load_px16(const uint8_t *row, int gx, int W, bool row_ok, bool fast8)
{
    if (row_ok && fast8 && gx >= 0 && gx + 16 <= W) return *reinterpret_cast<const Px16 *>(row + gx);
    Px16 r = {{0, 0, 0, 0}};
    if (row_ok) {
        for (int k = 0; k < 16; ++k) {
            const int x = gx + k;
            if (x >= 0 && x < W) r.w[k >> 2] |= (uint32_t)row[x] << (8 * (k & 3));
        }
    }
    return r;
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

__device__ __forceinline__ int dpp_row_shr(int v, int n)
{
    switch (n) {  // bound_ctrl: lanes shifted in from outside the 16-lane row read 0
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    }
}
__device__ __forceinline__ int dpp_row_ror(int v, int n)
{
    switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    }
}
// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ int row16_sum(int v)
{
    v += dpp_row_ror(v, 8);
    v += dpp_row_ror(v, 4);
    v += dpp_row_ror(v, 2);
    v += dpp_row_ror(v, 1);
    return v;
}

__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}

// FREAK::meanIntensity (box branch) on the tile-local integral modulo 2^16.  `ibase` = LDS byte address of logical
// (row 0, column 0) minus the region origin: the u16 of image corner (y, x) sits at ibase + 2 * (y * kIPitch + x).
__device__ __forceinline__ int mean_intensity_tile(const uint8_t *lds, int ibase, float kx, float ky, const PatternPoint P)
{
    const float xf = P.x + kx;
    const float yf = P.y + ky;
    const float radius = P.sigma;
    // int(xf - radius + 0.5), int(xf + radius + 1.5): the reference adds 0.5 / 1.5 in double, i.e. exactly
    const int x_left = round_half_up_pos(xf - radius);
    const int y_top = round_half_up_pos(yf - radius);
    const int x_right = round_half_up_pos(xf + radius) + 1;
    const int y_bottom = round_half_up_pos(yf + radius) + 1;
    const int w = x_right - x_left, h = y_bottom - y_top;
    const int w2 = 2 * w;
    int addr = ibase + 2 * ((int)__umul24(y_top, kIPitch) + x_left);
    // rows per slice: the largest count whose slice stays within 257 pixels (floor(257 / w); 257 is prime, so the
    // quotient is never within 1/64 of an integer and the float reciprocal cannot land on the wrong side)
    const int rps = max(1, (int)(257.0f * __builtin_amdgcn_rcpf((float)w)));
    int prev = (int)*reinterpret_cast<const uint16_t *>(lds + addr + w2) - (int)*reinterpret_cast<const uint16_t *>(lds + addr);
    int step = min(rps, h);
    addr += (int)__umul24(step, 2 * kIPitch);
    int cur = (int)*reinterpret_cast<const uint16_t *>(lds + addr + w2) - (int)*reinterpret_cast<const uint16_t *>(lds + addr);
    int sum = (cur - prev) & 0xffff;
    int left = h - step;
    while (left > 0) {  // outer rings of the larger patterns only
        prev = cur;
        step = min(rps, left);
        addr += (int)__umul24(step, 2 * kIPitch);
        cur = (int)*reinterpret_cast<const uint16_t *>(lds + addr + w2) - (int)*reinterpret_cast<const uint16_t *>(lds + addr);
        sum += (cur - prev) & 0xffff;
        left -= step;
    }
    return div_box_small(sum, (int)__umul24(w, h)) & 0xff;
}

// ------------------------------------------------------------------------------------------------
// binning
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int scale_index_scalar(const SmallTables *st, float size)
{
    if (!st->scale_normalized) return st->fixed_scale_index;
    int lo = 0, hi = kNbScales - 1;  // number of thresholds <= size (thresholds ascend)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (size >= st->scale_thresholds[mid])
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int pair_of(const int64_t *offs, int n_pairs, int64_t g)
{
    int lo = 0, hi = n_pairs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Pass 1: classify every keypoint (erased / tile path / gather path) and count tile populations.
__global__ __launch_bounds__(256) void bin_count_kernel(BinArgs a)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_kp) return;
    const SmallTables *st = a.small;
    const mofreak_keypoint kp = a.kps[g];
    const float kx = kp.x, ky = kp.y, size = kp.size;
    // DescriptorExtractor::compute + FREAK::computeImpl keypoint filter (same tests as describe_kernel)
    bool ok = (size >= FLT_EPSILON) && (size <= FLT_MAX) && (fabsf(kx) <= FLT_MAX) && (fabsf(ky) <= FLT_MAX);
    const int idx = ok ? scale_index_scalar(st, size) : 0;
    const int ps = st->pattern_sizes[idx];
    if (kx <= ps || ky <= ps || kx >= a.W - ps || ky >= a.H - ps) ok = false;
    int key = -1;
    int tile_ps = 0;
    if (ok) {
        const int x_i = (int)kx, y_i = (int)ky;
        const int half = ((int)size) / 2, L = (int)ceilf(size);
        const bool roi_in = (x_i - half >= 0) && (y_i - half >= 0) && (x_i - half + L <= a.W) && (y_i - half + L <= a.H);
        const bool fast = !a.force_slow && ps <= kTileHalo && L <= kTileMaxRoi && half <= kTileMipHalo &&
                          (L - half) <= kTileMipHalo + 1 && roi_in;
        if (fast) {
            const int tile = (y_i / kTileH) * a.tiles_x + (x_i / kTileW);
            const int64_t k64 = a.kp_offsets ? (int64_t)pair_of(a.kp_offsets, a.n_pairs, g) * (a.tiles_x * a.tiles_y) + tile : tile;
            key = (int)k64;
            tile_ps = ps;
            atomicAdd(&a.tile_start[key], 1);
            atomicMin(&a.tile_lmin[key], (uint32_t)L);
            atomicMax(&a.tile_lmax[key], (uint32_t)L);
        } else {
            key = -2;
        }
    }
    a.kp_key[g] = key;
    // how many keypoints already need the gather path: pass 2 decides from it whether thin tiles follow them there
    const unsigned long long slow = __ballot(key == -2);
    if (slow && lane_id() == __ffsll((long long)slow) - 1) atomicAdd(a.slow_count, __popcll(slow));
    // the largest pattern on the tile path sizes the tile kernel's integral halo
    int m = tile_ps;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (m > 0 && lane_id() == 0) atomicMax(a.max_ps, m);
}

// A tile costs the same whether it holds 5 keypoints or 90 (gray tiles, the integral), the gather path costs per
// keypoint.  When the gather path runs anyway for a good share of the call (large keypoints: a detector's output),
// thinly populated tiles are cheaper there; on dense grids nothing changes.
constexpr int kSparseTile = 16;  // keypoints below which a tile is handed to the gather path
constexpr int kSparseMarker = -(1 << 30);

// Pass 2: exclusive scan of the tile populations (single workgroup; n_keys is a few hundred to ~1e5).
__global__ __launch_bounds__(256) void bin_scan_kernel(int32_t *tile_start, int32_t *tile_cursor, int32_t *slow_count, int64_t n_kp,
                                                       int64_t n_keys)
{
    __shared__ int carry_s;
    __shared__ int wave_tot[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int n_slow = *slow_count;  // counted by pass 1; pass 3 counts again while it fills the list
    const bool drop_sparse = n_slow > 0 && (int64_t)n_slow * 8 >= n_kp;
    __syncthreads();
    if (threadIdx.x == 0) *slow_count = 0;
    for (int64_t b0 = 0; b0 < n_keys; b0 += 256) {
        const int64_t b = b0 + threadIdx.x;
        int v = b < n_keys ? tile_start[b] : 0;
        if (drop_sparse && v > 0 && v < kSparseTile) {
            v = 0;
            tile_cursor[b] = kSparseMarker;  // pass 3 sends this tile's keypoints to the slow list
        }
        const int incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int base = carry_s;
        for (int i = 0; i < w; ++i) base += wave_tot[i];
        if (b < n_keys) tile_start[b] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_start[n_keys] = carry_s;
}

// Pass 3: scatter keypoints into their tile's segment / the slow list; finalise erased keypoints.
__global__ __launch_bounds__(256) void bin_scatter_kernel(BinArgs a)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_kp) return;
    const int key = a.kp_key[g];
    if (key >= 0 && a.tile_cursor[key] < 0) {
        a.slow_list[atomicAdd(a.slow_count, 1)] = (int)g;
    } else if (key >= 0) {
        const int pos = a.tile_start[key] + atomicAdd(&a.tile_cursor[key], 1);
        const mofreak_keypoint kp = a.kps[g];
        SortedKp s;
        s.x = kp.x;
        s.y = kp.y;
        s.packed = (uint32_t)(int)ceilf(kp.size) | ((uint32_t)(((int)kp.size) / 2) << 8) |
                   ((uint32_t)scale_index_scalar(a.small, kp.size) << 16);  // :293-295 ROI side / half, FREAK scale index
        s.g = (int)g;
        a.sorted_kp[pos] = s;
    } else if (key == -2) {
        a.slow_list[atomicAdd(a.slow_count, 1)] = (int)g;
    } else {
        // erased: zero descriptor, valid = 0, in every pair that lists this keypoint
        const int reps = a.kp_offsets ? 1 : a.n_pairs;
        const SmallTables *st = a.small;
        for (int p = 0; p < reps; ++p) {
            const int64_t out_idx = a.kp_offsets ? g : (int64_t)p * a.n_kp + g;
            *reinterpret_cast<uint4 *>(a.out_desc + out_idx * 16) = make_uint4(0, 0, 0, 0);
            a.out_valid[out_idx] = 0;
            if (a.out_info) {
                const float size = a.kps[g].size;
                const bool fin = (size >= FLT_EPSILON) && (size <= FLT_MAX);
                *reinterpret_cast<int4 *>(a.out_info + out_idx * 4) = make_int4(fin ? scale_index_scalar(st, size) : 0, -1, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the tile kernel
// ------------------------------------------------------------------------------------------------
// Diagnostic build of the same kernel (STAMPS = true, launched only when the context was created with
// MOFREAK_TILE_STAMPS=1): thread 0 of every workgroup adds the s_memtime ticks it spent between consecutive
// stamps into a.stamps[phase].  In the product instantiation no stamp executes.
#define TILE_STAMP(i)                                                               \
    do {                                                                            \
        if (STAMPS && tid == 0) {                                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
            s_stamps[i] += now_ - last_stamp; /* LDS; flushed to memory at the end */ \
            last_stamp = now_;                                                      \
        }                                                                           \
    } while (0)

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

template <bool STAMPS>
__global__ __launch_bounds__(kTileThreads, 4) void tile_kernel(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    unsigned long long *s_stamps = reinterpret_cast<unsigned long long *>(lds + kOffStamps);
    if (STAMPS && threadIdx.x == 0)
        for (int i = 0; i < kTileStampSlots; ++i) s_stamps[i] = 0;
    unsigned long long last_stamp = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
    const int n_tiles = a.tiles_x * a.tiles_y;
    // 1-D grid, remapped so that each XCD (workgroups are dealt round-robin over the 8 XCDs) walks a contiguous range
    // of (pair, tile) work items: neighbouring tiles share halo pixels, and this way they share an L2.  Placement is
    // a speed matter only; nothing below depends on it.
    const int n_work = n_tiles * a.n_pairs;
    const int per_xcd = (n_work + 7) / 8;
    const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (work >= n_work) return;
    const int pair = work / n_tiles, tile = work - pair * n_tiles;
    const int key = a.kp_offsets ? pair * n_tiles + tile : tile;
    const int kp_begin = a.tile_start[key];
    const int n_tile_kp = a.tile_start[key + 1] - kp_begin;
    if (n_tile_kp == 0) return;

    uint8_t *scratch = lds + kOffScratch;
    lds_vu8 *ldsv = (lds_vu8 *)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.f.W, H = a.f.H;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    // integral halo of this call: the largest pattern the binning pass met, in steps of 8 pixels
    const int halo = min(kTileHalo, max(kMinHalo, (*a.max_ps + 7) & ~7));
    const int RW = kTileW + 2 * halo, RH = kTileH + 2 * halo;
    const int ox = tx * kTileW - halo, oy = ty * kTileH - halo;                    // integral region origin
    const int cx0 = tx * kTileW - kTileMipHalo, cy0 = ty * kTileH - kTileMipHalo;  // gray tile origin
    const uint8_t *cur = a.f.cur + (int64_t)pair * a.f.pair_stride;
    const uint8_t *prev = a.f.prev + (int64_t)pair * a.f.pair_stride;
    const bool fast8 = (((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)a.f.row_stride) & 7) == 0;
    const int64_t out_base = a.kp_offsets ? 0 : (int64_t)pair * a.n_kp;
    const SortedKp *tile_kps = a.sorted_kp + kp_begin;
    const SmallTables *st = a.small;
    const int bit_mode = st->bit_mode, mip_theta = st->mip_theta;
    const bool orientation_normalized = st->orientation_normalized != 0;

    ThetaBound *s_theta = reinterpret_cast<ThetaBound *>(lds + kOffTheta);
    if (tid < kThetaBounds) s_theta[tid] = a.theta[tid];

    uint2 *s_mot = reinterpret_cast<uint2 *>(lds + kOffMot);
    KpMip *km = reinterpret_cast<KpMip *>(scratch);            // stage-1 records (scratch area: gone once stage 2 starts)
    KpFreak *kf = reinterpret_cast<KpFreak *>(lds + kOffKf);   // stage-3 records (outside the scratch area)
    const bool one_batch = n_tile_kp <= kBatch;
    // The binning pass recorded the smallest and largest ROI side of the tile: equal in the usual case.
    const int tile_L = (int)a.tile_lmin[key];
    const bool uniform = tile_L == (int)a.tile_lmax[key];

    // Both stages' per-keypoint records from one binned record (ROI corner / side for the MIP; coordinates, scale
    // index for FREAK).
    auto make_records = [&](int b0, int nb) {
        if (tid < nb) {
            const SortedKp kp = tile_kps[b0 + tid];
            const int x_i = (int)kp.x, y_i = (int)kp.y;  // :460 float -> int parameters
            const int L = (int)(kp.packed & 0xff), half = (int)((kp.packed >> 8) & 0xff);
            KpMip m;
            m.g = kp.g;
            m.roi_off = (uint16_t)((y_i - half - cy0) * kTileCW + (x_i - half - cx0));
            m.L = (uint8_t)L;
            m.pad = 0;
            km[tid] = m;
            KpFreak k;
            k.kx = kp.x;
            k.ky = kp.y;
            k.g = kp.g;
            k.idx = (int16_t)(kp.packed >> 16);
            k.theta = 0;
            kf[tid] = k;
        }
    };

    // per-lane constants of the MIP sampling passes: where each sampled pixel goes, and (once the ROI side is known)
    // the LDS byte offsets of its two source rows relative to the ROI origin, frame base included (the second byte of
    // a row pair is the next one: where cv::resize clamps the column instead, its weight is zero).  Their loads are
    // issued here, ahead of stage 0, so that the global latencies overlap.
    int pos[kMipIters], a0[kMipIters], a1[kMipIters];
    uint32_t cxp[kMipIters];                   // the two 11-bit x weights, packed as loaded (c0 | c1 << 16)
    uint32_t c0ys[kMipIters], c1ys[kMipIters];  // the y weights << 12: (w * (t >> 4)) >> 16 == mul_hi_u24(t & ~15, w << 12)
#pragma unroll
    for (int u = 0; u < kMipIters; ++u) pos[u] = kOffP19 + wave * kP19Wave + a.mip_pos[min(lane + 64 * u, a.mip_stride - 1)];
    auto load_samples = [&](int L) {
        const MipSample *tab = a.mip_samples + (int64_t)L * a.mip_stride;
#pragma unroll
        for (int u = 0; u < kMipIters; ++u) {
            const MipSample sm = tab[min(lane + 64 * u, a.mip_stride - 1)];
            const int frame = (lane + 64 * u) < a.mip_n_cur ? kOffCur : kOffPrev;
            a0[u] = frame + sm.off00;
            a1[u] = frame + sm.off10;
            cxp[u] = (uint32_t)(uint16_t)sm.c0x | (uint32_t)(uint16_t)sm.c1x << 16;
            c0ys[u] = (uint32_t)(uint16_t)sm.c0y << 12;
            c1ys[u] = (uint32_t)(uint16_t)sm.c1y << 12;
        }
    };
    int have_L = -1;
    if (uniform) {  // one ROI side in the whole tile (the usual case): its samples stay in registers
        have_L = tile_L;
        load_samples(tile_L);
    }

    // ================= stage 0: gray tiles (tile + 8-px rim), 16 bytes per lane; all loads first, then the stores.
    // The first batch's keypoint records ride along.
    {
        Px16 v[kGrayIters];
#pragma unroll
        for (int u = 0; u < kGrayIters; ++u) {
            const int t = tid + u * kTileThreads;
            const int fr = t >= kGrayTasks / 2 ? 1 : 0, tt = t - fr * (kGrayTasks / 2);
            const int r = tt / (kTileCW / 16), q = tt - r * (kTileCW / 16);
            const int gy = cy0 + r, gx = cx0 + 16 * q;
            const bool row_ok = t < kGrayTasks && gy >= 0 && gy < H;
            const int64_t ro = (int64_t)gy * a.f.row_stride;
            v[u] = load_px16((fr ? prev : cur) + ro, gx, W, row_ok, fast8);
        }
        make_records(0, min(kBatch, n_tile_kp));
#pragma unroll
        for (int u = 0; u < kGrayIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < kGrayTasks)  // cur tile, then prev tile: contiguous in LDS
                reinterpret_cast<uint4 *>(lds + kOffCur)[t] = make_uint4(v[u].w[0], v[u].w[1], v[u].w[2], v[u].w[3]);
        }
    }
    __syncthreads();  TILE_STAMP(0);

    // ================= stage 1: MIP
    {
        uint8_t *p19 = lds + kOffP19 + wave * kP19Wave;  // this wave's two pairs of 19x19 buffers
        // per-lane constants of the bit pass: lane = 8*centre + offset (MoFREAKUtilities.cpp:56-70, 308-316)
        const int mc = lane >> 3, mi = lane & 7;
        const int mcx = (0xDDD99555u >> (4 * mc)) & 15, mcy = (0xD95D5D95u >> (4 * mc)) & 15;
        const int mdx = (int)((0x14787410u >> (4 * mi)) & 15) - 4, mdy = (int)((0x10147874u >> (4 * mi)) & 15) - 4;
        const int base_c = (mcy - 1) * kPatch + (mcx - 1);
        const int base_p = kP19Pad + (mcy + mdy - 1) * kPatch + (mcx + mdx - 1);
        const int cw = base_c >> 2, cs = base_c & 3, pw = base_p >> 2, ps = base_p & 3;  // covering dword, byte shift
        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (b0 > 0) {  // further batches of a crowded tile: their records
                __syncthreads();
                make_records(b0, nb);
                __syncthreads();
            }
            // Per wave, two keypoints at a time (their LDS reads are issued together, so one's latency hides under
            // the other's arithmetic): the sampled pixels of the two 19x19 resamples -> LDS, then -- same wave, so
            // only a wave-level sync -- lane = 8*centre + offset, strip SSD, ballot.
            for (int kk = wave; kk < nb; kk += 2 * kTileWaves) {
                const int kk2 = kk + kTileWaves;
                const bool two = kk2 < nb;
                const KpMip m = km[kk], m2 = km[two ? kk2 : kk];
                if (!uniform) {  // mixed ROI sides: rare; one keypoint at a time, reloading the samples when the side changes
                    for (int h = 0; h < (two ? 2 : 1); ++h) {
                        const KpMip mm = h ? m2 : m;
                        if (mm.L != have_L) {
                            have_L = mm.L;
                            load_samples(mm.L);
                        }
#pragma unroll
                        for (int u = 0; u < kMipIters; ++u) {
                            const int b0a = a0[u] + mm.roi_off, b1a = a1[u] + mm.roi_off;
                            u16x2 r0, r1;
                            r0.x = ldsv[b0a];
                            r0.y = ldsv[b0a + 1];
                            r1.x = ldsv[b1a];
                            r1.y = ldsv[b1a + 1];
                            const u16x2 wx = __builtin_bit_cast(u16x2, cxp[u]);
                            const uint32_t t0 = __builtin_amdgcn_udot2(r0, wx, 0u, false), t1 = __builtin_amdgcn_udot2(r1, wx, 0u, false);
                            const int px = resize_y(t0, t1, c0ys[u], c1ys[u]);
                            if (lane + 64 * u < a.mip_n) lds[pos[u] + h * (2 * kP19Pad)] = (uint8_t)px;
                        }
                    }
                } else {
                    u16x2 r0[2][kMipIters], r1[2][kMipIters];
#pragma unroll
                    for (int u = 0; u < kMipIters; ++u) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int roi = h ? m2.roi_off : m.roi_off;
                            const int b0a = a0[u] + roi, b1a = a1[u] + roi;
                            r0[h][u].x = ldsv[b0a];
                            r0[h][u].y = ldsv[b0a + 1];
                            r1[h][u].x = ldsv[b1a];
                            r1[h][u].y = ldsv[b1a + 1];
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int u = 0; u < kMipIters; ++u) {
                            const u16x2 wx = __builtin_bit_cast(u16x2, cxp[u]);
                            const uint32_t t0 = __builtin_amdgcn_udot2(r0[h][u], wx, 0u, false);
                            const uint32_t t1 = __builtin_amdgcn_udot2(r1[h][u], wx, 0u, false);
                            const int px = resize_y(t0, t1, c0ys[u], c1ys[u]);
                            if ((h == 0 || two) && lane + 64 * u < a.mip_n) lds[pos[u] + h * (2 * kP19Pad)] = (uint8_t)px;
                        }
                    }
                }
                wave_lds_sync();
                // the two 9-byte strips sit at arbitrary byte offsets: fetch the covering aligned dwords (a byte-wise
                // formulation lets the compiler fuse the loads into misaligned ds_read_b64s, 64 cycles each) and
                // shift the strips out; SSD = sum c^2 + sum p^2 - 2 sum c*p over the first eight bytes (packed u8
                // dot products) + the ninth byte's squared difference
                uint32_t cd[2][3], pd[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t *b32 = reinterpret_cast<const uint32_t *>(p19 + ((h && two) ? 2 * kP19Pad : 0));
#pragma unroll
                    for (int w3 = 0; w3 < 3; ++w3) {
                        cd[h][w3] = b32[cw + w3];
                        pd[h][w3] = b32[pw + w3];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t c0 = __builtin_amdgcn_alignbyte(cd[h][1], cd[h][0], cs), c1 = __builtin_amdgcn_alignbyte(cd[h][2], cd[h][1], cs);
                    const uint32_t p0 = __builtin_amdgcn_alignbyte(pd[h][1], pd[h][0], ps), p1 = __builtin_amdgcn_alignbyte(pd[h][2], pd[h][1], ps);
                    const int d8 = (int)((cd[h][2] >> (8 * cs)) & 0xffu) - (int)((pd[h][2] >> (8 * ps)) & 0xffu);
                    const uint32_t sq = __builtin_amdgcn_udot4(c0, c0, __builtin_amdgcn_udot4(c1, c1, (uint32_t)__mul24(d8, d8), false), false) +
                                        __builtin_amdgcn_udot4(p0, p0, __builtin_amdgcn_udot4(p1, p1, 0u, false), false);
                    const uint32_t cross = __builtin_amdgcn_udot4(c0, p0, __builtin_amdgcn_udot4(c1, p1, 0u, false), false);
                    const int ssd = (int)(sq - 2u * cross);
                    const uint64_t mot = __ballot(ssd > mip_theta);
                    if (lane == 0 && (h == 0 || two)) {
                        const uint2 mv = make_uint2((uint32_t)mot, (uint32_t)(mot >> 32));
                        if (one_batch)  // kept for one 16-byte store per descriptor at the end of stage 3
                            s_mot[h ? kk2 : kk] = mv;
                        else  // a crowded tile (several batches) sends its motion bytes out now
                            *reinterpret_cast<uint2 *>(a.out_desc + (out_base + (h ? m2.g : m.g)) * 16 + 8) = mv;
                    }
                }
                wave_lds_sync();  // the next pair of keypoints overwrites the 19x19 buffers
            }
        }
    }
    __syncthreads();  TILE_STAMP(1);

    // ================= stage 2: integral of |cur - prev| over tile + halo, modulo 2^16, in LDS
    {
        // 2a: row pass, no workgroup barrier.  The 16 lanes of a DPP row share one region row: a lane owns 16 pixels,
        //     |cur - prev| and the running sum inside them come from v_sad_u8 on masked dwords, the lane totals are
        //     scanned across the row with four DPP adds.  A wave takes four region rows per step.
        const int runs = RW >> 4;
        const int rr = lane >> 4, q = lane & 15;
        Px16 c[kRowGroupIters], p[kRowGroupIters];
#pragma unroll
        for (int u = 0; u < kRowGroupIters; ++u) {
            const int r = 4 * (wave + kTileWaves * u) + rr;
            const int gy = oy + r, gx = ox + 16 * q;
            const bool row_ok = r < RH && q < runs && gy >= 0 && gy < H;
            const int64_t ro = (int64_t)gy * a.f.row_stride;
            c[u] = load_px16(cur + ro, gx, W, row_ok, fast8);
            p[u] = load_px16(prev + ro, gx, W, row_ok, fast8);
        }
        for (int i = tid; i < kIPitchDw; i += kTileThreads) reinterpret_cast<uint32_t *>(lds + kOffIntegral)[i] = 0;  // integral row 0
#pragma unroll
        for (int u = 0; u < kRowGroupIters; ++u) {
            const int r = 4 * (wave + kTileWaves * u) + rr;
            uint32_t pk[8];
            uint32_t acc = 0;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                const uint32_t x = c[u].w[w4], y = p[u].w[w4];
                const uint32_t s0 = __builtin_amdgcn_sad_u8(x & 0xffu, y & 0xffu, acc);
                const uint32_t s1 = __builtin_amdgcn_sad_u8(x & 0xffffu, y & 0xffffu, acc);
                const uint32_t s2 = __builtin_amdgcn_sad_u8(x & 0xffffffu, y & 0xffffffu, acc);
                acc = __builtin_amdgcn_sad_u8(x, y, acc);
                pk[2 * w4] = __builtin_amdgcn_perm(s1, s0, 0x05040100u);      // low halves: s0 | s1 << 16
                pk[2 * w4 + 1] = __builtin_amdgcn_perm(acc, s2, 0x05040100u);
            }
            int incl = (int)acc;  // inclusive scan of the lane totals across the region row (16 x 255 x 12 < 2^16)
            incl += dpp_row_shr(incl, 1);
            incl += dpp_row_shr(incl, 2);
            incl += dpp_row_shr(incl, 4);
            incl += dpp_row_shr(incl, 8);
            const uint32_t excl = (uint32_t)incl - acc;
            const uint32_t carry2 = __builtin_amdgcn_perm(excl, excl, 0x05040504u);  // low half in both halves
            if (r < RH && q < runs) {
                uint8_t *row = lds + kOffIntegral + (r + 1) * (kIPitch * 2);
                uint4 *dst = reinterpret_cast<uint4 *>(row + (8 + 16 * q) * 2);
                dst[0] = make_uint4(pk_add_u16(pk[0], carry2), pk_add_u16(pk[1], carry2), pk_add_u16(pk[2], carry2), pk_add_u16(pk[3], carry2));
                dst[1] = make_uint4(pk_add_u16(pk[4], carry2), pk_add_u16(pk[5], carry2), pk_add_u16(pk[6], carry2), pk_add_u16(pk[7], carry2));
                if (q == 0) *reinterpret_cast<uint32_t *>(row + (kIColOff - 1) * 2) = 0;  // logical column 0
            }
        }
        __syncthreads();  TILE_STAMP(2);
        // 2b: column pass.  A thread owns a 16-row segment of one dword column (two pixels): running packed sums in
        //     registers, segment total -> LDS; after the barrier it adds the totals of the segments above.
        const int n_dcols = RW >> 1, n_blocks = RH / kColBlockRows;
        const uint32_t m20 = (1u << 20) / (uint32_t)n_dcols + 1;  // t / n_dcols for t < 2^10 as (t * m20) >> 20
        uint32_t *carry = reinterpret_cast<uint32_t *>(scratch);  // [block][kMaxDcols]
        uint32_t cv[kColIters][kColBlockRows];
#pragma unroll
        for (int u = 0; u < kColIters; ++u) {
            const int t = tid + u * kTileThreads;
            const bool ok = t < n_blocks * n_dcols;
            const int j = ok ? (int)(((uint32_t)t * m20) >> 20) : 0, cc = ok ? t - j * n_dcols : 0;
            const uint32_t *e = reinterpret_cast<const uint32_t *>(lds + kOffIntegral) + (j * kColBlockRows + 1) * kIPitchDw + (kIColOff + 1) / 2 + cc;
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) cv[u][r] = e[r * kIPitchDw];
            uint32_t acc = 0;
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) {
                acc = pk_add_u16(acc, cv[u][r]);
                cv[u][r] = acc;
            }
            if (ok) carry[j * kMaxDcols + cc] = acc;
        }
        __syncthreads();  TILE_STAMP(3);
#pragma unroll
        for (int u = 0; u < kColIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < n_blocks * n_dcols) {
                const int j = (int)(((uint32_t)t * m20) >> 20), cc = t - j * n_dcols;
                uint32_t tot[kMaxColBlocks - 1];
#pragma unroll
                for (int jj = 0; jj < kMaxColBlocks - 1; ++jj) tot[jj] = carry[jj * kMaxDcols + cc];
                uint32_t add = 0;
#pragma unroll
                for (int jj = 0; jj < kMaxColBlocks - 1; ++jj) add = pk_add_u16(add, jj < j ? tot[jj] : 0u);
                uint32_t *e = reinterpret_cast<uint32_t *>(lds + kOffIntegral) + (j * kColBlockRows + 1) * kIPitchDw + (kIColOff + 1) / 2 + cc;
#pragma unroll
                for (int r = 0; r < kColBlockRows; ++r) e[r * kIPitchDw] = pk_add_u16(cv[u][r], add);
            }
        }
        __syncthreads();  TILE_STAMP(4);
    }

    // ================= stage 3: FREAK on the difference image, one wave per group of four keypoints
    {
        uint8_t *vv = scratch + wave * (kGroup * kVStride);   // this wave's box means [kGroup][kVStride]
        const int ibase = kOffIntegral + 2 * (kIColOff - oy * kIPitch - ox);
        // box-mean tasks of a group: the outer two rings (whose boxes may need slices) of all four keypoints first
        int task_kq[kBoxIters], task_p[kBoxIters];
#pragma unroll
        for (int u = 0; u < kBoxIters; ++u) {
            const int t = lane + 64 * u;
            if (t < kGroup * kBigPoints) {
                task_kq[u] = t / kBigPoints;
                task_p[u] = t % kBigPoints;
            } else {
                const int t2 = t - kGroup * kBigPoints;
                task_kq[u] = t2 / (kNbPoints - kBigPoints);
                task_p[u] = kBigPoints + t2 % (kNbPoints - kBigPoints);
            }
            if (t >= kGroup * kNbPoints) task_kq[u] = -1;
        }
        // orientation pass: 16 lanes per keypoint, three of the 45 pairs each; weights as floats (w / 2048 is exact, and
        // so is its product with a difference of two bytes), truncated like the reference's integer division
        const int oq = lane >> 4, osub = lane & 15;
        int opi[3], opj[3];
        float owx[3], owy[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int m = osub + 16 * k;
            const OrientPair op = st->orient[min(m, kNbOrientPairs - 1)];
            opi[k] = op.i;
            opj[k] = op.j;
            owx[k] = m < kNbOrientPairs ? (float)op.weight_dx * (1.0f / 2048.0f) : 0.0f;
            owy[k] = m < kNbOrientPairs ? (float)op.weight_dy * (1.0f / 2048.0f) : 0.0f;
        }
        const int pi = st->bit_pair_i[lane], pj = st->bit_pair_j[lane];
        PatternPoint P0[kBoxIters];   // un-rotated pattern points of this lane's tasks, cached per scale index
        int have_idx = -1;

        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (!one_batch) {  // crowded tile: the records of this batch (a single batch still has them from stage 0;
                __syncthreads();  // km is rewritten too, harmlessly: the scratch area's head is free between batches)
                make_records(b0, nb);
                __syncthreads();
            }
            for (int kbase = wave * kGroup; kbase < nb; kbase += kGroup * kTileWaves) {
                KpFreak rec[kBoxIters];
                bool t_ok[kBoxIters];
#pragma unroll
                for (int u = 0; u < kBoxIters; ++u) {
                    t_ok[u] = task_kq[u] >= 0 && kbase + task_kq[u] < nb;
                    rec[u] = kf[t_ok[u] ? kbase + task_kq[u] : kbase];
                }
                if (orientation_normalized) {
                    // F1: un-rotated box means
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        if (rec[u].idx != have_idx)  // (per lane: all of a lane's tasks see the same scale in the usual case)
                            P0[u] = a.lut[(int64_t)rec[u].idx * kNbOrientation * kNbPoints + task_p[u]];
                    }
                    have_idx = rec[kBoxIters - 1].idx == rec[0].idx && rec[1].idx == rec[0].idx ? rec[0].idx : -1;
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        if (t_ok[u])
                            vv[task_kq[u] * kVStride + task_p[u]] = (uint8_t)mean_intensity_tile(lds, ibase, rec[u].kx, rec[u].ky, P0[u]);
                    }
                    wave_lds_sync();
                    // F2: orientation sums, theta
                    {
                        const uint8_t *v = vv + oq * kVStride;
                        int direction0 = 0, direction1 = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float delta = (float)((int)v[opi[k]] - (int)v[opj[k]]);
                            direction0 += (int)(delta * owx[k]);  // C division by 2048: truncates toward zero, per term
                            direction1 += (int)(delta * owy[k]);
                        }
                        direction0 = row16_sum(direction0);
                        direction1 = row16_sum(direction1);
                        const int theta = theta_index(s_theta, direction0, direction1);
                        if (osub == 0 && kbase + oq < nb) {
                            kf[kbase + oq].theta = (int16_t)theta;
                            if (a.out_info)
                                *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kbase + oq].g) * 4) =
                                    make_int4(kf[kbase + oq].idx, theta, direction0, direction1);
                        }
                    }
                    wave_lds_sync();
                } else if (a.out_info && lane < kGroup && kbase + lane < nb) {
                    *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kbase + lane].g) * 4) = make_int4(kf[kbase + lane].idx, 0, 0, 0);
                }
                // F3: box means of the rotated pattern
                {
                    PatternPoint P[kBoxIters];
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        const int theta = kf[t_ok[u] ? kbase + task_kq[u] : kbase].theta;
                        P[u] = a.lut[((int64_t)rec[u].idx * kNbOrientation + theta) * kNbPoints + task_p[u]];
                    }
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        if (t_ok[u])
                            vv[task_kq[u] * kVStride + task_p[u]] = (uint8_t)mean_intensity_tile(lds, ibase, rec[u].kx, rec[u].ky, P[u]);
                    }
                }
                wave_lds_sync();
                // F4: lane = descriptor bit; lane q stores keypoint q's descriptor
                {
                    int va[kGroup], vb[kGroup];
#pragma unroll
                    for (int qq = 0; qq < kGroup; ++qq) {
                        va[qq] = vv[qq * kVStride + pi];
                        vb[qq] = vv[qq * kVStride + pj];
                    }
                    uint2 app = make_uint2(0, 0);
#pragma unroll
                    for (int qq = 0; qq < kGroup; ++qq) {
                        bool bit;
                        if (bit_mode == MOFREAK_BITS_SSE)
                            bit = va[qq] >= vb[qq];
                        else if (bit_mode == MOFREAK_BITS_NATURAL)
                            bit = va[qq] > vb[qq];
                        else
                            bit = (int)(int8_t)va[qq] > (int)(int8_t)vb[qq];
                        const uint64_t bits = __ballot(bit);
                        if (lane == qq) app = make_uint2((uint32_t)bits, (uint32_t)(bits >> 32));
                    }
                    if (lane < kGroup && kbase + lane < nb) {  // descriptor and validity flag out, side by side
                        const int64_t out_idx = out_base + kf[kbase + lane].g;
                        if (one_batch) {
                            const uint2 mot = s_mot[kbase + lane];
                            *reinterpret_cast<uint4 *>(a.out_desc + out_idx * 16) = make_uint4(app.x, app.y, mot.x, mot.y);
                        } else {
                            *reinterpret_cast<uint2 *>(a.out_desc + out_idx * 16) = app;
                        }
                        a.out_valid[out_idx] = 1;
                    }
                }
                wave_lds_sync();  // the next group overwrites the box means
            }
        }
    }
    TILE_STAMP(5);
    if (STAMPS && tid == 0)
        for (int i = 0; i < kTileStampSlots; ++i)
            if (s_stamps[i]) atomicAdd(&a.stamps[i], s_stamps[i]);
}

}  // namespace

int launch_bin(const BinArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(a.tile_start, 0, (size_t)(a.n_keys + 1) * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.tile_cursor, 0, (size_t)a.n_keys * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.tile_lmin, 0xff, (size_t)a.n_keys * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.tile_lmax, 0, (size_t)a.n_keys * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.slow_count, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.max_ps, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    const int blocks = (int)((a.n_kp + 255) / 256);
    if (blocks > 0) hipLaunchKernelGGL(bin_count_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, a.tile_start, a.tile_cursor, a.slow_count, a.n_kp, a.n_keys);
    if (blocks > 0) hipLaunchKernelGGL(bin_scatter_kernel, dim3(blocks), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int launch_tile(const TileArgs &a, void *stream)
{
    if (a.mip_n > 64 * kMipIters || a.mip_stride < a.mip_n) return (int)hipErrorInvalidValue;
    const void *fn = a.stamps ? reinterpret_cast<const void *>(&tile_kernel<true>) : reinterpret_cast<const void *>(&tile_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kTileLdsBytes);
    if (e != hipSuccess) return (int)e;
    const int64_t n_work = (int64_t)a.tiles_x * a.tiles_y * a.n_pairs;
    if (n_work > (int64_t)1 << 28) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(((n_work + 7) / 8) * 8));
    if (a.stamps)
        hipLaunchKernelGGL(tile_kernel<true>, grid, dim3(kTileThreads), kTileLdsBytes, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(tile_kernel<false>, grid, dim3(kTileThreads), kTileLdsBytes, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // namespace mofreak
