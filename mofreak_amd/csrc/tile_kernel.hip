// Fused tile kernel of the MoFREAK path for gfx950 (CDNA4): one 1024-thread workgroup owns a 96x64-pixel tile of
// one frame pair and describes every keypoint whose pixel falls in it, entirely out of LDS.
//
//   stage 0  gray tiles: the tile + 8-px rim of `current` and `previous` (u8, 112x80 each) -> LDS
//   stage 1  MIP (MoFREAKUtilities.cpp:288-325, 46-99): for each keypoint the ~300 pixels of the two 19x19 resamples
//            that motionInterchangePattern actually reads (cv::resize fixed-point bilinear, host-built sample
//            table per ROI side, held in registers across the batch) -> LDS; then lane = 8*centre + offset,
//            strip SSD, __ballot = the 8 motion bytes
//   stage 2  integral image of |current - previous| over tile + 48-px halo (192x160), built in LDS with two-level
//            blocked scans; box sums are translation-invariant, so this tile-local integral gives bit-identical
//            box means to cv::integral of the whole frame -- and the frame-sized integral never exists in HBM
//   stage 3  FREAK (cv::FREAK::compute on the difference image, :427-428): (keypoint, retina point) pairs flattened
//            over all lanes for the 43 box means; orientation with 8 lanes per keypoint; rotated means;
//            lane = descriptor bit, __ballot = the 8 appearance bytes
//
// The kernel runs one workgroup per CU (LDS-bound), so latency is hidden inside the workgroup: 16 waves, and every
// phase issues all of a thread's loads before it consumes any of them.
// HBM traffic is the two frames (halo re-reads are served by L2 / Infinity Cache) + keypoints in + descriptors out.
// Keypoints whose FREAK pattern does not fit the 48-px halo (patternSizes[scale] > 48, i.e. size >= ~14.9) or whose
// ROI does not fit the rim are left to the gather path (describe_kernel over a global integral) by the binning pass.
#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kTileThreads = 1024;                   // 16 waves: 4 per SIMD
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kBatch = 96;                           // keypoints described per pass over a tile's list
constexpr int kIP = kTileRW + 4;                     // LDS integral pitch (int32); logical column c at physical c+3
constexpr int kIntegralInts = (kTileRH + 1) * kIP;   // 161 x 196
constexpr int kRunsPerRow = kTileRW / 16;            // 16-pixel runs per region row
constexpr int kColBlocks = 8;
constexpr int kColBlockRows = kTileRH / kColBlocks;  // 20
constexpr int kVStride = 44;                         // bytes per keypoint in the box-mean array (11 dwords: odd)
constexpr int kMipIters = 5;                         // 64-lane passes over the <= 320 sampled 19x19 positions
constexpr int kBoxIters = (kBatch * kNbPoints + kTileThreads - 1) / kTileThreads;    // box-mean tasks per thread
constexpr int kRunIters = (kTileRH * kRunsPerRow + kTileThreads - 1) / kTileThreads;  // 16-px runs per thread
constexpr int kGrayTasks = kTileCH * (kTileCW / 8);
constexpr int kGrayIters = (kGrayTasks + kTileThreads - 1) / kTileThreads;
constexpr int kOrientLanes = 8;                      // lanes that share one keypoint's 45 orientation pairs

// ---- LDS carve (bytes); every offset is a multiple of 16
constexpr int kOffIntegral = 0;
constexpr int kOffCur = kIntegralInts * 4;
constexpr int kOffPrev = kOffCur + kTileCW * kTileCH;
constexpr int kOffScratch = kOffPrev + kTileCW * kTileCH;
constexpr int kScratchBytes = kTileRH * kRunsPerRow * 4;  // 7680: row carries; also column carries / per-keypoint arrays
constexpr int kOffSmall = kOffScratch + kScratchBytes;
constexpr int kOffTheta = kOffSmall + (int)((sizeof(SmallTables) + 15) / 16 * 16);
constexpr int kOffKf = kOffTheta + kThetaBounds * (int)sizeof(ThetaBound);   // the batch's FREAK records (outlive the scratch area)
constexpr int kOffBits = kOffKf + kBatch * 16;                               // 8 descriptor bytes per keypoint, staged
constexpr int kOffMot = kOffBits + kBatch * 8;                               // motion bytes kept for the fused store
constexpr int kOffStamps = kOffMot + kBatch * 8;                             // diagnostic build only: 32 x u64
constexpr int kTileLdsBytes = kOffStamps + 256;
static_assert(kOffCur % 16 == 0 && kOffPrev % 16 == 0 && kOffScratch % 16 == 0 && kOffSmall % 16 == 0, "LDS carve alignment");
static_assert(kTileLdsBytes <= 160 * 1024, "tile kernel LDS budget");
static_assert(kTileRW % 16 == 0 && kTileRH % kColBlocks == 0, "region blocking");
static_assert(kColBlocks * kTileRW * 4 <= kScratchBytes, "column carries fit the scratch area");
// during stage 1 the (not yet built) integral area holds the 19x19 buffers
constexpr int kOffP19 = 0;
static_assert(kBatch * 2 * kP19Pad <= kOffCur, "stage-1 buffers fit the integral area");

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
static_assert(kBatch * kVStride <= kScratchBytes, "stage-3 box means fit the scratch area");
static_assert(kBatch * (int)sizeof(KpMip) <= kScratchBytes, "stage-1 records fit the scratch area");
static_assert(sizeof(KpFreak) == 16, "record size used by the LDS carve");

// 16 pixels of one row starting at image column gx (zero outside the image).
__device__ __forceinline__ uint4 load_px16(const uint8_t *row, int gx, int W, bool row_ok, bool fast16)
{
    if (row_ok && fast16 && gx >= 0 && gx + 16 <= W) return *reinterpret_cast<const uint4 *>(row + gx);
    uint32_t w[4] = {0, 0, 0, 0};
    if (row_ok) {
        for (int k = 0; k < 16; ++k) {
            const int x = gx + k;
            if (x >= 0 && x < W) w[k >> 2] |= (uint32_t)row[x] << (8 * (k & 3));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint2 load_px8(const uint8_t *row, int gx, int W, bool row_ok, bool fast8)
{
    if (row_ok && fast8 && gx >= 0 && gx + 8 <= W) return *reinterpret_cast<const uint2 *>(row + gx);
    uint32_t w[2] = {0, 0};
    if (row_ok) {
        for (int k = 0; k < 8; ++k) {
            const int x = gx + k;
            if (x >= 0 && x < W) w[k >> 2] |= (uint32_t)row[x] << (8 * (k & 3));
        }
    }
    return make_uint2(w[0], w[1]);
}

// (int)((double)a + 0.5) for a float 0 <= a < 2^23 without leaving single precision: the double sum is exact, so the
// result is trunc(a) plus one when the fraction reaches one half.  (Tile-path taps are always > 0: the keypoint
// passed FREAK's border filter.)
__device__ __forceinline__ int round_half_up_pos(float a)
{
    const int i = (int)a;
    return i + ((a - (float)i) >= 0.5f ? 1 : 0);
}

// Which 16-pixel run of the region a row-pass task owns.  Eight consecutive tasks (the lane group of one
// ds_write_b128) take a 4-row x 2-run cell: rows are kIP = 196 dwords apart (4 mod 32) and runs 16 dwords, so the eight
// 16-byte stores of a group fall into eight different bank quads.  (Row-major task order puts lanes l and l+2 on the
// same banks: a 4-way conflict on every store of the pass, measured at 29 % of the kernel's LDS cycles.)
static_assert(kTileRH % 4 == 0 && (kTileRW / 16) % 2 == 0, "run cells");
__device__ __forceinline__ int run_row(int t) { return 4 * ((t >> 3) / (kTileRW / 32)) + ((t & 7) >> 1); }
__device__ __forceinline__ int run_col(int t) { return 2 * ((t >> 3) % (kTileRW / 32)) + (t & 1); }

// FREAK::meanIntensity (box branch) on the tile-local integral; (ox, oy) = image coordinates of the region origin.
// div_box for the tile path: quotient <= 255 and box area < 2^12 (patterns up to the 48-pixel halo), so the fix-up
// product is a 24-bit multiply (full rate) instead of v_mul_lo_u32
__device__ __forceinline__ int div_box24(int v, int a)
{
    int q = (int)((float)v * __builtin_amdgcn_rcpf((float)a));
    int r = v - __mul24(q, a);
    if (r < 0) {
        --q;
        r += a;
    }
    if (r >= a) ++q;
    return q;
}

__device__ __forceinline__ int mean_intensity_tile(const int32_t *__restrict__ I, int ox, int oy, float kx, float ky,
                                                   const PatternPoint P)
{
    const float xf = P.x + kx;
    const float yf = P.y + ky;
    const float radius = P.sigma;
    // int(xf - radius + 0.5), int(xf + radius + 1.5): the reference adds 0.5 / 1.5 in double, i.e. exactly
    const int x_left = round_half_up_pos(xf - radius) - ox;
    const int y_top = round_half_up_pos(yf - radius) - oy;
    const int x_right = round_half_up_pos(xf + radius) + 1 - ox;
    const int y_bottom = round_half_up_pos(yf + radius) + 1 - oy;
    const int32_t *top = I + __mul24(y_top, kIP) + kIntegralColOffset;
    const int32_t *bot = I + __mul24(y_bottom, kIP) + kIntegralColOffset;
    int ret_val = bot[x_right];
    ret_val -= bot[x_left];
    ret_val += top[x_left];
    ret_val -= top[x_right];
    return div_box24(ret_val, __mul24(x_right - x_left, y_bottom - y_top)) & 0xff;
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
}

// A tile costs the same whether it holds 5 keypoints or 90 (gray tiles, the 192 x 160 integral), the gather path
// costs per keypoint.  When the gather path runs anyway for a good share of the call (large keypoints: a detector's
// output), thinly populated tiles are cheaper there; on dense grids nothing changes.
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
// MOFREAK_TILE_STAMPS=1): thread 0 of every workgroup adds the s_memtime ticks spent between consecutive
// workgroup barriers into a.stamps[phase].  In the product instantiation no stamp executes.
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
__device__ __forceinline__ int resize_y(int t0, int t1, uint32_t b0s, uint32_t b1s)
{
    uint32_t p0, p1;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(p0) : "v"((uint32_t)t0 & ~15u), "v"(b0s));
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(p1) : "v"((uint32_t)t1 & ~15u), "v"(b1s));
    return (int)((p0 + p1 + 2u) >> 2);
}

template <bool STAMPS>
__global__ __launch_bounds__(kTileThreads) void tile_kernel(TileArgs a)
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

    int32_t *I = reinterpret_cast<int32_t *>(lds + kOffIntegral);
    uint8_t *s_cur = lds + kOffCur, *s_prev = lds + kOffPrev;
    uint8_t *scratch = lds + kOffScratch;
    SmallTables &st = *reinterpret_cast<SmallTables *>(lds + kOffSmall);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.f.W, H = a.f.H;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int ox = tx * kTileW - kTileHalo, oy = ty * kTileH - kTileHalo;        // integral region origin
    const int cx0 = tx * kTileW - kTileMipHalo, cy0 = ty * kTileH - kTileMipHalo;  // gray tile origin
    const uint8_t *cur = a.f.cur + (int64_t)pair * a.f.pair_stride;
    const uint8_t *prev = a.f.prev + (int64_t)pair * a.f.pair_stride;
    const bool fast16 = (((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)a.f.row_stride) & 15) == 0;
    const bool fast8 = (((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)a.f.row_stride) & 7) == 0;
    const int64_t out_base = a.kp_offsets ? 0 : (int64_t)pair * a.n_kp;
    const SortedKp *tile_kps = a.sorted_kp + kp_begin;

    for (int i = tid; i < (int)(sizeof(SmallTables) / 4); i += kTileThreads)
        reinterpret_cast<int32_t *>(&st)[i] = reinterpret_cast<const int32_t *>(a.small)[i];
    ThetaBound *s_theta = reinterpret_cast<ThetaBound *>(lds + kOffTheta);
    for (int i = tid; i < kThetaBounds; i += kTileThreads) s_theta[i] = a.theta[i];

    uint2 *s_bits = reinterpret_cast<uint2 *>(lds + kOffBits);
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
    // the LDS byte offsets of its four source bytes relative to the ROI origin, frame base included.  Their loads are
    // issued here, ahead of stage 0, so that the three global latencies overlap.
    int pos[kMipIters], a0[kMipIters], a1[kMipIters], e0[kMipIters], e1[kMipIters];
    uint32_t cxp[kMipIters];                   // the two 11-bit x weights, packed as loaded (c0 | c1 << 16)
    uint32_t c0ys[kMipIters], c1ys[kMipIters];  // the y weights << 12: (w * (t >> 4)) >> 16 == mul_hi_u24(t & ~15, w << 12)
#pragma unroll
    for (int u = 0; u < kMipIters; ++u) pos[u] = a.mip_pos[min(lane + 64 * u, a.mip_stride - 1)];
    auto load_samples = [&](int L) {
        const MipSample *tab = a.mip_samples + (int64_t)L * a.mip_stride;
#pragma unroll
        for (int u = 0; u < kMipIters; ++u) {
            const MipSample sm = tab[min(lane + 64 * u, a.mip_stride - 1)];
            const int frame = (lane + 64 * u) < a.mip_n_cur ? kOffCur : kOffPrev;
            a0[u] = frame + sm.off00;
            a1[u] = frame + sm.off10;
            e0[u] = frame + sm.off01;
            e1[u] = frame + sm.off11;
            // The four bytes are fetched with four ds_read_u8.  Hide from the optimiser that e = a + 1 in most
            // lanes: it would fuse the pairs into ds_read_u16 at odd addresses, which the LDS replays slowly.
            asm volatile("" : "+v"(e0[u]), "+v"(e1[u]));
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

    // ================= stage 0: gray tiles (tile + 8-px rim), 8 bytes per lane; all loads first, then the stores.
    // The first batch's keypoint records ride along.
    {
        uint2 c[kGrayIters], p[kGrayIters];
#pragma unroll
        for (int u = 0; u < kGrayIters; ++u) {
            const int t = tid + u * kTileThreads;
            const int r = t / (kTileCW / 8), q = t - r * (kTileCW / 8);
            const int gy = cy0 + r, gx = cx0 + 8 * q;
            const bool row_ok = t < kGrayTasks && gy >= 0 && gy < H;
            const int64_t ro = (int64_t)gy * a.f.row_stride;
            c[u] = load_px8(cur + ro, gx, W, row_ok, fast8);
            p[u] = load_px8(prev + ro, gx, W, row_ok, fast8);
        }
        make_records(0, min(kBatch, n_tile_kp));
#pragma unroll
        for (int u = 0; u < kGrayIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < kGrayTasks) {
                reinterpret_cast<uint2 *>(s_cur)[t] = c[u];
                reinterpret_cast<uint2 *>(s_prev)[t] = p[u];
            }
        }
    }
    __syncthreads();  TILE_STAMP(0);

    // ================= stage 1: MIP
    {
        uint8_t *p19 = lds + kOffP19;
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
                        uint8_t *dst = p19 + (h ? kk2 : kk) * (2 * kP19Pad);
#pragma unroll
                        for (int u = 0; u < kMipIters; ++u) {
                            const int c0x = (int)(cxp[u] & 0xffffu), c1x = (int)(cxp[u] >> 16);
                            const int t0 = __mul24((int)lds[a0[u] + mm.roi_off], c0x) + __mul24((int)lds[e0[u] + mm.roi_off], c1x);
                            const int t1 = __mul24((int)lds[a1[u] + mm.roi_off], c0x) + __mul24((int)lds[e1[u] + mm.roi_off], c1x);
                            const int px = resize_y(t0, t1, c0ys[u], c1ys[u]);
                            if (lane + 64 * u < a.mip_n) dst[pos[u]] = (uint8_t)px;
                        }
                    }
                } else {
                    int s00[2][kMipIters], s01[2][kMipIters], s10[2][kMipIters], s11[2][kMipIters];
#pragma unroll
                    for (int u = 0; u < kMipIters; ++u) {
                        s00[0][u] = lds[a0[u] + m.roi_off];
                        s01[0][u] = lds[e0[u] + m.roi_off];
                        s10[0][u] = lds[a1[u] + m.roi_off];
                        s11[0][u] = lds[e1[u] + m.roi_off];
                        s00[1][u] = lds[a0[u] + m2.roi_off];
                        s01[1][u] = lds[e0[u] + m2.roi_off];
                        s10[1][u] = lds[a1[u] + m2.roi_off];
                        s11[1][u] = lds[e1[u] + m2.roi_off];
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint8_t *dst = p19 + (h ? kk2 : kk) * (2 * kP19Pad);
#pragma unroll
                        for (int u = 0; u < kMipIters; ++u) {
                            // every factor fits 24 bits: full-rate v_mul_i32_i24 / v_mad_i32_i24
                            const int c0x = (int)(cxp[u] & 0xffffu), c1x = (int)(cxp[u] >> 16);
                            const int t0 = __mul24(s00[h][u], c0x) + __mul24(s01[h][u], c1x);
                            const int t1 = __mul24(s10[h][u], c0x) + __mul24(s11[h][u], c1x);
                            const int px = resize_y(t0, t1, c0ys[u], c1ys[u]);
                            if ((h == 0 || two) && lane + 64 * u < a.mip_n) dst[pos[u]] = (uint8_t)px;
                        }
                    }
                }
                wave_lds_sync();
                // the two 9-byte strips sit at arbitrary byte offsets: fetch the covering aligned dwords (a byte-wise
                // formulation lets the compiler fuse the loads into misaligned ds_read_b64s, 64 cycles each) and
                // shift the strips out; SSD = sum c^2 + sum p^2 - 2 sum c*p with packed u8 dot products
                uint32_t cd[2][3], pd[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t *b32 = reinterpret_cast<const uint32_t *>(p19 + ((h && two) ? kk2 : kk) * (2 * kP19Pad));
#pragma unroll
                    for (int w3 = 0; w3 < 3; ++w3) {
                        cd[h][w3] = b32[cw + w3];
                        pd[h][w3] = b32[pw + w3];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t c0 = __builtin_amdgcn_alignbyte(cd[h][1], cd[h][0], cs), c1 = __builtin_amdgcn_alignbyte(cd[h][2], cd[h][1], cs);
                    const uint32_t c2 = (cd[h][2] >> (8 * cs)) & 0xffu;
                    const uint32_t p0 = __builtin_amdgcn_alignbyte(pd[h][1], pd[h][0], ps), p1 = __builtin_amdgcn_alignbyte(pd[h][2], pd[h][1], ps);
                    const uint32_t p2 = (pd[h][2] >> (8 * ps)) & 0xffu;
                    const uint32_t sq = __builtin_amdgcn_udot4(c0, c0, __builtin_amdgcn_udot4(c1, c1, c2 * c2, false), false) +
                                        __builtin_amdgcn_udot4(p0, p0, __builtin_amdgcn_udot4(p1, p1, p2 * p2, false), false);
                    const uint32_t cross = __builtin_amdgcn_udot4(c0, p0, __builtin_amdgcn_udot4(c1, p1, c2 * p2, false), false);
                    const int ssd = (int)(sq - 2u * cross);
                    const uint64_t mot = __ballot(ssd > st.mip_theta);
                    if (lane == 0 && (h == 0 || two)) s_mot[h ? kk2 : kk] = make_uint2((uint32_t)mot, (uint32_t)(mot >> 32));
                }
            }
            __syncthreads();  TILE_STAMP(5);
            // a crowded tile (several batches) sends its motion bytes out now; the usual single batch keeps them in
            // LDS for one 16-byte store per descriptor at the end of stage 3
            if (!one_batch && tid < nb) *reinterpret_cast<uint2 *>(a.out_desc + (out_base + km[tid].g) * 16 + 8) = s_mot[tid];
        }
    }

    // ================= stage 2: integral of |cur - prev| over tile + halo, in LDS
    // Two blocked scans, each keeping its values in registers across the carry exchange, so that every pixel costs
    // one LDS write (row pass) plus one read and one write (column pass).
    {
        int32_t *carry = reinterpret_cast<int32_t *>(scratch);  // run totals, then block totals; then their prefixes
        // 2a: row pass.  A thread owns 16-pixel runs: |cur - prev| and the running sum inside the run (v_sad_u8 on
        //     masked dwords), run total -> LDS.
        int4 rv[kRunIters][4];
        {
            uint4 c[kRunIters], p[kRunIters];
#pragma unroll
            for (int u = 0; u < kRunIters; ++u) {
                const int t = tid + u * kTileThreads;
                const int r = run_row(t), q = run_col(t);
                const int gy = oy + r, gx = ox + 16 * q;
                const bool row_ok = t < kTileRH * kRunsPerRow && gy >= 0 && gy < H;
                const int64_t ro = (int64_t)gy * a.f.row_stride;
                c[u] = load_px16(cur + ro, gx, W, row_ok, fast16);
                p[u] = load_px16(prev + ro, gx, W, row_ok, fast16);
            }
#pragma unroll
            for (int u = 0; u < kRunIters; ++u) {
                const uint32_t cw4[4] = {c[u].x, c[u].y, c[u].z, c[u].w}, pw4[4] = {p[u].x, p[u].y, p[u].z, p[u].w};
                uint32_t acc = 0;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const uint32_t x = cw4[w4], y = pw4[w4];
                    rv[u][w4].x = (int)__builtin_amdgcn_sad_u8(x & 0xffu, y & 0xffu, acc);
                    rv[u][w4].y = (int)__builtin_amdgcn_sad_u8(x & 0xffffu, y & 0xffffu, acc);
                    rv[u][w4].z = (int)__builtin_amdgcn_sad_u8(x & 0xffffffu, y & 0xffffffu, acc);
                    acc = __builtin_amdgcn_sad_u8(x, y, acc);
                    rv[u][w4].w = (int)acc;
                }
                const int t = tid + u * kTileThreads;
                if (t < kTileRH * kRunsPerRow) carry[run_row(t) * kRunsPerRow + run_col(t)] = (int)acc;
            }
        }
        for (int i = tid; i < kIP; i += kTileThreads) I[i] = 0;                                    // integral row 0
        for (int r = tid; r <= kTileRH; r += kTileThreads) I[r * kIP + kIntegralColOffset] = 0;     // logical column 0
        __syncthreads();  TILE_STAMP(6);
        // 2b: per row, exclusive prefix of its 12 run totals (in place)
        if (tid < kTileRH) {
            int tot[kRunsPerRow];
#pragma unroll
            for (int q = 0; q < kRunsPerRow; ++q) tot[q] = carry[tid * kRunsPerRow + q];
            int run = 0;
#pragma unroll
            for (int q = 0; q < kRunsPerRow; ++q) {
                carry[tid * kRunsPerRow + q] = run;
                run += tot[q];
            }
        }
        __syncthreads();  TILE_STAMP(7);
        // 2c: finish the row pass: add the run's carry, one 16-byte LDS store per 4 pixels
#pragma unroll
        for (int u = 0; u < kRunIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < kTileRH * kRunsPerRow) {
                const int r = run_row(t), q = run_col(t);
                const int add = carry[r * kRunsPerRow + q];
                int4 *dst = reinterpret_cast<int4 *>(I + (r + 1) * kIP + 4 + 16 * q);
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4)
                    dst[w4] = make_int4(rv[u][w4].x + add, rv[u][w4].y + add, rv[u][w4].z + add, rv[u][w4].w + add);
            }
        }
        __syncthreads();  TILE_STAMP(8);
        // 2d: column pass.  A thread owns 20-row column segments: running sum in registers, segment total -> LDS
        constexpr int kColTasks = kColBlocks * kTileRW;
        constexpr int kColIters = (kColTasks + kTileThreads - 1) / kTileThreads;
        int cv[kColIters][kColBlockRows];
#pragma unroll
        for (int u = 0; u < kColIters; ++u) {
            const int t = min(tid + u * kTileThreads, kColTasks - 1);
            const int j = t / kTileRW, c = t - j * kTileRW;  // logical column c + 1
            const int32_t *e = I + (j * kColBlockRows + 1) * kIP + kIntegralColOffset + 1 + c;
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) cv[u][r] = e[r * kIP];
            int acc = 0;
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) {
                acc += cv[u][r];
                cv[u][r] = acc;
            }
            if (tid + u * kTileThreads < kColTasks) carry[t] = acc;
        }
        __syncthreads();  TILE_STAMP(9);
        // 2e: per column, exclusive prefix of its 8 segment totals (in place)
        if (tid < kTileRW) {
            int tot[kColBlocks];
#pragma unroll
            for (int j = 0; j < kColBlocks; ++j) tot[j] = carry[j * kTileRW + tid];
            int run = 0;
#pragma unroll
            for (int j = 0; j < kColBlocks; ++j) {
                carry[j * kTileRW + tid] = run;
                run += tot[j];
            }
        }
        __syncthreads();  TILE_STAMP(10);
        // 2f: finish the column pass
#pragma unroll
        for (int u = 0; u < kColIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < kColTasks) {
                const int j = t / kTileRW, c = t - j * kTileRW;
                const int add = carry[t];
                int32_t *e = I + (j * kColBlockRows + 1) * kIP + kIntegralColOffset + 1 + c;
#pragma unroll
                for (int r = 0; r < kColBlockRows; ++r) e[r * kIP] = cv[u][r] + add;
            }
        }
        __syncthreads();  TILE_STAMP(11);
    }

    // ================= stage 3: FREAK on the difference image
    {
        uint8_t *vv = scratch;                                             // [kBatch][kVStride] box means
        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (!one_batch) {  // crowded tile: the records of this batch (a single batch still has them from stage 0;
                __syncthreads();  // km is rewritten too, harmlessly: the scratch area is free between batches)
                make_records(b0, nb);
                __syncthreads();
            }
            const int n_box = nb * kNbPoints;
            if (st.orientation_normalized) {
                // F1: un-rotated box means; pattern points fetched for all of a thread's tasks before any is used
                {
                    PatternPoint P[kBoxIters];
                    float kx[kBoxIters], ky[kBoxIters];
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        const int t = min(tid + u * kTileThreads, n_box - 1);
                        const int kk = t / kNbPoints, p = t - kk * kNbPoints;
                        const KpFreak k = kf[kk];
                        kx[u] = k.kx;
                        ky[u] = k.ky;
                        P[u] = a.lut[(int64_t)k.idx * kNbOrientation * kNbPoints + p];
                    }
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        const int t = tid + u * kTileThreads;
                        if (t < n_box) {
                            const int kk = t / kNbPoints, p = t - kk * kNbPoints;
                            vv[kk * kVStride + p] = (uint8_t)mean_intensity_tile(I, ox, oy, kx[u], ky[u], P[u]);
                        }
                    }
                }
                __syncthreads();  TILE_STAMP(13);
                // F2: 8 lanes per keypoint share the 45 orientation pairs; theta
                for (int t = tid; t < nb * kOrientLanes; t += kTileThreads) {
                    const int kk = t / kOrientLanes, sub = t % kOrientLanes;
                    const uint8_t *v = vv + kk * kVStride;
                    int direction0 = 0, direction1 = 0;
#pragma unroll
                    for (int m0 = 0; m0 < kNbOrientPairs; m0 += kOrientLanes) {
                        const int m = m0 + sub;
                        if (m < kNbOrientPairs) {
                            const OrientPair op = st.orient[m];
                            const int delta = (int)v[op.i] - (int)v[op.j];
                            direction0 += __mul24(delta, op.weight_dx) / 2048;  // C division: truncates toward zero, per term
                            direction1 += __mul24(delta, op.weight_dy) / 2048;
                        }
                    }
#pragma unroll
                    for (int o = 1; o < kOrientLanes; o <<= 1) {
                        direction0 += __shfl_xor(direction0, o);
                        direction1 += __shfl_xor(direction1, o);
                    }
                    const int theta = theta_index(s_theta, direction0, direction1);
                    if (sub == 0) {
                        kf[kk].theta = (int16_t)theta;
                        if (a.out_info)
                            *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kk].g) * 4) = make_int4(kf[kk].idx, theta, direction0, direction1);
                    }
                }
                __syncthreads();  TILE_STAMP(14);
            } else if (a.out_info && tid < nb) {
                *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[tid].g) * 4) = make_int4(kf[tid].idx, 0, 0, 0);
            }
            // F3: box means of the rotated pattern
            {
                PatternPoint P[kBoxIters];
                float kx[kBoxIters], ky[kBoxIters];
#pragma unroll
                for (int u = 0; u < kBoxIters; ++u) {
                    const int t = min(tid + u * kTileThreads, n_box - 1);
                    const int kk = t / kNbPoints, p = t - kk * kNbPoints;
                    const KpFreak k = kf[kk];
                    kx[u] = k.kx;
                    ky[u] = k.ky;
                    P[u] = a.lut[((int64_t)k.idx * kNbOrientation + k.theta) * kNbPoints + p];
                }
#pragma unroll
                for (int u = 0; u < kBoxIters; ++u) {
                    const int t = tid + u * kTileThreads;
                    if (t < n_box) {
                        const int kk = t / kNbPoints, p = t - kk * kNbPoints;
                        vv[kk * kVStride + p] = (uint8_t)mean_intensity_tile(I, ox, oy, kx[u], ky[u], P[u]);
                    }
                }
            }
            __syncthreads();  TILE_STAMP(15);
            // F4: lane = descriptor bit
            {
                const int pi = st.bit_pair_i[lane], pj = st.bit_pair_j[lane];
                constexpr int kBitIters = (kBatch + kTileWaves - 1) / kTileWaves;
                int va[kBitIters], vb[kBitIters];
#pragma unroll
                for (int u = 0; u < kBitIters; ++u) {  // clamped, unguarded: the byte pairs of all the wave's keypoints in flight
                    const uint8_t *v = vv + min(wave + u * kTileWaves, nb - 1) * kVStride;
                    va[u] = v[pi];
                    vb[u] = v[pj];
                }
#pragma unroll
                for (int u = 0; u < kBitIters; ++u) {
                    const int kk = wave + u * kTileWaves;
                    bool bit;
                    if (st.bit_mode == MOFREAK_BITS_SSE)
                        bit = va[u] >= vb[u];
                    else if (st.bit_mode == MOFREAK_BITS_NATURAL)
                        bit = va[u] > vb[u];
                    else
                        bit = (int)(int8_t)va[u] > (int)(int8_t)vb[u];
                    const uint64_t app = __ballot(bit);
                    if (lane == 0 && kk < nb) s_bits[kk] = make_uint2((uint32_t)app, (uint32_t)(app >> 32));
                }
            }
            __syncthreads();  TILE_STAMP(16);
            if (tid < nb) {  // descriptor and validity flag out, side by side
                const int64_t out_idx = out_base + kf[tid].g;
                const uint2 app = s_bits[tid];
                if (one_batch) {
                    const uint2 mot = s_mot[tid];
                    *reinterpret_cast<uint4 *>(a.out_desc + out_idx * 16) = make_uint4(app.x, app.y, mot.x, mot.y);
                } else {
                    *reinterpret_cast<uint2 *>(a.out_desc + out_idx * 16) = app;
                }
                a.out_valid[out_idx] = 1;
            }
        }
    }
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
