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
constexpr int kTileLdsBytes = kOffSmall + (int)((sizeof(SmallTables) + 15) / 16 * 16);
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
static_assert(kBatch * kVStride + kBatch * (int)sizeof(KpFreak) <= kScratchBytes, "stage-3 arrays fit the scratch area");
static_assert(kBatch * (int)sizeof(KpMip) + 16 <= kScratchBytes, "stage-1 arrays fit the scratch area");

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

// FREAK::meanIntensity (box branch) on the tile-local integral; (ox, oy) = image coordinates of the region origin.
__device__ __forceinline__ int mean_intensity_tile(const int32_t *__restrict__ I, int ox, int oy, float kx, float ky,
                                                   const PatternPoint P)
{
    const float xf = P.x + kx;
    const float yf = P.y + ky;
    const float radius = P.sigma;
    const int x_left = (int)((double)(xf - radius) + 0.5) - ox;
    const int y_top = (int)((double)(yf - radius) + 0.5) - oy;
    const int x_right = (int)((double)(xf + radius) + 1.5) - ox;
    const int y_bottom = (int)((double)(yf + radius) + 1.5) - oy;
    const int32_t *top = I + y_top * kIP + kIntegralColOffset;
    const int32_t *bot = I + y_bottom * kIP + kIntegralColOffset;
    int ret_val = bot[x_right];
    ret_val -= bot[x_left];
    ret_val += top[x_left];
    ret_val -= top[x_right];
    return div_box(ret_val, (x_right - x_left) * (y_bottom - y_top)) & 0xff;
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
        } else {
            key = -2;
        }
    }
    a.kp_key[g] = key;
}

// Pass 2: exclusive scan of the tile populations (single workgroup; n_keys is a few hundred to ~1e5).
__global__ __launch_bounds__(256) void bin_scan_kernel(int32_t *tile_start, int64_t n_keys)
{
    __shared__ int carry_s;
    __shared__ int wave_tot[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n_keys; b0 += 256) {
        const int64_t b = b0 + threadIdx.x;
        const int v = b < n_keys ? tile_start[b] : 0;
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
    if (key >= 0) {
        const int pos = a.tile_start[key] + atomicAdd(&a.tile_cursor[key], 1);
        const mofreak_keypoint kp = a.kps[g];
        SortedKp s;
        s.x = kp.x;
        s.y = kp.y;
        s.size = kp.size;
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
__global__ __launch_bounds__(kTileThreads) void tile_kernel(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int n_tiles = a.tiles_x * a.tiles_y;
    const int tile = blockIdx.x, pair = blockIdx.y;
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

    // ================= stage 0: gray tiles (tile + 8-px rim), 8 bytes per lane; all loads first, then the stores
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
#pragma unroll
        for (int u = 0; u < kGrayIters; ++u) {
            const int t = tid + u * kTileThreads;
            if (t < kGrayTasks) {
                reinterpret_cast<uint2 *>(s_cur)[t] = c[u];
                reinterpret_cast<uint2 *>(s_prev)[t] = p[u];
            }
        }
    }
    __syncthreads();

    // ================= stage 1: MIP
    {
        uint8_t *p19 = lds + kOffP19;
        KpMip *km = reinterpret_cast<KpMip *>(scratch);
        int *s_flags = reinterpret_cast<int *>(scratch + kBatch * sizeof(KpMip));  // [0] = first L, [1] = mixed sides
        // per-lane constants of the sampling passes: where each sampled pixel goes, which frame it comes from
        int pos[kMipIters];
#pragma unroll
        for (int u = 0; u < kMipIters; ++u) pos[u] = a.mip_pos[min(lane + 64 * u, a.mip_stride - 1)];
        // per-lane constants of the bit pass: lane = 8*centre + offset (MoFREAKUtilities.cpp:56-70, 308-316)
        const int mc = lane >> 3, mi = lane & 7;
        const int mcx = (0xDDD99555u >> (4 * mc)) & 15, mcy = (0xD95D5D95u >> (4 * mc)) & 15;
        const int mdx = (int)((0x14787410u >> (4 * mi)) & 15) - 4, mdy = (int)((0x10147874u >> (4 * mi)) & 15) - 4;
        const int base_c = (mcy - 1) * kPatch + (mcx - 1);
        const int base_p = kP19Pad + (mcy + mdy - 1) * kPatch + (mcx + mdx - 1);
        MipSample sm[kMipIters];
        int have_L = -1;
        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (tid < 2) s_flags[tid] = 0;
            __syncthreads();
            if (tid < nb) {
                const SortedKp kp = tile_kps[b0 + tid];
                const int x_i = (int)kp.x, y_i = (int)kp.y;                    // :460 float -> int parameters
                const int half = ((int)kp.size) / 2, L = (int)ceilf(kp.size);  // :293-295
                KpMip m;
                m.g = kp.g;
                m.roi_off = (uint16_t)((y_i - half - cy0) * kTileCW + (x_i - half - cx0));
                m.L = (uint8_t)L;
                m.pad = 0;
                km[tid] = m;
                if (tid == 0) s_flags[0] = L;
            }
            __syncthreads();
            if (tid < nb && km[tid].L != s_flags[0]) s_flags[1] = 1;
            __syncthreads();
            const bool uniform = s_flags[1] == 0;
            if (uniform && s_flags[0] != have_L) {  // one ROI side in the batch (the usual case): samples stay in registers
                have_L = s_flags[0];
                const MipSample *tab = a.mip_samples + (int64_t)have_L * a.mip_stride;
#pragma unroll
                for (int u = 0; u < kMipIters; ++u) sm[u] = tab[min(lane + 64 * u, a.mip_stride - 1)];
            }
            // the pixels of the two 19x19 resamples that the MIP reads
            for (int kk = wave; kk < nb; kk += kTileWaves) {
                const KpMip m = km[kk];
                if (!uniform && m.L != have_L) {
                    have_L = m.L;
                    const MipSample *tab = a.mip_samples + (int64_t)have_L * a.mip_stride;
#pragma unroll
                    for (int u = 0; u < kMipIters; ++u) sm[u] = tab[min(lane + 64 * u, a.mip_stride - 1)];
                }
                uint8_t *dst = p19 + kk * (2 * kP19Pad);
                int px[kMipIters];
#pragma unroll
                for (int u = 0; u < kMipIters; ++u) {
                    const uint8_t *src = ((lane + 64 * u) < a.mip_n_cur ? s_cur : s_prev) + m.roi_off;
                    const int t0 = (int)src[sm[u].off00] * sm[u].c0x + (int)src[sm[u].off01] * sm[u].c1x;
                    const int t1 = (int)src[sm[u].off10] * sm[u].c0x + (int)src[sm[u].off11] * sm[u].c1x;
                    px[u] = ((((int)sm[u].c0y * (t0 >> 4)) >> 16) + (((int)sm[u].c1y * (t1 >> 4)) >> 16) + 2) >> 2;
                }
#pragma unroll
                for (int u = 0; u < kMipIters; ++u)
                    if (lane + 64 * u < a.mip_n) dst[pos[u]] = (uint8_t)px[u];
            }
            __syncthreads();
            for (int kk = wave; kk < nb; kk += kTileWaves) {
                const uint8_t *b = p19 + kk * (2 * kP19Pad);
                int ssd = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int d = (int)b[base_c + k] - (int)b[base_p + k];
                    ssd += d * d;
                }
                const uint64_t mot = __ballot(ssd > st.mip_theta);
                if (lane == 0)
                    *reinterpret_cast<uint2 *>(a.out_desc + (out_base + km[kk].g) * 16 + 8) = make_uint2((uint32_t)mot, (uint32_t)(mot >> 32));
            }
            __syncthreads();
        }
    }

    // ================= stage 2: integral of |cur - prev| over tile + halo, in LDS
    {
        // 2a: per 16-pixel run: absolute differences and their running sum inside the run
        {
            uint4 c[kRunIters], p[kRunIters];
#pragma unroll
            for (int u = 0; u < kRunIters; ++u) {
                const int t = tid + u * kTileThreads;
                const int r = t / kRunsPerRow, q = t - r * kRunsPerRow;
                const int gy = oy + r, gx = ox + 16 * q;
                const bool row_ok = t < kTileRH * kRunsPerRow && gy >= 0 && gy < H;
                const int64_t ro = (int64_t)gy * a.f.row_stride;
                c[u] = load_px16(cur + ro, gx, W, row_ok, fast16);
                p[u] = load_px16(prev + ro, gx, W, row_ok, fast16);
            }
#pragma unroll
            for (int u = 0; u < kRunIters; ++u) {
                const int t = tid + u * kTileThreads;
                if (t < kTileRH * kRunsPerRow) {
                    const int r = t / kRunsPerRow, q = t - r * kRunsPerRow;
                    const uint32_t cw[4] = {c[u].x, c[u].y, c[u].z, c[u].w}, pw[4] = {p[u].x, p[u].y, p[u].z, p[u].w};
                    int4 *dst = reinterpret_cast<int4 *>(I + (r + 1) * kIP + 4 + 16 * q);
                    int acc = 0;
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        int4 o;
                        acc += absdiff_u8(cw[w4], pw[w4], 0);
                        o.x = acc;
                        acc += absdiff_u8(cw[w4], pw[w4], 1);
                        o.y = acc;
                        acc += absdiff_u8(cw[w4], pw[w4], 2);
                        o.z = acc;
                        acc += absdiff_u8(cw[w4], pw[w4], 3);
                        o.w = acc;
                        dst[w4] = o;
                    }
                }
            }
        }
        for (int i = tid; i < kIP; i += kTileThreads) I[i] = 0;                                    // integral row 0
        for (int r = tid; r <= kTileRH; r += kTileThreads) I[r * kIP + kIntegralColOffset] = 0;     // logical column 0
        __syncthreads();
        // 2b: per row, exclusive prefix of the run totals
        int32_t *rowcarry = reinterpret_cast<int32_t *>(scratch);
        if (tid < kTileRH) {
            int tot[kRunsPerRow];
#pragma unroll
            for (int q = 0; q < kRunsPerRow; ++q) tot[q] = I[(tid + 1) * kIP + kIntegralColOffset + 16 * q + 16];
            int run = 0;
#pragma unroll
            for (int q = 0; q < kRunsPerRow; ++q) {
                rowcarry[tid * kRunsPerRow + q] = run;
                run += tot[q];
            }
        }
        __syncthreads();
        // 2c: column pass in blocks of 20 rows, adding the row carries on the way
        for (int t = tid; t < kColBlocks * kTileRW; t += kTileThreads) {
            const int j = t / kTileRW, c = t - j * kTileRW;  // logical column c + 1
            int32_t *e = I + (j * kColBlockRows + 1) * kIP + kIntegralColOffset + 1 + c;
            const int32_t *rc = rowcarry + j * kColBlockRows * kRunsPerRow + (c >> 4);
            int v[kColBlockRows];
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) v[r] = e[r * kIP] + rc[r * kRunsPerRow];
            int acc = 0;
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) {
                acc += v[r];
                e[r * kIP] = acc;
            }
        }
        __syncthreads();
        // 2d: per column, exclusive prefix of the block totals
        int32_t *colcarry = reinterpret_cast<int32_t *>(scratch);
        if (tid < kTileRW) {
            int tot[kColBlocks];
#pragma unroll
            for (int j = 0; j < kColBlocks; ++j) tot[j] = I[(j + 1) * kColBlockRows * kIP + kIntegralColOffset + 1 + tid];
            int run = 0;
#pragma unroll
            for (int j = 0; j < kColBlocks; ++j) {
                colcarry[j * kTileRW + tid] = run;
                run += tot[j];
            }
        }
        __syncthreads();
        // 2e: add the block carries
        for (int t = tid; t < (kColBlocks - 1) * kTileRW; t += kTileThreads) {
            const int j = 1 + t / kTileRW, c = t % kTileRW;
            const int add = colcarry[j * kTileRW + c];
            int32_t *e = I + (j * kColBlockRows + 1) * kIP + kIntegralColOffset + 1 + c;
            int v[kColBlockRows];
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) v[r] = e[r * kIP];
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r) e[r * kIP] = v[r] + add;
        }
        __syncthreads();
    }

    // ================= stage 3: FREAK on the difference image
    {
        uint8_t *vv = scratch;                                             // [kBatch][kVStride] box means
        KpFreak *kf = reinterpret_cast<KpFreak *>(scratch + kBatch * kVStride);
        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (tid < nb) {
                const SortedKp kp = tile_kps[b0 + tid];
                KpFreak k;
                k.kx = kp.x;
                k.ky = kp.y;
                k.g = kp.g;
                k.idx = (int16_t)scale_index_scalar(&st, kp.size);
                k.theta = 0;
                kf[tid] = k;
            }
            __syncthreads();
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
                __syncthreads();
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
                            direction0 += delta * op.weight_dx / 2048;  // C division: truncates toward zero, per term
                            direction1 += delta * op.weight_dy / 2048;
                        }
                    }
#pragma unroll
                    for (int o = 1; o < kOrientLanes; o <<= 1) {
                        direction0 += __shfl_xor(direction0, o);
                        direction1 += __shfl_xor(direction1, o);
                    }
                    const int theta = theta_index(direction0, direction1);
                    if (sub == 0) {
                        kf[kk].theta = (int16_t)theta;
                        if (a.out_info)
                            *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kk].g) * 4) = make_int4(kf[kk].idx, theta, direction0, direction1);
                    }
                }
                __syncthreads();
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
            __syncthreads();
            // F4: lane = descriptor bit
            {
                const int pi = st.bit_pair_i[lane], pj = st.bit_pair_j[lane];
                for (int kk = wave; kk < nb; kk += kTileWaves) {
                    const uint8_t *v = vv + kk * kVStride;
                    const int va = v[pi], vb = v[pj];
                    bool bit;
                    if (st.bit_mode == MOFREAK_BITS_SSE)
                        bit = va >= vb;
                    else if (st.bit_mode == MOFREAK_BITS_NATURAL)
                        bit = va > vb;
                    else
                        bit = (int)(int8_t)va > (int)(int8_t)vb;
                    const uint64_t app = __ballot(bit);
                    if (lane == 0) {
                        const int64_t out_idx = out_base + kf[kk].g;
                        *reinterpret_cast<uint2 *>(a.out_desc + out_idx * 16) = make_uint2((uint32_t)app, (uint32_t)(app >> 32));
                        a.out_valid[out_idx] = 1;
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

int launch_bin(const BinArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(a.tile_start, 0, (size_t)(a.n_keys + 1) * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.tile_cursor, 0, (size_t)a.n_keys * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(a.slow_count, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    const int blocks = (int)((a.n_kp + 255) / 256);
    if (blocks > 0) hipLaunchKernelGGL(bin_count_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, a.tile_start, a.n_keys);
    if (blocks > 0) hipLaunchKernelGGL(bin_scatter_kernel, dim3(blocks), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int launch_tile(const TileArgs &a, void *stream)
{
    if (a.mip_n > 64 * kMipIters || a.mip_stride < a.mip_n) return (int)hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kTileLdsBytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tile_kernel, dim3(a.tiles_x * a.tiles_y, a.n_pairs), dim3(kTileThreads), kTileLdsBytes,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // namespace mofreak
