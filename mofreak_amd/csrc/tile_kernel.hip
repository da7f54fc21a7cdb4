// Fused tile kernel of the MoFREAK path for gfx950 (CDNA4): one 512-thread workgroup owns a 128x64-pixel tile of one
// frame pair and describes every keypoint whose pixel falls in it, entirely out of LDS.  Two workgroups share a CU
// (79.5 KB of LDS each), so one workgroup's barriers and global-memory latencies are covered by the other's work.
//
//   stage 0  the gray bytes of the tile + halo of `current` and `previous` -> LDS, each region row into the LDS row
//            that will hold its integral: the frames are fetched once, in one round trip to memory per tile
//   stage 1  MIP (MoFREAKUtilities.cpp:288-325, 46-99), one wave per keypoint, no workgroup barrier: the ~280 pixels of
//            the two 19x19 resamples that motionInterchangePattern actually reads (cv::resize fixed-point bilinear,
//            host-built sample table per ROI side held in registers) -> the wave's own LDS buffer; then
//            lane = 8*centre + offset, strip SSD, __ballot = the 8 motion bytes
//   stage 2  integral image of |current - previous| over tile + halo, kept MODULO 2^16 (u16, two pixels per dword),
//            built in place over the staged bytes: the row pass scans inside a wave (v_sad_u8 inside a lane's 16
//            pixels, DPP row_shr across the 16 lanes of a region row), the column pass adds packed pairs
//            (v_pk_add_u16).  A box sum is exact modulo 2^16 as long as the box holds at most 257 pixels
//            (257 * 255 < 2^16); larger boxes are summed in horizontal slices of at most 257 pixels each.  Box sums
//            are translation-invariant, so the tile-local integral gives the same box means as cv::integral of the
//            whole frame -- which never exists in HBM.
//            The halo is sized per call from the largest FREAK pattern among the call's tile-path keypoints
//            (binning pass, device-resident word): 24, 32 or 40 pixels.
//   stage 3  FREAK (cv::FREAK::compute on the difference image, :427-428), one wave per group of four keypoints, no
//            workgroup barrier: 43 box means per keypoint (172 tasks over three 64-lane passes), orientation with
//            16 lanes per keypoint (DPP row reduction), rotated means, lane = descriptor bit, __ballot = the 8
//            appearance bytes, one 16-byte store per descriptor
//
// HBM traffic is the two frames (halo re-reads are served by L2 / Infinity Cache) + keypoints in + descriptors out.
// Keypoints whose FREAK pattern does not fit the 40-px halo (patternSizes[scale] > 40, i.e. size >= ~12.56) or whose
// ROI side exceeds 16 are left to the gather path (describe_kernel over a global integral) by the binning pass.
#include <type_traits>

#include "device_helpers.h"
#include "mip_lane.h"

namespace mofreak {
namespace {

constexpr int kTileThreads = 512;                    // 8 waves; two workgroups per CU = 4 waves per SIMD (ten-wave workgroups at
                                                     // <= 96 registers were tried: the second one does not fit beside the first)
constexpr int kMipLaneMinL = 7, kMipLaneMaxL = 13;   // ROI sides the lane-per-keypoint MIP is compiled for (keypoint sizes 6 < s <= 13: FREAK's smallest
                                                     // pattern up to what the tile's halo admits); other sides take the wave-per-keypoint stage
constexpr int kMipWaves = 4;                         // waves that compute the MIP (a lane per keypoint, mip_lane.h) while the others stage the tile
constexpr int kStageThreads = kTileThreads - 64 * kMipWaves;
constexpr int kStageRows = kStageThreads / 16;       // region rows a staging step takes (16 lanes per row)
constexpr int kTileLdsLimit = 80 * 1024;             // a workgroup's LDS budget (two per CU); debug builds check accesses against it
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kBatch = 128;                          // keypoints described per pass over a tile's list
constexpr int kGroup = 4;                            // keypoints one wave describes together in stage 3
constexpr int kMinHalo = 24;                         // smallest integral halo (patternSizes[0] = 23)
constexpr int kIPitch = kTileStagePitch / 2;         // LDS integral pitch (u16); logical column c at physical c+7
constexpr int kIColOff = 7;
constexpr int kIPitchDw = kIPitch / 2;
constexpr int kIntegralBytes = (kTileRH + 1) * kTileStagePitch;
constexpr int kStageIters = (kTileRH + kStageRows - 1) / kStageRows;           // staging steps over the largest region
constexpr int kRowGroupIters = (kTileRH / 4 + kTileWaves - 1) / kTileWaves;   // 4-row groups per wave in the row pass
constexpr int kColBlockRows = 16;
constexpr int kMaxColBlocks = kTileRH / kColBlockRows;                         // 10
constexpr int kMaxQcols = kTileRW / 4;                                         // 48 columns of four pixels (8 bytes of u16)
constexpr int kVStride = 44;                         // bytes per keypoint in a wave's box-mean buffer (11 dwords: odd)
constexpr int kMipIters = 5;                         // 64-lane passes over the <= 320 sampled 19x19 positions
constexpr int kBoxIters = (kGroup * kNbPoints + 63) / 64;                       // 3
constexpr int kP19Wave = 2 * kP19Pad;                // a wave's MIP buffer: (cur19, prev19) of one keypoint

// ---- LDS carve (bytes); every offset is a multiple of 16
constexpr int kOffIntegral = 0;                      // row 0: zeros; row r + 1: region row r (staged bytes, then integral)
constexpr int kOffScratch = (kIntegralBytes + 15) / 16 * 16;
constexpr int kScratchBytes = kTileWaves * kP19Wave;           // stage 1: 19x19 buffers; stage 2: column-block totals; stage 3: box means
constexpr int kOffTheta = kOffScratch + kScratchBytes;
constexpr int kOffKf = kOffTheta + kThetaBounds * (int)sizeof(ThetaBound);   // the batch's keypoint records
constexpr int kOffMot = kOffKf + kBatch * 16;                                // motion bytes kept for the fused store
constexpr int kOffKint = kOffMot + kBatch * 8;                               // per keypoint: its corner in the integral (integer coordinates); where its ROI starts in
                                                                             // the staged rows (stage 1), then its block of the pattern tables (stage 3)
constexpr int kStampLds = 8;                                                 // diagnostic build only: the phases' u64 tick sums
constexpr int kOffStamps = kOffKint + kBatch * 8;
constexpr int kTileLdsBytes = kOffStamps + kStampLds * 8;
constexpr int kDirsInScratch = kTileWaves * kGroup * kVStride;               // stage 3: behind the waves' box means, every wave's keypoints' orientation sums
constexpr int kDirsPerWave = (kBatch / kTileWaves) * 8;
static_assert(kDirsInScratch % 8 == 0 && kDirsInScratch + kTileWaves * kDirsPerWave <= kScratchBytes, "the orientation sums fit the scratch area");
static_assert(kP19Wave % 16 == 0 && kOffScratch % 16 == 0 && kOffTheta % 16 == 0, "LDS carve alignment");
static_assert(2 * kTileLdsBytes <= 160 * 1024 && kTileLdsBytes <= kTileLdsLimit, "two workgroups per CU");
static_assert(kTileRW % 16 == 0 && kTileRH % kColBlockRows == 0 && kTileRW / 16 <= 16 && kTileThreads % 64 == 0, "region blocking");
static_assert(kTileH + 2 * kMinHalo >= kStageRows, "a last partial staging step can be shifted up to a full one");
static_assert(kTileStagePitch % 16 == 0 && 2 * kTileRW <= kTileStagePitch, "a region row's staged bytes fit its integral row");
static_assert(kMaxColBlocks * kMaxQcols * 8 <= kScratchBytes, "column-block totals fit the scratch area");
static_assert(kMaxColBlocks * kMaxQcols <= kTileThreads, "one column task per thread");
static_assert(kTileWaves * kGroup * kVStride <= kScratchBytes, "stage-3 box means fit the scratch area");
static_assert(kBatch % (kGroup * kTileWaves) == 0, "whole groups per wave in a full batch");
static_assert(kTileMipHalo + 1 <= kMinHalo, "a tile-path ROI stays inside the smallest staged region");

struct KpRec {   // per-keypoint record of a batch
    float kx, ky;
    int32_t g;
    uint16_t pk;    // FREAK scale index | ROI side << 6 | ROI half << 11 (MoFREAKUtilities.cpp:293-295)
    int16_t unused;
};
static_assert(sizeof(KpRec) == 16, "record size used by the LDS carve");

// LDS accesses by integer byte address (the address of the dynamic LDS block is folded into the scalar bases once):
// the per-lane address arithmetic stays 32-bit and the instruction's immediate offset takes the constant part.
//
// Debug build (-DMOFREAK_DEBUG_BOUNDS, libmofreak_hip_debug.so): every such access is checked against
// the workgroup's LDS allocation and every descriptor store against the output's extent; an access outside is not made,
// the fact is recorded and comes back from mofreak_check_status as an error.  The product build has none of it.
#ifdef MOFREAK_DEBUG_BOUNDS
__device__ unsigned int g_tile_oob;  // bit 0: LDS access outside the allocation, bit 1: descriptor store outside the output
#define MOFREAK_LDS_CHECK(addr, T)                                             \
    if ((uint64_t)(addr) + sizeof(T) > (uint64_t)kTileLdsLimit || ((addr) % alignof(T)) != 0) { \
        atomicOr(&g_tile_oob, 1u);                                             \
        (addr) = 0;                                                            \
    }
#else
#define MOFREAK_LDS_CHECK(addr, T)
#endif
template <class T>
__device__ __forceinline__ T lds_ld(uint32_t addr)
{
    MOFREAK_LDS_CHECK(addr, T)
    return *(const __attribute__((address_space(3))) T *)(uintptr_t)addr;
}
template <class T>
__device__ __forceinline__ void lds_st(uint32_t addr, T v)
{
    MOFREAK_LDS_CHECK(addr, T)
    *(__attribute__((address_space(3))) T *)(uintptr_t)addr = v;
}

typedef uint32_t LdsU4 __attribute__((ext_vector_type(4)));
typedef uint32_t LdsU2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(8))) Px16 {
    uint32_t w[4];
};

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
// P.rows_per_slice (host): how many rows of this point's box are certain to hold at most 257 pixels.
__device__ __forceinline__ int mean_intensity_tile(uint32_t ibase, float kx, float ky, const PatternPoint P)
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
    const uint32_t w2 = 2u * (uint32_t)w;
    const int rps = P.rows_per_slice;
    uint32_t addr = ibase + 2u * (__umul24(y_top, kIPitch) + (uint32_t)x_left);
    int prev = (int)lds_ld<uint16_t>(addr + w2) - (int)lds_ld<uint16_t>(addr);
    int step = min(rps, h);
    addr += __umul24(step, 2 * kIPitch);
    int cur = (int)lds_ld<uint16_t>(addr + w2) - (int)lds_ld<uint16_t>(addr);
    int sum = (cur - prev) & 0xffff;
    int left = h - step;
    while (left > 0) {  // outer rings of the larger patterns only
        prev = cur;
        step = min(rps, left);
        addr += __umul24(step, 2 * kIPitch);
        cur = (int)lds_ld<uint16_t>(addr + w2) - (int)lds_ld<uint16_t>(addr);
        sum += (cur - prev) & 0xffff;
        left -= step;
    }
    return div_box_small(sum, (int)__umul24(w, h)) & 0xff;
}

// The same box mean for a keypoint at integer coordinates whose box is a fixed offset from it (BoxInt, tables.h):
// `kaddr` = LDS byte address of the keypoint's own corner in the integral.
__device__ __forceinline__ int mean_intensity_int(uint32_t kaddr, const BoxInt b)
{
    const uint32_t a0 = kaddr + (uint32_t)(int)b.off_tl, a1 = a0 + b.step1;
    int prev = (int)lds_ld<uint16_t>(a0 + b.w2) - (int)lds_ld<uint16_t>(a0);
    int cur = (int)lds_ld<uint16_t>(a1 + b.w2) - (int)lds_ld<uint16_t>(a1);
    int sum = (cur - prev) & 0xffff;
    if (b.left) {  // outer rings of the larger patterns only
        uint32_t addr = a1;
        int left = b.left;
        do {
            prev = cur;
            const int step = min((int)b.rps, left);
            addr += __umul24(step, kTileStagePitch);
            cur = (int)lds_ld<uint16_t>(addr + b.w2) - (int)lds_ld<uint16_t>(addr);
            sum += (cur - prev) & 0xffff;
            left -= step;
        } while (left > 0);
    }
    return (int)__builtin_fmaf((float)sum, b.inv_area, 0.5f * b.inv_area) & 0xff;  // div_box_small with the host's reciprocal
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

// Pass 1: classify every keypoint (erased / tile path / gather path) and count tile populations.  The two searches a
// keypoint needs -- its FREAK scale among the 63 size thresholds, its pair among the CSR offsets -- run on copies in
// LDS: in global memory each is a chain of five or six dependent loads.
constexpr int kBinOffsetsLds = 2048;  // CSR offsets (pairs + 1) a workgroup keeps in LDS; more: searched in global memory

__global__ __launch_bounds__(256) void bin_count_kernel(BinArgs a)
{
    __shared__ float s_thr[kNbScales];
    __shared__ int32_t s_ps[kNbScales];
    __shared__ int64_t s_off[kBinOffsetsLds];
    const SmallTables *st = a.small;
    if (threadIdx.x < kNbScales) {
        s_thr[threadIdx.x] = st->scale_thresholds[threadIdx.x];
        s_ps[threadIdx.x] = st->pattern_sizes[threadIdx.x];
    }
    const bool offsets_in_lds = a.kp_offsets != nullptr && a.n_pairs + 1 <= kBinOffsetsLds;
    if (offsets_in_lds)
        for (int i = threadIdx.x; i <= a.n_pairs; i += 256) s_off[i] = a.kp_offsets[i];
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = g < a.n_kp;
    int key = -1;
    int tile_ps = 0, tile_L = 0;
    if (live) {
        const mofreak_keypoint kp = a.kps[g];
        const float kx = kp.x, ky = kp.y, size = kp.size;
        // DescriptorExtractor::compute + FREAK::computeImpl keypoint filter (same tests as describe_kernel)
        bool ok = (size >= FLT_EPSILON) && (size <= FLT_MAX) && (fabsf(kx) <= FLT_MAX) && (fabsf(ky) <= FLT_MAX);
        int idx = 0;
        if (ok) {
            if (!st->scale_normalized) {
                idx = st->fixed_scale_index;
            } else {
                int lo = 0, hi = kNbScales - 1;  // number of thresholds <= size (thresholds ascend)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (size >= s_thr[mid])
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                idx = lo;
            }
        }
        a.kp_scale[g] = (uint8_t)idx;  // pass 3 packs it into the sorted record
        const int ps = s_ps[idx];
        if (kx <= ps || ky <= ps || kx >= a.W - ps || ky >= a.H - ps) ok = false;
        if (ok) {
            const int x_i = (int)kx, y_i = (int)ky;
            const int half = ((int)size) / 2, L = (int)ceilf(size);
            const bool roi_in = (x_i - half >= 0) && (y_i - half >= 0) && (x_i - half + L <= a.W) && (y_i - half + L <= a.H);
            const bool fast = !a.force_slow && ps <= kTileHalo && L <= kTileMaxRoi && half <= kTileMipHalo &&
                              (L - half) <= kTileMipHalo + 1 && roi_in;
            const int pr = a.kp_offsets ? pair_of(offsets_in_lds ? s_off : a.kp_offsets, a.n_pairs, g) : 0;
            if (fast) {
                key = pr * (a.tiles_x * a.tiles_y) + (y_i / kTileH) * a.tiles_x + (x_i / kTileW);
                tile_ps = ps;
                tile_L = L;
            } else {  // gather path: binned too, by the band of kTileH rows of its pair (band key b as -3 - b)
                key = -3 - (pr * a.tiles_y + y_i / kTileH);
            }
        }
        a.kp_key[g] = key;
    }
    // Tile populations and ROI-side ranges: one set of atomics per distinct tile in the wave, not per keypoint (neighbouring
    // keypoints share tiles, and atomics on one address from all over the chip are served one after the other)
    for (unsigned long long todo = __ballot(key >= 0); todo;) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader), lead_L = __shfl(tile_L, leader);
        const unsigned long long same = __ballot(key == k);
        if (lane_id() == leader) {
            atomicAdd(&a.tile_start[k], __popcll(same));
            atomicMax(&a.tile_lmin_c[k], ~(uint32_t)lead_L);
            atomicMax(&a.tile_lmax[k], (uint32_t)lead_L);
        } else if (key == k && tile_L != lead_L) {  // mixed sizes in one tile: rare
            atomicMax(&a.tile_lmin_c[k], ~(uint32_t)tile_L);
            atomicMax(&a.tile_lmax[k], (uint32_t)tile_L);
        }
        todo &= ~same;
    }
    for (unsigned long long todo = __ballot(key <= -3); todo;) {  // the gather path's bands: keys behind the tiles'
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long same = __ballot(key == k);
        if (lane_id() == leader) atomicAdd(&a.tile_start[a.n_keys + (-3 - k)], __popcll(same));
        todo &= ~same;
    }
    // How many keypoints already need the gather path (pass 2 decides from it whether thin tiles follow them there) and
    // the largest pattern on the tile path (it sizes the tile kernel's integral halo): one pair of numbers per
    // workgroup, reduced by pass 2 -- thousands of atomics on one address are served one after the other, and a wave
    // does not retire before its own has been.
    __shared__ int s_wave_slow[4], s_wave_ps[4];
    const unsigned long long slow = __ballot(key <= -3);
    int m = tile_ps;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (lane_id() == 0) {
        s_wave_slow[threadIdx.x >> 6] = __popcll(slow);
        s_wave_ps[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.wg_slow[blockIdx.x] = s_wave_slow[0] + s_wave_slow[1] + s_wave_slow[2] + s_wave_slow[3];
        a.wg_maxps[blockIdx.x] = max(max(s_wave_ps[0], s_wave_ps[1]), max(s_wave_ps[2], s_wave_ps[3]));
    }
}

// A tile costs the same whether it holds 5 keypoints or 90 (gray tiles, the integral), the gather path costs per
// keypoint.  When the gather path runs anyway for a good share of the call (large keypoints: a detector's output),
// thinly populated tiles are cheaper there; on dense grids (86 keypoints per tile on the benchmark) nothing changes.
// With the gather path's keypoints in bands (pass 3) a keypoint more in a band it works on anyway costs it ~1.1 ns, a
// tile ~18 ns before its first keypoint: measured on detector output (32 full-HD pairs, bin + tile + gather per call)
// 0.618 ms at 8, 0.594 at 12, 0.586 at 16, 0.578 at 24, 0.576 at 32, 0.560 at 48 (profiles/r03_gather_experiments.txt).
constexpr int kSparseTile = 48;  // keypoints below which a tile is handed to the gather path
constexpr int kSparseMarker = -(1 << 30);

// Pass 2: exclusive scan of the populations -- the tiles', then, continuing behind them, those of the gather path's bands
// (single workgroup; n_keys is a few hundred to ~1e5): every thread sums a run of consecutive keys, the run totals are
// scanned across the workgroup, every thread writes its run's starts -- two sweeps of independent loads instead of a
// barrier-separated step per 256 keys.  The keypoints of a tile that is handed to the gather path are added to their
// band's population first (a tile row IS a band).
constexpr int kScanThreads = 1024, kScanWaves = kScanThreads / 64;

__global__ __launch_bounds__(kScanThreads) void bin_scan_kernel(int32_t *tile_start, int32_t *tile_cursor, int32_t *slow_count, int32_t *max_ps,
                                                       const int32_t *wg_slow, const int32_t *wg_maxps, int n_blocks, int64_t n_kp, int64_t n_keys,
                                                       int64_t n_bkeys, int tiles_x, int tiles_y)
{
    __shared__ int wave_tot[kScanWaves], wave_slow[kScanWaves], wave_ps[kScanWaves];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // pass 1's per-workgroup figures: keypoints that need the gather path anyway, the largest pattern on the tile path
    int ns = 0, mp = 0;
    for (int i = threadIdx.x; i < n_blocks; i += kScanThreads) {
        ns += wg_slow[i];
        mp = max(mp, wg_maxps[i]);
    }
    ns = wave_sum(ns);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mp = max(mp, __shfl_xor(mp, o));
    if (lane == 0) {
        wave_slow[w] = ns;
        wave_ps[w] = mp;
    }
    __syncthreads();
    int n_slow = 0, max_pattern = 0;
    for (int i = 0; i < kScanWaves; ++i) {
        n_slow += wave_slow[i];
        max_pattern = max(max_pattern, wave_ps[i]);
    }
    const bool drop_sparse = n_slow > 0 && (int64_t)n_slow * 8 >= n_kp;
    if (threadIdx.x == 0) *max_ps = max_pattern;
    const int tiles = tiles_x * tiles_y;
    constexpr int kPre = 8;  // keys whose populations are requested together
    // one scan of `n` populations at `pop`, the starts continuing from `first`; returns the total behind them
    auto scan = [&](int32_t *pop, int64_t n, int first, bool tile_keys) -> int {
        const int64_t run = (n + kScanThreads - 1) / kScanThreads, b0 = min((int64_t)threadIdx.x * run, n), b1 = min(b0 + run, n);
        int sum = 0;
        for (int64_t bb = b0; bb < b1; bb += kPre) {
            int v[kPre];
#pragma unroll
            for (int u = 0; u < kPre; ++u) v[u] = bb + u < b1 ? __hip_atomic_load(&pop[bb + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                if (tile_keys && drop_sparse && v[u] > 0 && v[u] < kSparseTile) {
                    const int key = (int)(bb + u);
                    atomicAdd(&tile_start[n_keys + (key / tiles) * tiles_y + (key % tiles) / tiles_x], v[u]);  // its band's population
                    v[u] = 0;
                    pop[bb + u] = 0;
                    tile_cursor[bb + u] = kSparseMarker;  // pass 3 sends this tile's keypoints to the gather path's list
                }
                sum += v[u];
            }
        }
        const int incl = wave_inclusive_scan(sum);
        __syncthreads();  // (wave_tot is used twice)
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int base = first + incl - sum, total = first;
        for (int i = 0; i < kScanWaves; ++i) {
            if (i < w) base += wave_tot[i];
            total += wave_tot[i];
        }
        for (int64_t bb = b0; bb < b1; bb += kPre) {
            int v[kPre];
#pragma unroll
            for (int u = 0; u < kPre; ++u) v[u] = bb + u < b1 ? __hip_atomic_load(&pop[bb + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                if (bb + u < b1) pop[bb + u] = base;
                base += v[u];
            }
        }
        return total;
    };
    const int n_tile_kp = scan(tile_start, n_keys, 0, true);
    __threadfence();  // the bands' populations now hold the thin tiles' keypoints
    __syncthreads();
    const int n_all = scan(tile_start + n_keys, n_bkeys, n_tile_kp, false);
    if (threadIdx.x == 0) {
        tile_start[n_keys + n_bkeys] = n_all;  // (with no band at all this is tile_start[n_keys]: where the last tile ends)
        *slow_count = n_all - n_tile_kp;
    }
}

// Pass 3: scatter keypoints into their tile's segment / the slow list; finalise erased keypoints.
__global__ __launch_bounds__(256) void bin_scatter_kernel(BinArgs a)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = g < a.n_kp;
    const int key = live ? a.kp_key[g] : -1;
    // one key space: a tile's keypoints go to its segment of sorted_kp, the gather path's to their band's segment of its
    // list (bands of kTileH rows, pair after pair: describe_kernel's wavefronts in flight then work on one strip of one
    // pair's integral instead of on a whole layer's keypoints all over it -- half the HBM fetches)
    int ukey = -1;
    bool to_slow = false;
    if (key >= 0) {
        if (a.tile_cursor[key] < 0) {  // a thin tile, handed over by pass 2
            const int tiles = a.tiles_x * a.tiles_y;
            to_slow = true;
            ukey = (int)a.n_keys + (key / tiles) * a.tiles_y + (key % tiles) / a.tiles_x;
        } else {
            ukey = key;
        }
    } else if (key <= -3) {
        to_slow = true;
        ukey = (int)a.n_keys + (-3 - key);
    }
    if (!live) return;
    // one cursor bump per distinct key in the wave, the wave's keypoints of that key in order behind it.  Who leads which
    // key is worked out first, without touching memory, so that all the bumps (which return a value: a memory round trip
    // each) are in flight together.
    int leader = 0, rank = 0, count = 0;
    for (unsigned long long todo = __ballot(ukey >= 0); todo;) {
        const int first = __ffsll((long long)todo) - 1;
        const int k = __shfl(ukey, first);
        const unsigned long long same = __ballot(ukey == k);
        if (ukey == k) {
            leader = first;
            rank = __popcll(same & ((1ull << lane_id()) - 1));
            count = __popcll(same);
        }
        todo &= ~same;
    }
    int pos = 0;
    if (ukey >= 0 && lane_id() == leader) pos = a.tile_start[ukey] + atomicAdd(&a.tile_cursor[ukey], count);
    pos = __shfl(pos, leader) + rank;
    if (to_slow) {
        a.slow_list[pos - a.tile_start[a.n_keys]] = (int)g;
    } else if (key >= 0) {
        const mofreak_keypoint kp = a.kps[g];
        SortedKp s;
        s.x = kp.x;
        s.y = kp.y;
        s.packed = (uint32_t)(int)ceilf(kp.size) | ((uint32_t)(((int)kp.size) / 2) << 8) |
                   ((uint32_t)a.kp_scale[g] << 16);  // :293-295 ROI side / half, FREAK scale index (pass 1)
        s.g = (int)g;
        a.sorted_kp[pos] = s;
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

template <bool STAMPS>
__global__ __launch_bounds__(kTileThreads, 4) void tile_kernel(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;  // LDS address of the block
    unsigned long long *s_stamps = reinterpret_cast<unsigned long long *>(lds + kOffStamps);
    if (STAMPS && threadIdx.x == 0)
        for (int i = 0; i < kStampLds; ++i) s_stamps[i] = 0;
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


    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // the same in every lane: kept in a scalar register, and so is what is computed from it
    const int W = a.f.W, H = a.f.H;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    // integral halo of this call: the largest pattern the binning pass met, in steps of 8 pixels
    const int halo = min(kTileHalo, max(kMinHalo, (*a.max_ps + 7) & ~7));
    const int halo_x = (halo + 15) & ~15;  // 16-byte pieces of a region row start on 16 bytes of the frame row
    const int RW = kTileW + 2 * halo_x, RH = kTileH + 2 * halo;
    const int ox = tx * kTileW - halo_x, oy = ty * kTileH - halo;                  // integral region origin
    const uint8_t *cur = a.f.cur + (int64_t)pair * a.f.pair_stride;
    const uint8_t *prev = a.f.prev + (int64_t)pair * a.f.pair_stride;
    const bool fast8 = (((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)a.f.row_stride) & 7) == 0;
    const int64_t out_base = a.kp_offsets ? 0 : (int64_t)pair * a.n_kp;
    const SortedKp *tile_kps = a.sorted_kp + kp_begin;
    const SmallTables *st = a.small;
    const int bit_mode = st->bit_mode, mip_theta = st->mip_theta;
    const int bit_ge = bit_mode == MOFREAK_BITS_SSE ? 1 : 0;
    const int bit_flip = bit_mode == MOFREAK_BITS_SSE || bit_mode == MOFREAK_BITS_NATURAL ? 0 : 0x80;
    const bool orientation_normalized = st->orientation_normalized != 0;

    ThetaBound *s_theta = reinterpret_cast<ThetaBound *>(lds + kOffTheta);

    uint2 *s_mot = reinterpret_cast<uint2 *>(lds + kOffMot);
    KpRec *kf = reinterpret_cast<KpRec *>(lds + kOffKf);
    uint2 *kint = reinterpret_cast<uint2 *>(lds + kOffKint);
    const bool one_batch = n_tile_kp <= kBatch;
    // The binning pass recorded the smallest and largest ROI side of the tile: equal in the usual case.
    const int tile_L = (int)~a.tile_lmin_c[key];
    const bool uniform = tile_L == (int)a.tile_lmax[key];

    // (a last group of fewer than kGroup keypoints is filled up with copies of the batch's last one: stage 3 then has whole
    // groups only, and what it computes for the copies is not stored)
    auto table_block = [](uint32_t scale) { return scale * (uint32_t)(kNbOrientation * kNbPoints * 16); };
    auto make_records = [&](int b0, int nb, bool for_mip, int rt) {  // rt: the thread's index among the threads that make records
        if (rt >= 0 && rt < ((nb + kGroup - 1) & ~(kGroup - 1))) {
            const SortedKp kp = tile_kps[b0 + min(rt, nb - 1)];
            KpRec k;
            k.kx = kp.x;
            k.ky = kp.y;
            k.g = kp.g;
            k.pk = (uint16_t)((kp.packed >> 16) | (kp.packed & 0xff) << 6 | ((kp.packed >> 8) & 0xff) << 11);
            k.unused = 0;
            kf[rt] = k;
            // a keypoint at integer coordinates: LDS address of its own corner (ky, kx) in the integral; else the top bit
            const int xi = (int)kp.x, yi = (int)kp.y;
            const bool integral = (float)xi == kp.x && (float)yi == kp.y;
            // .y, stage 1: the ROI's first byte in the staged rows (:293-295, :460 float -> int parameters); stage 3: byte offset
            // of the keypoint's 43 un-rotated points in the pattern tables (16-byte entries), which the orientation pass
            // replaces by the offset of the rotated ones
            const int half = (int)((kp.packed >> 8) & 0xff);
            kint[rt] = make_uint2(integral ? lds0 + kOffIntegral + 2 * (kIColOff + (yi - oy) * kIPitch + (xi - ox)) : 0x80000000u,
                                   for_mip ? (uint32_t)((yi - half - oy + 1) * kTileStagePitch + (xi - half - ox)) : table_block(kp.packed >> 16));
        }
    };

    // ================= stage 0: the region's gray bytes -> LDS, 16 per lane and step.  Region row r goes to LDS row
    // r + 1: `current` at byte 0, `previous` at byte kTileRW.  The 16 lanes of a lane row share a region row (as in
    // the row pass), a workgroup takes 32 rows (kStageRows) per step: per-lane offsets are computed once, a step moves the scalar
    // base.  All loads are issued before anything waits on one (a branch per load would make the compiler drain the
    // memory queue at every join); the small tables and the first batch's keypoint records are fetched behind them.
    // Waves 0 .. kMipWaves - 1 take no part in it: when the tile qualifies (mip_lane below) they compute the MIP of the
    // tile's keypoints meanwhile, a lane per keypoint, straight from the frames -- the time the others spend waiting for
    // memory.  stid: a thread's index among the staging threads (negative in the MIP waves).
    const int stid = tid - 64 * kMipWaves;
    const bool stager = stid >= 0;
    const int runs = RW >> 4;
    const int sq = stid & 15, sr = stid >> 4;
    const bool inside = ox >= 0 && oy >= 0 && ox + RW <= W && oy + RH <= H;  // the whole region lies in the image
    const bool wide = fast8 && (inside || (W & 7) == 0);
    // The lane-per-keypoint MIP: one ROI side in the tile, a side it is compiled for, frames and rows on 4-byte boundaries
    // (a lane fetches 12 to 20 bytes from each ROI row's start rounded down to 4), not the tile(s) at the end of the
    // frame's last rows, where that fetch could pass the end of the caller's buffer, and at most kBatch keypoints (64 per
    // MIP wave; a crowded tile's batches go through stage 1 below).
    const bool mip_lane = uniform && tile_L >= kMipLaneMinL && tile_L <= kMipLaneMaxL && (!STAMPS || tile_L == 12) &&
                          (((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)a.f.row_stride) & 3) == 0 && (int64_t)H * a.f.row_stride < ((int64_t)1 << 32) &&
                          !(ty == a.tiles_y - 1 && (tx + 1) * kTileW + 32 > W) && one_batch;
    // Then nobody reads the staged gray bytes in LDS: the staging waves run the integral's row pass (stage 2a) on the bytes as
    // they arrive in their registers -- a staging step's 16 lanes per region row ARE the row pass's -- and write integral
    // rows, not bytes, in the time the MIP waves still need: no stores and re-loads of the bytes, one workgroup barrier less.
    const bool fuse_rows = mip_lane && wide;
    // per-lane constants of the MIP sampling passes: the LDS address the lane's pixels go to, and -- once the ROI side
    // is known -- the LDS addresses of each pixel's two source rows for a ROI at the region's origin (the second byte
    // of a row pair is the next one: where cv::resize clamps the column instead, its weight is zero) and its weights.
    struct MipLane {
        uint32_t dst_dword, dst_tail;              // where the lane's packed four pixels / its last-pass pixel go
        uint32_t a0[kMipIters], a1[kMipIters];
        uint32_t cxp[kMipIters];                   // the two 11-bit x weights, c0 | c1 << 16
        uint32_t c0ys[kMipIters], c1ys[kMipIters];  // the y weights << 12
    } ml;
    const uint32_t p19 = lds0 + kOffScratch + wave * kP19Wave;  // this wave's pair of 19x19 buffers
    auto load_samples = [&](int L) {
        const uint16_t *pos = a.mip_pos + (int64_t)L * a.mip_stride + lane;  // which dword this lane resamples for this ROI side
        ml.dst_dword = p19 + pos[0];  // byte 0 of the lane's dword: passes 0..3 are its four bytes (tables.cpp)
        ml.dst_tail = p19 + pos[64 * (kMipIters - 1)];
        const MipSample *tab = a.mip_samples + (int64_t)L * a.mip_stride + lane;
#pragma unroll
        for (int u = 0; u < kMipIters; ++u) {
            const MipSample sm = tab[64 * u];
            ml.a0[u] = lds0 + kOffIntegral + sm.off_row0;
            ml.a1[u] = lds0 + kOffIntegral + sm.off_row1;
            ml.cxp[u] = sm.cx;
            ml.c0ys[u] = sm.c0y_s12;
            ml.c1ys[u] = sm.c1y_s12;
        }
    };
    const bool tail_ok = lane + 64 * (kMipIters - 1) < a.mip_n;  // the last pass is a partial one (launch_tile checks mip_n)

    // One 16-pixel piece of a region row of the integral's row pass (stage 2a): |cur - prev| and the running sum inside the lane's
    // 16 pixels from v_sad_u8 on masked dwords, the lane totals scanned across the 16 lanes of the DPP row that share the region
    // row, the piece's 16 sums (modulo 2^16) stored at their place in LDS row r + 1.  cc / pp: the piece's bytes of the two frames;
    // q: the piece's index in the row.  Every lane of a DPP row must call it (the scan), lanes past the row's pieces with any bytes.
    auto integral_row_piece = [&](const LdsU4 cc, const LdsU4 pp, int r, int q) {
        uint32_t pk[8];
        uint32_t acc = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
            const uint32_t x = w4 == 0 ? cc.x : w4 == 1 ? cc.y : w4 == 2 ? cc.z : cc.w;
            const uint32_t y = w4 == 0 ? pp.x : w4 == 1 ? pp.y : w4 == 2 ? pp.z : pp.w;
            // the first one, two, three bytes: the other bytes of y replaced by x's, where they add nothing (one v_bfi
            // per prefix instead of two masks)
            const uint32_t s0 = __builtin_amdgcn_sad_u8(x, (y & 0xffu) | (x & ~0xffu), acc);
            const uint32_t s1 = __builtin_amdgcn_sad_u8(x, (y & 0xffffu) | (x & ~0xffffu), acc);
            const uint32_t s2 = __builtin_amdgcn_sad_u8(x, (y & 0xffffffu) | (x & ~0xffffffu), acc);
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
            uint8_t *row = lds + kOffIntegral + __umul24(r + 1, kTileStagePitch);
            uint4 *dst = reinterpret_cast<uint4 *>(row + (8 + 16 * q) * 2);
            dst[0] = make_uint4(pk_add_u16(pk[0], carry2), pk_add_u16(pk[1], carry2), pk_add_u16(pk[2], carry2), pk_add_u16(pk[3], carry2));
            dst[1] = make_uint4(pk_add_u16(pk[4], carry2), pk_add_u16(pk[5], carry2), pk_add_u16(pk[6], carry2), pk_add_u16(pk[7], carry2));
            if (q == 0) *reinterpret_cast<uint32_t *>(row + (kIColOff - 1) * 2) = 0;  // logical column 0
        }
    };

    if (mip_lane && wave < kMipWaves) {  // (a scalar branch: nothing of the staging path is alive in here)
        // ---- stage 1 for the usual tile, beside stage 0: waves 2 s and 2 s + 1 take keypoints 64 s .. 64 s + 63 (a single
        // batch: mip_lane), one the patch centres {0, 1, 3, 5}, the other {2, 4, 6, 7} -- four waves instead of two for 14 % more
        // instructions (the halves share some resampled cells): the MIP is what the other waves wait for in this stage
        auto run = [&](auto LL, auto CM) __attribute__((always_inline)) {
            constexpr int L = decltype(LL)::value;
            const int k = 64 * (wave >> 1) + lane;
            uint32_t roi;  // the ROI's byte offset in both frames
            {
                const SortedKp kp = tile_kps[min(k, n_tile_kp - 1)];  // (a partial wave: the spare lanes repeat the last keypoint)
                // :293-295 with :460's float -> int parameters: the ROI starts at (x - size / 2, y - size / 2)
                const int half = (int)((kp.packed >> 8) & 0xff);
                roi = (uint32_t)((int)kp.y - half) * (uint32_t)a.f.row_stride + (uint32_t)((int)kp.x - half);
            }
            const uint2 mv = mip_lane_keypoint<L, decltype(CM)::value>(cur, prev, roi, a.f.row_stride, mip_theta);
            // kept for one 16-byte store per descriptor at the end of stage 3 (spare lanes: spare slots): .x = the bytes of
            // centres 0, 1, 3, 5, .y = those of 2, 4, 6, 7 (put in order there)
            reinterpret_cast<uint32_t *>(s_mot)[2 * k + (wave & 1)] = mv.x;
        };
        auto run_half = [&](auto LL) __attribute__((always_inline)) {
            if (wave & 1)
                run(LL, std::integral_constant<int, kMipMaskB>{});
            else
                run(LL, std::integral_constant<int, kMipMaskA>{});
        };
        switch (64 * (wave >> 1) < n_tile_kp ? tile_L : 0) {  // (a sparse tile: the second pair of waves has no keypoint)
        case 0: break;
        case 7: run_half(std::integral_constant<int, 7>{}); break;
        case 8: run_half(std::integral_constant<int, 8>{}); break;
        case 9: run_half(std::integral_constant<int, 9>{}); break;
        case 10: run_half(std::integral_constant<int, 10>{}); break;
        case 11: run_half(std::integral_constant<int, 11>{}); break;
        case 12: run_half(std::integral_constant<int, 12>{}); break;
        default: run_half(std::integral_constant<int, 13>{}); break;
        }
        TILE_STAMP(1);  // (thread 0 is in a MIP wave: this interval is the lane-per-keypoint MIP, the next one its wait for the staging waves)
    } else {
        const uint32_t stage_lds = lds0 + kOffIntegral + (sr + 1) * kTileStagePitch + 16 * sq;
        Px16 v[2][kStageIters];
        if (wide && stager) {
            if (inside) {
                const uint32_t voff = (uint32_t)sr * (uint32_t)a.f.row_stride + 16u * (uint32_t)min(sq, runs - 1);
#pragma unroll
                for (int u = 0; u < kStageIters; ++u) {
                    const int64_t base = (int64_t)(oy + min(kStageRows * u, RH - kStageRows)) * a.f.row_stride + ox;  // (a last partial step re-reads rows)
                    v[0][u] = *reinterpret_cast<const Px16 *>(cur + base + voff);
                    v[1][u] = *reinterpret_cast<const Px16 *>(prev + base + voff);
                }
            } else {
                // a tile on the image border, W a multiple of 8: every 8-byte half of a piece is all inside or all outside
                // the image.  Outside pixels are never read by a keypoint that passed the border tests: any value will do,
                // so the loads are clamped into the image instead of branching.
                const int gx = ox + 16 * min(sq, runs - 1);
                const uint32_t xlo = (uint32_t)min(max(gx, 0), W - 8), xhi = (uint32_t)min(max(gx + 8, 0), W - 8);
#pragma unroll
                for (int u = 0; u < kStageIters; ++u) {
                    const int gy = min(max(oy + min(kStageRows * u, RH - kStageRows) + sr, 0), H - 1);
                    const int64_t ro = (int64_t)gy * a.f.row_stride;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const uint8_t *row = (f ? prev : cur) + ro;
                        const uint2 lo = *reinterpret_cast<const uint2 *>(row + xlo);
                        const uint2 hi = *reinterpret_cast<const uint2 *>(row + xhi);
                        v[f][u].w[0] = lo.x;
                        v[f][u].w[1] = lo.y;
                        v[f][u].w[2] = hi.x;
                        v[f][u].w[3] = hi.y;
                    }
                }
            }
        }

        if (uniform && !mip_lane) load_samples(tile_L);  // one ROI side in the whole tile (the usual case): its samples stay in registers
        make_records(0, min(kBatch, n_tile_kp), !mip_lane, stid);
        if (stid >= 0 && stid < kThetaBounds) s_theta[stid] = a.theta[stid];
        if (fuse_rows) {
            if (stager) {
                for (int i = stid; i < kIPitchDw; i += kStageThreads) reinterpret_cast<uint32_t *>(lds + kOffIntegral)[i] = 0;  // integral row 0
#pragma unroll
                for (int u = 0; u < kStageIters; ++u) {
                    if (kStageRows * u < RH)  // (a last partial step: the rows it re-read are written again, with the same sums)
                        integral_row_piece(LdsU4{v[0][u].w[0], v[0][u].w[1], v[0][u].w[2], v[0][u].w[3]}, LdsU4{v[1][u].w[0], v[1][u].w[1], v[1][u].w[2], v[1][u].w[3]},
                                           min(kStageRows * u, RH - kStageRows) + sr, sq);
                }
            }
        } else if (wide) {
            if (stager && sq < runs) {
#pragma unroll
                for (int u = 0; u < kStageIters; ++u) {
                    if (kStageRows * u < RH) {  // a last partial step: the rows it re-read are written again, with the same bytes
                        const uint32_t d = stage_lds + min(kStageRows * u, RH - kStageRows) * kTileStagePitch;
                        lds_st<LdsU4>(d, LdsU4{v[0][u].w[0], v[0][u].w[1], v[0][u].w[2], v[0][u].w[3]});
                        lds_st<LdsU4>(d + kTileRW, LdsU4{v[1][u].w[0], v[1][u].w[1], v[1][u].w[2], v[1][u].w[3]});
                    }
                }
            }
        } else if (stager) {  // unaligned frames or an odd width: byte by byte, zero outside the image
            for (int t = stid; t < 2 * RH * runs; t += kStageThreads) {
                const int fr = t >= RH * runs ? 1 : 0, tt = t - fr * RH * runs;
                const int r = tt / runs, q = tt - r * runs;
                const int gy = oy + r, gx = ox + 16 * q;
                const uint8_t *row = (fr ? prev : cur) + (int64_t)gy * a.f.row_stride;
                uint32_t w4[4] = {0, 0, 0, 0};
                if (gy >= 0 && gy < H) {
                    for (int k = 0; k < 16; ++k) {
                        const int x = gx + k;
                        if (x >= 0 && x < W) w4[k >> 2] |= (uint32_t)row[x] << (8 * (k & 3));
                    }
                }
                lds_st<LdsU4>(lds0 + kOffIntegral + (r + 1) * kTileStagePitch + fr * kTileRW + 16 * q, LdsU4{w4[0], w4[1], w4[2], w4[3]});
            }
        }
    }
    __syncthreads();  TILE_STAMP(0);

    // ================= stage 1: MIP, one wave per keypoint out of the staged rows -- for tiles the lane-per-keypoint form
    // above does not take (mixed ROI sides, sides outside kMipLaneMinL .. kMipLaneMaxL, frames off 4-byte boundaries)
    if (!mip_lane) {  // <stage 1>  (mofreak_amd/tools/ab_tile.py --ablate builds copies of this file with a stage's block disabled)
        // per-lane constants of the bit pass: lane = 8*centre + offset (MoFREAKUtilities.cpp:56-70, 308-316)
        const int mc = lane >> 3, mi = lane & 7;
        const int mcx = (0xDDD99555u >> (4 * mc)) & 15, mcy = (0xD95D5D95u >> (4 * mc)) & 15;
        const int mdx = (int)((0x14787410u >> (4 * mi)) & 15) - 4, mdy = (int)((0x10147874u >> (4 * mi)) & 15) - 4;
        const int base_c = (mcy - 1) * kPatch + (mcx - 1);
        const int base_p = kP19Pad + (mcy + mdy - 1) * kPatch + (mcx + mdx - 1);
        const uint32_t cw = p19 + 4 * (base_c >> 2), pw = p19 + 4 * (base_p >> 2);  // covering dwords
        const int cs = base_c & 3, ps = base_p & 3;                                // byte shifts

        // one sampled pixel of one keypoint: four source bytes -> horizontal step as a packed dot product -> vertical step
        auto sample = [&](const MipLane &c, int u, uint32_t roi) -> uint32_t {
            const uint32_t b0 = c.a0[u] + roi, b1 = c.a1[u] + roi;
            const uint32_t r0 = (uint32_t)lds_ld<uint8_t>(b0) | (uint32_t)lds_ld<uint8_t>(b0 + 1) << 16;
            const uint32_t r1 = (uint32_t)lds_ld<uint8_t>(b1) | (uint32_t)lds_ld<uint8_t>(b1 + 1) << 16;
            const u16x2 wx = __builtin_bit_cast(u16x2, c.cxp[u]);
            const uint32_t t0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, r0), wx, 0u, false);
            const uint32_t t1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, r1), wx, 0u, false);
            return (uint32_t)resize_y(t0, t1, c.c0ys[u], c.c1ys[u]);
        };
        // lane = 8*centre + offset: the two 9-byte strips sit at arbitrary byte offsets: fetch the covering aligned
        // dwords (a byte-wise formulation lets the compiler fuse the loads into misaligned ds_read_b64s, 64 cycles
        // each) and shift the strips out; SSD = sum c^2 + sum p^2 - 2 sum c*p over the first eight bytes (packed u8
        // dot products) + the ninth byte's squared difference
        // a lane's five resampled pixels -> LDS: the first four are the bytes of one dword
        auto put_pixels = [&](const MipLane &c, const uint32_t (&px)[kMipIters]) {
            lds_st<uint32_t>(c.dst_dword, px[0] | px[1] << 8 | px[2] << 16 | px[3] << 24);
            if (tail_ok) lds_st<uint8_t>(c.dst_tail, (uint8_t)px[4]);
        };
        struct Strips {
            uint32_t cd[3], pd[3];
        };
        auto read_strips = [&]() -> Strips {
            Strips t;
#pragma unroll
            for (int w3 = 0; w3 < 3; ++w3) {
                t.cd[w3] = lds_ld<uint32_t>(cw + 4 * w3);
                t.pd[w3] = lds_ld<uint32_t>(pw + 4 * w3);
            }
            return t;
        };
        auto strip_bits = [&](const Strips &t) -> uint64_t {
            const uint32_t c0 = __builtin_amdgcn_alignbyte(t.cd[1], t.cd[0], cs), c1 = __builtin_amdgcn_alignbyte(t.cd[2], t.cd[1], cs);
            const uint32_t p0 = __builtin_amdgcn_alignbyte(t.pd[1], t.pd[0], ps), p1 = __builtin_amdgcn_alignbyte(t.pd[2], t.pd[1], ps);
            const int d8 = (int)((t.cd[2] >> (8 * cs)) & 0xffu) - (int)((t.pd[2] >> (8 * ps)) & 0xffu);
            const uint32_t sq = __builtin_amdgcn_udot4(c0, c0, __builtin_amdgcn_udot4(c1, c1, (uint32_t)__mul24(d8, d8), false), false) +
                                __builtin_amdgcn_udot4(p0, p0, __builtin_amdgcn_udot4(p1, p1, 0u, false), false);
            const uint32_t cross = __builtin_amdgcn_udot4(c0, p0, __builtin_amdgcn_udot4(c1, p1, 0u, false), false);
            return __ballot((int)(sq - 2u * cross) > mip_theta);
        };
        auto put_motion = [&](int kk, uint64_t mot) {
            const uint2 mv = make_uint2((uint32_t)mot, (uint32_t)(mot >> 32));
            if (one_batch)  // kept for one 16-byte store per descriptor at the end of stage 3
                s_mot[kk] = mv;
            else  // a crowded tile (several batches) sends its motion bytes out now
                *reinterpret_cast<uint2 *>(a.out_desc + (out_base + kf[kk].g) * 16 + 8) = mv;
        };

        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (b0 > 0) {  // further batches of a crowded tile: their records
                __syncthreads();
                make_records(b0, nb, true, tid);
                __syncthreads();
            }
            if (uniform) {
                // Per wave, two keypoints at a time (their LDS reads are issued together, so one's latency hides under
                // the other's arithmetic): the sampled pixels of the two 19x19 resamples -> registers; then, one
                // keypoint after the other through the wave's buffer -- same wave, so only wave-level syncs --
                // pixels -> LDS, strips back, SSDs, ballot.  A last odd keypoint is simply done twice.
                const MipLane c = ml;  // never changes inside this loop
                for (int kk = wave; kk < nb; kk += 2 * kTileWaves) {
                    const int kk2 = kk + kTileWaves;
                    const bool two = kk2 < nb;
                    const uint32_t roi = kint[kk].y, roi2 = kint[two ? kk2 : kk].y;
                    uint32_t px[2][kMipIters];
#pragma unroll
                    for (int u = 0; u < kMipIters; ++u) {
                        px[0][u] = sample(c, u, roi);
                        px[1][u] = sample(c, u, roi2);
                    }
                    put_pixels(c, px[0]);
                    wave_lds_sync();
                    const Strips s1 = read_strips();
                    wave_lds_sync();
                    put_pixels(c, px[1]);
                    wave_lds_sync();
                    const Strips s2 = read_strips();
                    const uint64_t mot = strip_bits(s1), mot2 = strip_bits(s2);
                    if (lane == 0) {
                        put_motion(kk, mot);
                        if (two) put_motion(kk2, mot2);
                    }
                    wave_lds_sync();  // the next pair of keypoints overwrites the 19x19 buffers
                }
            } else {
                // mixed ROI sides: rare; one keypoint at a time, reloading the samples when the side changes
                int have_L = -1;
                for (int kk = wave; kk < nb; kk += kTileWaves) {
                    const KpRec m = kf[kk];
                    const int L = (m.pk >> 6) & 31;
                    if (L != have_L) {
                        have_L = L;
                        load_samples(L);
                    }
                    const uint32_t roi = kint[kk].y;
                    uint32_t px[kMipIters];
#pragma unroll
                    for (int u = 0; u < kMipIters; ++u) px[u] = sample(ml, u, roi);
                    put_pixels(ml, px);
                    wave_lds_sync();
                    const uint64_t mot = strip_bits(read_strips());
                    if (lane == 0) put_motion(kk, mot);
                    wave_lds_sync();
                }
            }
        }
        __syncthreads();
        // a single batch keeps its records for stage 3: the slot that held the ROI offsets now takes the table blocks (stage
        // 2's barriers come before anyone reads them)
        if (one_batch && tid < ((n_tile_kp + kGroup - 1) & ~(kGroup - 1))) kint[tid].y = table_block(kf[tid].pk & 63);
    }
    TILE_STAMP(1);

    // ================= stage 2: integral of |cur - prev| over tile + halo, modulo 2^16, in LDS
    {  // <stage 2>
        // 2a: row pass, no workgroup barrier.  The 16 lanes of a DPP row share one region row: a lane owns 16 pixels,
        //     |cur - prev| and the running sum inside them come from v_sad_u8 on masked dwords, the lane totals are
        //     scanned across the row with four DPP adds.  A wave takes four region rows per step.
        if (!fuse_rows) {
        const int rr = lane >> 4, q = lane & 15;
        LdsU4 c[kRowGroupIters], p[kRowGroupIters];
#pragma unroll
        for (int u = 0; u < kRowGroupIters; ++u) {  // all of a wave's rows are read before any is overwritten
            const int r = min(4 * (wave + kTileWaves * u) + rr, RH - 1);
            const uint32_t row = lds0 + kOffIntegral + (r + 1) * kTileStagePitch + 16 * min(q, runs - 1);
            c[u] = lds_ld<LdsU4>(row);
            p[u] = lds_ld<LdsU4>(row + kTileRW);
        }
        for (int i = tid; i < kIPitchDw; i += kTileThreads) reinterpret_cast<uint32_t *>(lds + kOffIntegral)[i] = 0;  // integral row 0
#pragma unroll
        for (int u = 0; u < kRowGroupIters; ++u) integral_row_piece(c[u], p[u], 4 * (wave + kTileWaves * u) + rr, q);
        __syncthreads();
        }
        TILE_STAMP(2);
        // 2b: column pass.  A thread owns a 16-row segment of two adjacent dword columns (four pixels, 8-byte LDS
        //     accesses): running packed sums in registers, segment totals -> LDS; after the barrier it adds the totals
        //     of the segments above.
        const int n_qcols = RW >> 2, n_blocks = RH / kColBlockRows;
        const uint32_t m20 = (1u << 20) / (uint32_t)n_qcols + 1;  // t / n_qcols for t < 2^9 as (t * m20) >> 20
        const bool col_ok = tid < n_blocks * n_qcols;
        const int cj = col_ok ? (int)(((uint32_t)tid * m20) >> 20) : 0, cc = col_ok ? tid - cj * n_qcols : 0;
        const uint32_t col0 = lds0 + kOffIntegral + (cj * kColBlockRows + 1) * kTileStagePitch + (kIColOff + 1) * 2 + 8 * cc;
        const uint32_t carry = lds0 + kOffScratch;  // uint2 [block][kMaxQcols]
        LdsU2 cv[kColBlockRows];
#pragma unroll
        for (int r = 0; r < kColBlockRows; ++r) cv[r] = lds_ld<LdsU2>(col0 + r * kTileStagePitch);
        LdsU2 acc = {0, 0};
#pragma unroll
        for (int r = 0; r < kColBlockRows; ++r) {
            acc.x = pk_add_u16(acc.x, cv[r].x);
            acc.y = pk_add_u16(acc.y, cv[r].y);
            cv[r] = acc;
        }
        if (col_ok) lds_st<LdsU2>(carry + (cj * kMaxQcols + cc) * 8, acc);
        __syncthreads();  TILE_STAMP(3);
        if (col_ok) {
            LdsU2 add = {0, 0};
#pragma unroll
            for (int jj = 0; jj < kMaxColBlocks - 1; ++jj) {
                const LdsU2 tot = lds_ld<LdsU2>(carry + (jj * kMaxQcols + cc) * 8);
                add.x = pk_add_u16(add.x, jj < cj ? tot.x : 0u);
                add.y = pk_add_u16(add.y, jj < cj ? tot.y : 0u);
            }
#pragma unroll
            for (int r = 0; r < kColBlockRows; ++r)
                lds_st<LdsU2>(col0 + r * kTileStagePitch, LdsU2{pk_add_u16(cv[r].x, add.x), pk_add_u16(cv[r].y, add.y)});
        }
        __syncthreads();  TILE_STAMP(4);
    }

    // ================= stage 3: FREAK on the difference image, one wave per group of four keypoints
    {  // <stage 3>
        const uint32_t vv = lds0 + kOffScratch + wave * (kGroup * kVStride);   // this wave's box means [kGroup][kVStride]
        const uint32_t ibase = lds0 + kOffIntegral + 2 * (kIColOff - oy * kIPitch - ox);
        // per-lane constants (host-built, TileLane): the box-mean tasks of a group -- the outer two rings (whose boxes may
        // need slices) of all four keypoints first; tasks past the group's 172 recompute keypoint 0's point 42 (same
        // value, same address: harmless) -- the lane's description pair, and for the orientation pass (16 lanes per
        // keypoint, three of the 45 pairs each) the pairs and their weights as floats: w / 2048 is exact, and so is its
        // product with a difference of two bytes; truncated like the reference's integer division.
        const TileLane tl = a.lanes[lane];
        int task_kq[kBoxIters], task_p[kBoxIters];
#pragma unroll
        for (int u = 0; u < kBoxIters; ++u) {
            task_kq[u] = tl.task[u] & 0xff;
            task_p[u] = tl.task[u] >> 8;
        }
        const int oq = lane >> 4, osub = lane & 15;
        uint32_t opi[3], opj[3];
        float owx[3], owy[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            opi[k] = vv + oq * kVStride + tl.opi[k];
            opj[k] = vv + oq * kVStride + tl.opj[k];
            owx[k] = tl.owx[k];
            owy[k] = tl.owy[k];
        }
        const uint32_t pi = vv + tl.pair_i, pj = vv + tl.pair_j;
        // constant per lane: where a full group's tasks find their keypoint's record and put their box mean
        uint32_t rec_full[kBoxIters], kint_full[kBoxIters], vdst_full[kBoxIters];
#pragma unroll
        for (int u = 0; u < kBoxIters; ++u) {
            rec_full[u] = lds0 + kOffKf + task_kq[u] * 16;
            kint_full[u] = lds0 + kOffKint + task_kq[u] * 8;
            vdst_full[u] = vv + task_kq[u] * kVStride + task_p[u];
        }
        struct Task {       // one box-mean task of a group
            uint32_t c0, c1;    // (kx, ky) as bits -- or, when every keypoint of the batch has integer coordinates, the LDS
                                // addresses of its corner in the integral and of its record
            uint32_t tab;       // byte offset of the keypoint's 43 points (un-rotated in pass A, rotated in pass B) in the tables
        };
        const float box_margin = a.box_margin;

        for (int b0 = 0; b0 < n_tile_kp; b0 += kBatch) {
            const int nb = min(kBatch, n_tile_kp - b0);
            if (!one_batch) {  // crowded tile: the records of this batch (a single batch still has them from stage 0)
                __syncthreads();
                make_records(b0, nb, false, tid);
                __syncthreads();
            }
            // What the batch's keypoints have in common (every wave works it out for itself from the records): one
            // FREAK scale -- then a lane's un-rotated pattern points are loop invariants -- and integer coordinates --
            // then boxes are fixed offsets from the keypoint (BoxInt) and no float arithmetic is needed to place them.
            bool one_scale, all_int;
            {
                const int k1 = min(lane, nb - 1), k2 = min(lane + 64, nb - 1);
                const uint32_t s0 = kf[0].pk & 63;
                one_scale = __all((kf[k1].pk & 63) == s0 && (kf[k2].pk & 63) == s0);
                all_int = __all((int)(kint[k1].x | kint[k2].x) >= 0) && box_margin < 1.0f;
            }
            // The rest of the batch is compiled twice, for keypoints at integer coordinates and for the general case: the
            // choice is made once per batch here instead of once per box inside the loops.
            auto run_batch = [&](auto int_tag) {
                constexpr bool kAllInt = decltype(int_tag)::value;
                // (groups are whole: make_records fills the last one up)
                auto group_tasks = [&](int kbase, Task (&t)[kBoxIters]) {
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        const uint32_t ra = rec_full[u] + kbase * 16, ka = kint_full[u] + kbase * 8;
                        if (kAllInt) {
                            const LdsU2 ct = lds_ld<LdsU2>(ka);
                            t[u].c0 = ct.x;
                            t[u].c1 = ra;
                            t[u].tab = ct.y;
                        } else {
                            const LdsU2 xy = lds_ld<LdsU2>(ra);
                            t[u].c0 = xy.x;
                            t[u].c1 = xy.y;
                            t[u].tab = lds_ld<uint32_t>(ka + 4);
                        }
                    }
                };
                // entry: the task's BoxInt (kAllInt) or PatternPoint, as loaded; e: its byte offset in the tables
                auto box = [&](const Task t, const LdsU4 entry, uint32_t e) -> int {
                    if (kAllInt) {
                        const BoxInt B = __builtin_bit_cast(BoxInt, entry);
                        if (B.margin > box_margin) return mean_intensity_int(t.c0, B);
                        // a corner too close to a rounding boundary (rare): the float expressions
                        return mean_intensity_tile(ibase, lds_ld<float>(t.c1), lds_ld<float>(t.c1 + 4), *reinterpret_cast<const PatternPoint *>(reinterpret_cast<const uint8_t *>(a.lut) + e));
                    }
                    return mean_intensity_tile(ibase, __builtin_bit_cast(float, t.c0), __builtin_bit_cast(float, t.c1), __builtin_bit_cast(PatternPoint, entry));
                };
                // both tables have 16-byte entries with the same indexing: one (wave-uniform) base, 32-bit byte offsets
                const uint8_t *table = kAllInt ? reinterpret_cast<const uint8_t *>(a.lut_int) : reinterpret_cast<const uint8_t *>(a.lut);
                auto load_entry = [&](uint32_t e) -> LdsU4 { return *reinterpret_cast<const LdsU4 *>(table + e); };
                auto entry_of = [&](const Task t, int u) -> uint32_t { return t.tab + task_p[u] * 16; };

                // ---- pass A over the wave's groups: un-rotated box means, orientation sums, theta -> the records
                if (orientation_normalized) {
                    LdsU4 E0[kBoxIters];   // the lane's un-rotated entries, kept across groups of one scale
                    uint32_t have_tab = ~0u;
                    if (one_scale) {
                        have_tab = (kf[0].pk & 63) * (uint32_t)(kNbOrientation * kNbPoints * 16);
#pragma unroll
                        for (int u = 0; u < kBoxIters; ++u) E0[u] = load_entry(have_tab + task_p[u] * 16);
                    }
                    const uint32_t dirs = lds0 + kOffScratch + kDirsInScratch + wave * kDirsPerWave;  // the wave's keypoints' orientation sums: group j's at [4 j .. 4 j + 3]
                    int j = 0;
                    for (int kbase = wave * kGroup; kbase < nb; kbase += kGroup * kTileWaves, ++j) {
                        const int last = min(kGroup, nb - kbase) - 1;
                        Task t[kBoxIters];
                        group_tasks(kbase, t);
                        if (!one_scale) {
#pragma unroll
                            for (int u = 0; u < kBoxIters; ++u)
                                if (t[u].tab != have_tab) E0[u] = load_entry(entry_of(t[u], u));
                            have_tab = t[1].tab == t[0].tab && t[kBoxIters - 1].tab == t[0].tab ? t[0].tab : ~0u;
                        }
#pragma unroll
                        for (int u = 0; u < kBoxIters; ++u)
                            lds_st<uint8_t>(vdst_full[u], (uint8_t)box(t[u], E0[u], entry_of(t[u], u)));
                        wave_lds_sync();
                        int direction0 = 0, direction1 = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float delta = (float)((int)lds_ld<uint8_t>(opi[k]) - (int)lds_ld<uint8_t>(opj[k]));
                            direction0 += (int)(delta * owx[k]);  // C division by 2048: truncates toward zero, per term
                            direction1 += (int)(delta * owy[k]);
                        }
                        direction0 = row16_sum(direction0);
                        direction1 = row16_sum(direction1);
                        if (osub == 0 && oq <= last) lds_st<LdsU2>(dirs + (j * kGroup + oq) * 8, LdsU2{(uint32_t)direction0, (uint32_t)direction1});
                        wave_lds_sync();  // the next group overwrites the box means
                    }
                    TILE_STAMP(5);
                    // thetas of all the wave's keypoints in one go: lane = keypoint (group lane / 4, slot lane % 4)
                    {
                        const int kp = (wave + kTileWaves * (lane >> 2)) * kGroup + (lane & 3);
                        if (kp < nb) {
                            const LdsU2 dd = lds_ld<LdsU2>(dirs + lane * 8);  // (lane = 4 * group + slot, as stored)
                            const int2 d = make_int2((int)dd.x, (int)dd.y);
                            const int theta = theta_index(s_theta, d.x, d.y);
                            kint[kp].y = (((kf[kp].pk & 63) * kNbOrientation + theta) * kNbPoints) * 16u;
                            if (a.out_info)
                                *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kp].g) * 4) = make_int4(kf[kp].pk & 63, theta, d.x, d.y);
                        }
                        wave_lds_sync();  // pass B reads the thetas
                    }
                    TILE_STAMP(6);
                } else if (a.out_info) {
                    for (int kbase = wave * kGroup; kbase < nb; kbase += kGroup * kTileWaves)
                        if (lane < min(kGroup, nb - kbase))
                            *reinterpret_cast<int4 *>(a.out_info + (out_base + kf[kbase + lane].g) * 4) = make_int4(kf[kbase + lane].pk & 63, 0, 0, 0);
                }
                // ---- pass B: box means of the rotated pattern, bits, store.  The pattern points (or boxes) of the next
                //      group are fetched (L2) while the current group's boxes are summed.
                LdsU4 ec[kBoxIters];
                Task tc[kBoxIters];
                if (wave * kGroup < nb) {
                    group_tasks(wave * kGroup, tc);
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) ec[u] = load_entry(entry_of(tc[u], u));
                }
                for (int kbase = wave * kGroup; kbase < nb; kbase += kGroup * kTileWaves) {
                    const int last = min(kGroup, nb - kbase) - 1;
                    const int knext = kbase + kGroup * kTileWaves;
                    LdsU4 en[kBoxIters];
                    Task tn[kBoxIters];
                    group_tasks(knext < nb ? knext : kbase, tn);
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) en[u] = load_entry(entry_of(tn[u], u));
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) lds_st<uint8_t>(vdst_full[u], (uint8_t)box(tc[u], ec[u], entry_of(tc[u], u)));
                    wave_lds_sync();
                    // lane = descriptor bit; lane q stores keypoint q's descriptor
                    {
                        int va[kGroup], vb[kGroup];
#pragma unroll
                        for (int qq = 0; qq < kGroup; ++qq) {
                            va[qq] = lds_ld<uint8_t>(pi + qq * kVStride);
                            vb[qq] = lds_ld<uint8_t>(pj + qq * kVStride);
                        }
                        // one comparison for the three bit modes: signed chars compare like their values with the top bit
                        // flipped (bit_flip = 0x80), and a >= b is a + 1 > b (bit_ge = 1)
                        uint2 app = make_uint2(0, 0);
#pragma unroll
                        for (int qq = 0; qq < kGroup; ++qq) {
                            const uint64_t bits = __ballot((va[qq] ^ bit_flip) + bit_ge > (vb[qq] ^ bit_flip));
                            if (lane == qq) app = make_uint2((uint32_t)bits, (uint32_t)(bits >> 32));
                        }
                        if (lane <= last) {  // descriptor and validity flag out, side by side
                            int64_t out_idx = out_base + kf[kbase + lane].g;
#ifdef MOFREAK_DEBUG_BOUNDS
                            if (out_idx < 0 || out_idx >= a.out_items) {
                                atomicOr(&g_tile_oob, 2u);
                                out_idx = 0;
                            }
#endif
                            if (one_batch) {
                                uint2 mot = s_mot[kbase + lane];
                                if (mip_lane)  // the two MIP waves' words: centres (0, 1, 3, 5) and (2, 4, 6, 7) -> 0..3, 4..7
                                    mot = make_uint2(__builtin_amdgcn_perm(mot.y, mot.x, 0x02040100u), __builtin_amdgcn_perm(mot.y, mot.x, 0x07060305u));
                                *reinterpret_cast<uint4 *>(a.out_desc + out_idx * 16) = make_uint4(app.x, app.y, mot.x, mot.y);
                            } else {
                                *reinterpret_cast<uint2 *>(a.out_desc + out_idx * 16) = app;
                            }
                            a.out_valid[out_idx] = 1;
                        }
                    }
                    wave_lds_sync();  // the next group overwrites the box means
#pragma unroll
                    for (int u = 0; u < kBoxIters; ++u) {
                        ec[u] = en[u];
                        tc[u] = tn[u];
                    }
                }
            };
            if (all_int)
                run_batch(std::true_type{});
            else
                run_batch(std::false_type{});
        }
    }
    TILE_STAMP(7);
#ifdef MOFREAK_DEBUG_BOUNDS
    if (tid == 0 && g_tile_oob) atomicOr(a.status, 64);  // mofreak_check_status reports it
#endif
    if (STAMPS && tid == 0)
        for (int i = 0; i < kStampLds; ++i)
            if (s_stamps[i]) atomicAdd(&a.stamps[i], s_stamps[i]);
}

}  // namespace

int launch_bin(const BinArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(a.slow_count, 0, a.counter_bytes, s);  // all counters of the pass (BinArgs)
    if (e != hipSuccess) return (int)e;
    const int blocks = (int)((a.n_kp + 255) / 256);
    if (blocks > 0) hipLaunchKernelGGL(bin_count_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, a.tile_start, a.tile_cursor, a.slow_count, a.max_ps, a.wg_slow, a.wg_maxps, blocks,
                       a.n_kp, a.n_keys, a.n_bkeys, a.tiles_x, a.tiles_y);
    if (blocks > 0) hipLaunchKernelGGL(bin_scatter_kernel, dim3(blocks), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int launch_tile(const TileArgs &a, void *stream)
{
    if (a.mip_n > 64 * kMipIters || a.mip_n <= 64 * (kMipIters - 1) || a.mip_stride < a.mip_n) return (int)hipErrorInvalidValue;
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
