// C ABI of libmofreak_hip.so (include/mofreak_hip.h): context, workspace and launch sequencing.
// No exception leaves this file; every entry point returns a status code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_types.h"

using namespace mofreak;

namespace {
thread_local std::string g_create_error;

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};
}  // namespace

struct mofreak_ctx {
    int device = 0;
    mofreak_params params{};
    Tables tables;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cus = 256;
    // device-resident tables
    PatternPoint *d_lut = nullptr;
    ResizeTap *d_resize = nullptr;
    SmallTables *d_small = nullptr;
    int32_t *d_status = nullptr;
    // workspace (grown on demand, never shrunk)
    DeviceBuffer integral, band_totals, scratch_desc, scratch_valid, compact_offsets, stage[6], offsets_dev;
    DeviceBuffer kp_key, sorted_idx, slow_list, slow_count;  // keypoint binning (slow_count: all its counters, BinArgs)
    const int32_t *slow_band_start = nullptr;  // the last binning pass's band starts (inside slow_count) for the gather path behind it
    int slow_bands_per_pair = 0;
    DeviceBuffer bow_counts, bow_expanded, pair_label, fmt_ws;
    // keypoint detector workspace
    DeviceBuffer det_img, det_score, det_touch, det_status, det_rows, det_cand_xy, det_cand_flag, det_cand_emit, det_cand_spec, det_cand_asked, det_cand_win, det_cand_res, det_layer_start,
        det_emit_count, det_emit_chunks, det_geom, det_emit_offsets, det_out_kps, det_out_offsets, det_out_resp, det_out_layer, det_planes_out, det_hit_mask, det_walk_list, det_walk_count, det_cand_cells, det_tie_list, det_tie_count;
    bool det_maps_dirty = false;  // a detector call stopped half way: its touch / status bytes may still be set
    DetGeom det_geom_sent{};      // what det_geom (the device copy) holds
    bool det_geom_sent_valid = false;
    // |cur - prev| of the pairs of the last mofreak_detect_pairs call, as the detector left it (layer 0 of its planes) --
    // valid while that call had ONE batch; mofreak_compute_stream lets the gather path's integral read it instead of
    // both frames (use_det_diff, for the extract that follows the detector on the same pairs)
    struct DetDiff {
        const uint8_t *base = nullptr;
        int64_t pair_stride = 0;
        int W = 0, H = 0, pairs = 0;
    } det_diff;
    bool use_det_diff = false;
    int det_cand_cap = 131072;  // most candidates (corners) per pair the detector will reserve room for (mofreak_detect_set_capacity)
    bool det_cand_auto = true;  // the room follows the frames (below); false once the caller has named a number: then that is what is reserved
    int det_cand_shift = 0;     // automatic: a call starts from W * H / 8 << det_cand_shift, below the limit; grown (x 4) by a call that met more
    size_t det_counter_bytes = 0;  // row counts + tie counters behind the running total in det_rows
    int64_t det_kp_capacity = 0;
    ThetaBound *d_theta = nullptr;
    unsigned long long *d_stamps = nullptr;  // MOFREAK_TILE_STAMPS=1: per-phase tick sums of the diagnostic tile kernel
    MipSample *d_mip_samples = nullptr;
    uint16_t *d_mip_pos = nullptr;
    TileLane *d_tile_lanes = nullptr;
    BoxInt *d_lut_int = nullptr;
    // mofreak_extract_stream_pipelined: two slots of staging buffers, copy streams and events (created on first use)
    struct Pipe {
        hipStream_t s_in = nullptr, s_out = nullptr;
        hipEvent_t ev_in[2]{}, ev_comp[2]{}, ev_out[2]{};
        void *h_frames[2]{}, *h_rows[2]{}, *h_pair_rows[2]{};
        int64_t *h_count[2]{};
        size_t h_frames_bytes = 0, h_rows_bytes = 0, h_pair_rows_bytes = 0;
        DeviceBuffer d_frames[2], d_rows[2], d_pair_rows[2];
        bool ready = false;
    } pipe;
    int path_mode = MOFREAK_PATH_AUTO;
    int chunk_pairs_hint = 0;
    // optional per-launch timing (mofreak_set_profiling): events around the integral group and the describe launch
    bool profiling = false;
    struct Span {
        hipEvent_t ev[4];
        int64_t pairs, items;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    mofreak_profile prof{};
    mutable std::string err;
};

namespace {

int fail(const mofreak_ctx *ctx, int code, const std::string &msg)
{
    if (ctx)
        ctx->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MOFREAK_ERR_OOM : MOFREAK_ERR_HIP,             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

#define NEED_DEVICE(ctx)                                                                                  \
    do {                                                                                                  \
        if ((ctx)->device < 0)                                                                            \
            return fail(ctx, MOFREAK_ERR_NO_DEVICE, "tables-only context: no device work possible (there is no CPU fallback)"); \
        HIP_TRY(ctx, hipSetDevice((ctx)->device));                                                        \
    } while (0)

int ensure(mofreak_ctx *ctx, DeviceBuffer &b, size_t bytes)
{
    if (b.bytes >= bytes && b.ptr) return MOFREAK_OK;
    if (b.ptr) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.bytes = 0;
    }
    bytes = std::max<size_t>(bytes, 256);
    HIP_TRY(ctx, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    // A new buffer starts as zeros, whatever the allocator hands out (fresh device memory usually is zero, memory this process
    // freed a moment ago is not): a precaution -- the GPU suite and the fuzz tool also pass with new buffers holding a pattern
    // (below) -- that makes a grown buffer a fresh one.  (Buffers grow rarely; the fill is not on any hot path.  On the
    // context's stream and waited for: the stream does not wait for the null stream, where a plain hipMemset would run.)
#ifdef MOFREAK_DEBUG_BOUNDS
    // (bounds-checking build) MOFREAK_FILL_NEW_BUFFERS=165 fills new buffers with 0xA5 instead: whoever reads an entry that
    // nobody wrote shows up in the tests (profiles/README.md, round 4: the suite passes with it)
    const char *fill_env = std::getenv("MOFREAK_FILL_NEW_BUFFERS");
    const int fill = fill_env ? std::atoi(fill_env) & 0xff : 0;
    HIP_TRY(ctx, hipMemsetAsync(b.ptr, fill, bytes, ctx->stream));
#else
    HIP_TRY(ctx, hipMemsetAsync(b.ptr, 0, bytes, ctx->stream));
#endif
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOFREAK_OK;
}

void release(DeviceBuffer &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

inline int integral_pitch(int W) { return ((W + 3) / 4) * 4 + 4; }

// Pairs per chunk: keep the chunk's integral images around 64 MiB so that they are still in the 256 MiB
// Infinity Cache when the describe kernel reads them back.
int choose_chunk(const mofreak_ctx *ctx, int W, int H, int n_pairs)
{
    const size_t per_pair = (size_t)(H + 1) * integral_pitch(W) * sizeof(int32_t);
    int chunk = ctx->chunk_pairs_hint > 0 ? ctx->chunk_pairs_hint : (int)std::max<size_t>(1, ((size_t)64 << 20) / per_pair);
    return std::max(1, std::min(chunk, n_pairs));
}

// Pairs per chunk of the gather path when it runs BEHIND the tile kernel (usually with nothing to do): few big chunks keep
// the number of (device-gated, empty) launches small.  mofreak_reserve sizes the workspace for the same figure, so that
// calls after it do not allocate.
int slow_chunk(const mofreak_ctx *ctx, int W, int H)
{
    const size_t per_pair = (size_t)(H + 1) * integral_pitch(W) * sizeof(int32_t);
    const int chunk = (int)std::max<size_t>(1, ((size_t)1 << 30) / per_pair);
    return ctx->chunk_pairs_hint > 0 ? std::min(chunk, ctx->chunk_pairs_hint) : chunk;  // (mofreak_reserve's chunk_pairs: also here)
}

struct Geometry {
    int W, H;
    int64_t row_stride, pair_stride;
};

int validate_frames(const mofreak_ctx *ctx, const void *cur, const void *prev, int W, int H, int64_t row_stride,
                    int64_t pair_stride, int n_pairs)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_pairs < 0 || W <= 0 || H <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "W, H must be positive and n_pairs >= 0");
    if (n_pairs > 0 && (!cur || !prev)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null frame pointer");
    if (row_stride < W) return fail(ctx, MOFREAK_ERR_BAD_ARG, "row_stride < W");
    if (n_pairs > 1 && pair_stride == 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "pair_stride is 0");
    if ((size_t)kBandRows * integral_pitch(W) * sizeof(int32_t) > 160 * 1024)
        return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "frame wider than the banded integral kernel supports (W <= 10232)");
    if ((int64_t)(H + 1) * integral_pitch(W) >= ((int64_t)1 << 31))
        return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "frame too large for 32-bit integral indexing");
    // the gather path addresses a frame's bytes by 32-bit offsets built from 24-bit factors (row index x row_stride)
    if (row_stride >= ((int64_t)1 << 23) || (int64_t)H * row_stride >= ((int64_t)1 << 31))
        return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "row_stride >= 2^23 or a frame of 2 GiB and more");
    return MOFREAK_OK;
}

// integral images of pairs [p0, p0+np) into ctx->integral
int run_integral(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, const Geometry &g, int p0, int np,
                 const int32_t *gate = nullptr)
{
    const int pitch = integral_pitch(g.W);
    const int n_bands = (g.H + kBandRows * kBandGroup - 1) / (kBandRows * kBandGroup);
    int rc = ensure(ctx, ctx->integral, (size_t)np * (g.H + 1) * pitch * sizeof(int32_t));
    if (rc) return rc;
    rc = ensure(ctx, ctx->band_totals, (size_t)np * n_bands * pitch * sizeof(int32_t));
    if (rc) return rc;
    IntegralArgs a;
    a.gate = gate;
    a.f.cur = cur + (int64_t)p0 * g.pair_stride;
    a.f.prev = prev + (int64_t)p0 * g.pair_stride;
    a.f.W = g.W;
    a.f.H = g.H;
    a.f.row_stride = g.row_stride;
    a.f.pair_stride = g.pair_stride;
    if (ctx->use_det_diff && ctx->det_diff.base && ctx->det_diff.W == g.W && ctx->det_diff.H == g.H && p0 + np <= ctx->det_diff.pairs) {
        // the frame loop: the detector has left |cur - prev| of these very pairs in HBM: one byte per pixel to read instead of two
        a.f.cur = ctx->det_diff.base + (int64_t)p0 * ctx->det_diff.pair_stride;
        a.f.prev = nullptr;
        a.f.row_stride = g.W;
        a.f.pair_stride = ctx->det_diff.pair_stride;
    }
    a.integral = static_cast<int32_t *>(ctx->integral.ptr);
    a.band_totals = static_cast<int32_t *>(ctx->band_totals.ptr);
    a.pitch = pitch;
    a.n_bands = n_bands;
    a.n_pairs = np;
    const int e = launch_integral(a, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("integral launch: ") + hipGetErrorString((hipError_t)e));
    return MOFREAK_OK;
}

// Events for mofreak_set_profiling: [0] start, [1] after binning, [2] after the tile kernel, [3] after the gather path.
int span_begin(mofreak_ctx *ctx, mofreak_ctx::Span &span)
{
    for (auto &e : span.ev) {
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(ctx, hipEventCreate(&e));
        }
    }
    HIP_TRY(ctx, hipEventRecord(span.ev[0], ctx->stream));
    return MOFREAK_OK;
}

// Gather path ("v1"): global integral of a chunk of pairs + one wavefront per keypoint instance.
// With slow_mode the instance list is the binning pass's slow list and all kernels are gated on its count.
int run_gather(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, const Geometry &g, int n_pairs,
               const mofreak_keypoint *kps, const int64_t *d_offsets, const int64_t *h_offsets, int64_t n_kp,
               uint8_t *out_desc, uint8_t *out_valid, int32_t *out_info, uint8_t *out_roi19, bool slow_mode)
{
    int chunk = slow_mode ? slow_chunk(ctx, g.W, g.H) : choose_chunk(ctx, g.W, g.H, n_pairs);
    chunk = std::max(1, std::min(chunk, n_pairs));
    const int32_t *gate = slow_mode ? static_cast<const int32_t *>(ctx->slow_count.ptr) : nullptr;
    for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int np = std::min(chunk, n_pairs - p0);
        int64_t item_base, n_items;
        if (h_offsets) {
            item_base = h_offsets[p0];
            n_items = h_offsets[p0 + np] - h_offsets[p0];
        } else {
            item_base = (int64_t)p0 * n_kp;
            n_items = (int64_t)np * n_kp;
        }
        if (n_items == 0) continue;
        int rc = run_integral(ctx, cur, prev, g, p0, np, gate);
        if (rc) return rc;
        DescribeArgs a;
        a.f.cur = cur + (int64_t)p0 * g.pair_stride;
        a.f.prev = prev + (int64_t)p0 * g.pair_stride;
        a.f.W = g.W;
        a.f.H = g.H;
        a.f.row_stride = g.row_stride;
        a.f.pair_stride = g.pair_stride;
        a.integral = static_cast<const int32_t *>(ctx->integral.ptr);
        a.pitch = integral_pitch(g.W);
        a.n_pairs = np;
        a.lut = ctx->d_lut;
        a.resize = ctx->d_resize;
        a.small = ctx->d_small;
        a.theta = ctx->d_theta;
        a.kps = kps;
        a.kp_offsets = d_offsets;
        a.n_kp = n_kp;
        a.first_pair = p0;
        a.item_base = item_base;
        a.n_items = n_items;
        a.out_desc = out_desc;
        a.out_valid = out_valid;
        a.out_info = out_info;
        a.out_roi19 = out_roi19;
        a.status = ctx->d_status;
        a.slow_list = slow_mode ? static_cast<const int32_t *>(ctx->slow_list.ptr) : nullptr;
        a.slow_count = gate;
        a.band_start = slow_mode ? ctx->slow_band_start : nullptr;
        a.bands_per_pair = ctx->slow_bands_per_pair;
        a.n_pairs_total = n_pairs;
        const int64_t want = (n_items + 3) / 4;
        const int n_blocks = (int)std::min<int64_t>(want, (int64_t)ctx->n_cus * 5);  // 5 workgroups of 4 waves: what a CU holds at describe_kernel's register count
        const int e = launch_describe(a, n_blocks, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("describe launch: ") + hipGetErrorString((hipError_t)e));
    }
    return MOFREAK_OK;
}

// The device-pointer implementation behind mofreak_extract_pairs and the component entry points:
// bin the keypoints by image tile, describe them with the fused tile kernel, and leave what does not fit a tile's
// halo (large keypoints) to the gather path.
int extract_device(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, const Geometry &g, int n_pairs,
                   const mofreak_keypoint *kps, const int64_t *d_offsets, const int64_t *h_offsets, int64_t n_kp,
                   uint8_t *out_desc, uint8_t *out_valid, int32_t *out_info, uint8_t *out_roi19)
{
    if (n_pairs == 0 || n_kp == 0) return MOFREAK_OK;
    if (n_kp >= ((int64_t)1 << 31)) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "more than 2^31 keypoints in one call");
    const int64_t n_items = h_offsets ? n_kp : (int64_t)n_pairs * n_kp;
    mofreak_ctx::Span span{};
    int rc;
    if (ctx->profiling && (rc = span_begin(ctx, span))) return rc;
    span.pairs = n_pairs;
    span.items = n_items;

    const bool gather_only = ctx->path_mode == MOFREAK_PATH_GATHER || out_roi19 != nullptr;
    if (gather_only) {
        if (ctx->profiling) {
            HIP_TRY(ctx, hipEventRecord(span.ev[1], ctx->stream));
            HIP_TRY(ctx, hipEventRecord(span.ev[2], ctx->stream));
        }
        rc = run_gather(ctx, cur, prev, g, n_pairs, kps, d_offsets, h_offsets, n_kp, out_desc, out_valid, out_info, out_roi19, false);
        if (rc) return rc;
    } else {
        const int tiles_x = (g.W + kTileW - 1) / kTileW, tiles_y = (g.H + kTileH - 1) / kTileH;
        // A list per pair with few keypoints per tile (a detector's output: some thousand keypoints on a frame of 255 tiles)
        // would be handed to the gather path tile by tile in the binning pass: it goes there as a whole instead -- no tile
        // keys to count, scan and scatter, no tile kernel launch.  (A tile pays from about 48 keypoints up, tile_kernel.hip.)
        const bool all_gather = d_offsets != nullptr && n_kp < (int64_t)n_pairs * tiles_x * tiles_y * 32;
        const int64_t n_keys = all_gather ? 0 : (int64_t)tiles_x * tiles_y * (d_offsets ? n_pairs : 1);
        const int64_t n_bkeys = (int64_t)tiles_y * (d_offsets ? n_pairs : 1);  // the gather path's keypoints are binned too: by band of rows
        if (n_keys + n_bkeys >= ((int64_t)1 << 31)) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "too many (pair, tile) bins in one call");
        const size_t bin_blocks = ((size_t)n_kp + 255) / 256;
        // keys, then a byte of scale index per keypoint, then (4-byte aligned) pass 1's two figures per workgroup
        const size_t wg_at = ((size_t)n_kp * 5 + 3) & ~(size_t)3;
        if ((rc = ensure(ctx, ctx->kp_key, wg_at + 2 * bin_blocks * sizeof(int32_t) + 16))) return rc;
        if ((rc = ensure(ctx, ctx->sorted_idx, (size_t)n_kp * sizeof(SortedKp)))) return rc;
        if ((rc = ensure(ctx, ctx->slow_list, (size_t)n_kp * 4))) return rc;
        // every counter of the binning pass in one buffer (one fill clears them): the gather path's count and the largest
        // pattern size on a 64-byte line each, then per key the population / start, the scatter cursor and the smallest
        // (stored complemented, so that it too starts from zero) and largest ROI side
        const size_t key_pad = ((size_t)(n_keys + n_bkeys) + 1 + 15) & ~(size_t)15;
        if ((rc = ensure(ctx, ctx->slow_count, (kBinHeaderInts + 4 * key_pad) * sizeof(int32_t)))) return rc;
        int32_t *bin_base = static_cast<int32_t *>(ctx->slow_count.ptr);
        BinArgs b;
        b.kps = kps;
        b.kp_offsets = d_offsets;
        b.n_kp = n_kp;
        b.n_pairs = n_pairs;
        b.W = g.W;
        b.H = g.H;
        b.tiles_x = tiles_x;
        b.tiles_y = tiles_y;
        b.force_slow = all_gather ? 1 : 0;
        b.small = ctx->d_small;
        b.kp_key = static_cast<int32_t *>(ctx->kp_key.ptr);
        b.kp_scale = static_cast<uint8_t *>(ctx->kp_key.ptr) + (size_t)n_kp * 4;
        b.wg_slow = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(ctx->kp_key.ptr) + wg_at);
        b.wg_maxps = b.wg_slow + bin_blocks;
        b.tile_start = bin_base + kBinHeaderInts;
        b.tile_cursor = b.tile_start + key_pad;
        b.tile_lmin_c = reinterpret_cast<uint32_t *>(b.tile_cursor + key_pad);
        b.tile_lmax = b.tile_lmin_c + key_pad;
        b.counter_bytes = (kBinHeaderInts + 4 * key_pad) * sizeof(int32_t);
        b.sorted_kp = static_cast<SortedKp *>(ctx->sorted_idx.ptr);
        b.slow_list = static_cast<int32_t *>(ctx->slow_list.ptr);
        b.slow_count = bin_base;
        b.max_ps = bin_base + 16;  // its own 64-byte line
        b.out_desc = out_desc;
        b.out_valid = out_valid;
        b.out_info = out_info;
        b.n_keys = n_keys;
        b.n_bkeys = n_bkeys;
        ctx->slow_band_start = b.tile_start + n_keys;
        ctx->slow_bands_per_pair = tiles_y;
        int e = launch_bin(b, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("binning launch: ") + hipGetErrorString((hipError_t)e));
        if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(span.ev[1], ctx->stream));

        const int kMaxGridY = 32768;  // pairs per tile_kernel launch (keeps the 1-D grid well inside 2^28 work items)
        for (int p0 = 0; p0 < n_pairs && !all_gather; p0 += kMaxGridY) {
            const int np = std::min(kMaxGridY, n_pairs - p0);
            TileArgs t;
            t.f.cur = cur + (int64_t)p0 * g.pair_stride;
            t.f.prev = prev + (int64_t)p0 * g.pair_stride;
            t.f.W = g.W;
            t.f.H = g.H;
            t.f.row_stride = g.row_stride;
            t.f.pair_stride = g.pair_stride;
            t.n_pairs = np;
            t.tiles_x = tiles_x;
            t.tiles_y = tiles_y;
            t.lut = ctx->d_lut;
            t.lut_int = ctx->d_lut_int;
            // coordinates below 2^11 / 2^12 / 2^13 (pattern reach included): two float roundings of at most 2^-14 /
            // 2^-13 / 2^-12 each; twice their sum as the margin
            {
                const int reach = std::max(g.W, g.H) + kTileHalo;
                t.box_margin = reach < 2048 ? 1.0f / 4096 : reach < 4096 ? 1.0f / 2048 : reach < 8192 ? 1.0f / 1024 : 2.0f;
            }
            t.small = ctx->d_small;
            t.theta = ctx->d_theta;
            t.mip_samples = ctx->d_mip_samples;
            t.mip_pos = ctx->d_mip_pos;
            t.lanes = ctx->d_tile_lanes;
            t.mip_n_cur = ctx->tables.mip_n_cur;
            t.mip_n = ctx->tables.mip_n;
            t.mip_stride = ctx->tables.mip_stride;
            t.kp_offsets = d_offsets;
            t.n_kp = n_kp;
            // CSR: bins are per (pair, tile); shared list: per tile, outputs offset by the pair
            t.tile_start = b.tile_start + (d_offsets ? (int64_t)p0 * tiles_x * tiles_y : 0);
            t.sorted_kp = static_cast<const SortedKp *>(ctx->sorted_idx.ptr);
            t.max_ps = b.max_ps;
            t.status = ctx->d_status;
            t.out_items = d_offsets ? n_kp : (int64_t)np * n_kp;
            t.tile_lmin_c = b.tile_lmin_c + (d_offsets ? (int64_t)p0 * tiles_x * tiles_y : 0);
            t.tile_lmax = b.tile_lmax + (d_offsets ? (int64_t)p0 * tiles_x * tiles_y : 0);
            t.out_desc = d_offsets ? out_desc : out_desc + (int64_t)p0 * n_kp * 16;
            t.out_valid = d_offsets ? out_valid : out_valid + (int64_t)p0 * n_kp;
            t.stamps = ctx->d_stamps;
            t.out_info = out_info ? (d_offsets ? out_info : out_info + (int64_t)p0 * n_kp * 4) : nullptr;
            e = launch_tile(t, ctx->stream);
            if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("tile kernel launch: ") + hipGetErrorString((hipError_t)e));
        }
        if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(span.ev[2], ctx->stream));
        rc = run_gather(ctx, cur, prev, g, n_pairs, kps, d_offsets, h_offsets, n_kp, out_desc, out_valid, out_info, nullptr, true);
        if (rc) return rc;
    }
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventRecord(span.ev[3], ctx->stream));
        ctx->spans.push_back(span);
    }
    return MOFREAK_OK;
}

// Host copy of CSR offsets (needed to cut the call into chunks); validates monotonicity.
int fetch_offsets(mofreak_ctx *ctx, const int64_t *offsets, int n_pairs, bool host_ptr, std::vector<int64_t> &h)
{
    h.resize((size_t)n_pairs + 1);
    if (host_ptr) {
        std::memcpy(h.data(), offsets, h.size() * sizeof(int64_t));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), offsets, h.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (h[0] != 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "kp_offsets[0] must be 0");
    for (int p = 0; p < n_pairs; ++p)
        if (h[p + 1] < h[p]) return fail(ctx, MOFREAK_ERR_BAD_ARG, "kp_offsets must be non-decreasing");
    return MOFREAK_OK;
}

int upload(mofreak_ctx *ctx, DeviceBuffer &b, const void *src, size_t bytes)
{
    int rc = ensure(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MOFREAK_OK;
}

// Bytes spanned by n_pairs frames with the given strides.
size_t frame_span(const Geometry &g, int n_pairs)
{
    if (n_pairs <= 0) return 0;
    return (size_t)((int64_t)(n_pairs - 1) * g.pair_stride + (int64_t)(g.H - 1) * g.row_stride + g.W);
}

int compact_device(mofreak_ctx *ctx, const mofreak_keypoint *kps, const int64_t *d_offsets, int64_t n_kp, int n_pairs,
                   int64_t n_items, int first_frame, const uint8_t *desc, const uint8_t *valid, mofreak_row *rows,
                   int64_t capacity, int64_t *n_rows_out)
{
    const int64_t nb64 = (n_items + kCompactItemsPerBlock - 1) / kCompactItemsPerBlock;
    if (nb64 > 0x7fffffff) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "too many items for one compaction");
    const int n_blocks = (int)nb64;
    int rc = ensure(ctx, ctx->compact_offsets, ((size_t)n_blocks + 1) * sizeof(int64_t));
    if (rc) return rc;
    CompactArgs a;
    a.kps = kps;
    a.kp_offsets = d_offsets;
    a.n_kp = n_kp;
    a.n_items = n_items;
    a.n_pairs = n_pairs;
    a.first_frame_number = first_frame;
    a.desc = desc;
    a.valid = valid;
    a.rows = rows;
    a.capacity = capacity;
    a.block_offsets = static_cast<int64_t *>(ctx->compact_offsets.ptr);
    a.n_blocks = n_blocks;
    const int e = launch_compact(a, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("compact launch: ") + hipGetErrorString((hipError_t)e));
    int64_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, a.block_offsets + n_blocks, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_rows_out) *n_rows_out = total;
    if (total > capacity) return fail(ctx, MOFREAK_ERR_CAPACITY, "rows_out too small: need " + std::to_string(total));
    return MOFREAK_OK;
}

}  // namespace

extern "C" {

int mofreak_abi_version(void) { return MOFREAK_ABI_VERSION; }

int mofreak_build_flags(void)
{
#ifdef MOFREAK_DEBUG_BOUNDS
    return 1;
#else
    return 0;
#endif
}

int mofreak_default_params(mofreak_params *p)
{
    if (!p) return MOFREAK_ERR_BAD_ARG;
    p->struct_size = (int32_t)sizeof(mofreak_params);
    p->gap_for_frame_difference = 5;
    p->mip_theta = 288;
    p->freak_pattern_scale = 22.0f;
    p->freak_n_octaves = 4;
    p->freak_orientation_normalized = 1;
    p->freak_scale_normalized = 1;
    p->freak_bit_mode = MOFREAK_BITS_SSE;
    p->brisk_fp_model = MOFREAK_FP_X87;
    return MOFREAK_OK;
}

int mofreak_create(int device_id, const mofreak_params *params, mofreak_ctx **out)
{
    if (!out) return fail(nullptr, MOFREAK_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    mofreak_params p;
    mofreak_default_params(&p);
    if (params) {
        if (params->struct_size != (int32_t)sizeof(mofreak_params))
            return fail(nullptr, MOFREAK_ERR_BAD_ARG, "params->struct_size mismatch (use mofreak_default_params)");
        p = *params;
    }
    if (p.gap_for_frame_difference < 1 || p.freak_n_octaves < 1 || !(p.freak_pattern_scale > 0) ||
        p.freak_bit_mode < 0 || p.freak_bit_mode > 2 || p.mip_theta < 0 || p.brisk_fp_model < 0 || p.brisk_fp_model > 1)
        return fail(nullptr, MOFREAK_ERR_BAD_ARG, "parameter out of range");

    int n_dev = 0;
    if (device_id != MOFREAK_TABLES_ONLY) {
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
            return fail(nullptr, MOFREAK_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
        if (device_id < 0 || device_id >= n_dev) return fail(nullptr, MOFREAK_ERR_BAD_ARG, "device_id out of range");
    }

    mofreak_ctx *ctx = new (std::nothrow) mofreak_ctx;
    if (!ctx) return fail(nullptr, MOFREAK_ERR_OOM, "host allocation failed");
    ctx->device = device_id;
    ctx->params = p;
    try {
        FreakParams fp;
        fp.pattern_scale = p.freak_pattern_scale;
        fp.n_octaves = p.freak_n_octaves;
        fp.orientation_normalized = p.freak_orientation_normalized != 0;
        fp.scale_normalized = p.freak_scale_normalized != 0;
        fp.bit_mode = p.freak_bit_mode;
        build_tables(fp, ctx->tables);
    } catch (const std::exception &e) {
        const std::string m = std::string("table construction failed: ") + e.what();
        delete ctx;
        return fail(nullptr, MOFREAK_ERR_OOM, m);
    }
    if (ctx->tables.min_sigma < 0.5f) {
        delete ctx;
        return fail(nullptr, MOFREAK_ERR_UNSUPPORTED,
                    "freak_pattern_scale gives a pattern sigma < 0.5: the bilinear branch of FREAK::meanIntensity is not implemented");
    }

    if (device_id == MOFREAK_TABLES_ONLY) {  // host tables only: every compute entry point refuses this context
        *out = ctx;
        return MOFREAK_OK;
    }

#define CREATE_TRY(expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            const std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_); \
            mofreak_destroy(ctx);                                                     \
            return fail(nullptr, MOFREAK_ERR_HIP, m_);                                \
        }                                                                             \
    } while (0)

    CREATE_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device_id));
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;

    const Tables &t = ctx->tables;
    CREATE_TRY(hipMalloc((void **)&ctx->d_lut, t.lut.size() * sizeof(PatternPoint)));
    CREATE_TRY(hipMalloc((void **)&ctx->d_lut_int, t.lut_int.size() * sizeof(BoxInt)));
    CREATE_TRY(hipMemcpy(ctx->d_lut_int, t.lut_int.data(), t.lut_int.size() * sizeof(BoxInt), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc((void **)&ctx->d_resize, t.resize.size() * sizeof(ResizeTap)));
    CREATE_TRY(hipMalloc((void **)&ctx->d_small, sizeof(SmallTables)));
    CREATE_TRY(hipMalloc((void **)&ctx->d_status, 2 * sizeof(int32_t)));  // [0] describe kernels (read by mofreak_check_status), [1] detector
    if (const char *ev = std::getenv("MOFREAK_TILE_STAMPS"); ev && ev[0] == '1') {
        CREATE_TRY(hipMalloc((void **)&ctx->d_stamps, kTileStampSlots * sizeof(unsigned long long)));
        CREATE_TRY(hipMemsetAsync(ctx->d_stamps, 0, kTileStampSlots * sizeof(unsigned long long), ctx->stream));  // (on the stream the kernels run on: it does not wait for the null stream)
    }
    CREATE_TRY(hipMalloc((void **)&ctx->d_theta, t.theta_bounds.size() * sizeof(ThetaBound)));
    CREATE_TRY(hipMemcpy(ctx->d_theta, t.theta_bounds.data(), t.theta_bounds.size() * sizeof(ThetaBound), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc((void **)&ctx->d_mip_samples, t.mip_samples.size() * sizeof(MipSample)));
    CREATE_TRY(hipMalloc((void **)&ctx->d_mip_pos, t.mip_pos.size() * sizeof(uint16_t)));
    CREATE_TRY(hipMemcpy(ctx->d_mip_samples, t.mip_samples.data(), t.mip_samples.size() * sizeof(MipSample), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(ctx->d_mip_pos, t.mip_pos.data(), t.mip_pos.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    {
        // the tile kernel's per-lane constants: task order of a group's 4 x 43 box means (the two outer rings of all
        // four keypoints first: those boxes may need slices), the lane's descriptor pair and orientation pairs
        TileLane lanes[64];
        std::memset(lanes, 0, sizeof(lanes));
        const int big = 12, group = 4;
        for (int lane = 0; lane < 64; ++lane) {
            for (int u = 0; u < 3; ++u) {
                const int tk = lane + 64 * u;
                int kq = 0, pt = kNbPoints - 1;  // past the group's 172 tasks: keypoint 0's last point once more
                if (tk < group * big) {
                    kq = tk / big;
                    pt = tk % big;
                } else if (tk < group * kNbPoints) {
                    kq = (tk - group * big) / (kNbPoints - big);
                    pt = big + (tk - group * big) % (kNbPoints - big);
                }
                lanes[lane].task[u] = static_cast<uint16_t>(kq | pt << 8);
            }
            lanes[lane].pair_i = t.bit_pair_i[lane];
            lanes[lane].pair_j = t.bit_pair_j[lane];
            for (int k = 0; k < 3; ++k) {
                const int m = (lane & 15) + 16 * k;
                if (m < kNbOrientPairs) {
                    lanes[lane].opi[k] = static_cast<uint8_t>(t.orient[m].i);
                    lanes[lane].opj[k] = static_cast<uint8_t>(t.orient[m].j);
                    lanes[lane].owx[k] = static_cast<float>(t.orient[m].weight_dx) * (1.0f / 2048.0f);
                    lanes[lane].owy[k] = static_cast<float>(t.orient[m].weight_dy) * (1.0f / 2048.0f);
                }
            }
        }
        CREATE_TRY(hipMalloc((void **)&ctx->d_tile_lanes, sizeof(lanes)));
        CREATE_TRY(hipMemcpy(ctx->d_tile_lanes, lanes, sizeof(lanes), hipMemcpyHostToDevice));
    }
    SmallTables st;
    std::memset(&st, 0, sizeof(st));
    std::memcpy(st.pattern_sizes, t.pattern_sizes, sizeof(st.pattern_sizes));
    std::memcpy(st.scale_thresholds, t.scale_thresholds, sizeof(st.scale_thresholds));
    std::memcpy(st.orient, t.orient, sizeof(st.orient));
    std::memcpy(st.bit_pair_i, t.bit_pair_i, 64);
    std::memcpy(st.bit_pair_j, t.bit_pair_j, 64);
    for (int k = 0; k < kNbOrientPairs; ++k) st.need_orient |= 1ull << t.orient[k].i | 1ull << t.orient[k].j;
    for (int b = 0; b < 64; ++b) st.need_bits |= 1ull << t.bit_pair_i[b] | 1ull << t.bit_pair_j[b];
    st.fixed_scale_index = t.fixed_scale_index;
    st.orientation_normalized = p.freak_orientation_normalized != 0;
    st.scale_normalized = p.freak_scale_normalized != 0;
    st.bit_mode = p.freak_bit_mode;
    st.mip_theta = p.mip_theta;
    if (t.mip_need_cur.size() > 64 || t.mip_need_prev.size() > 256) {
        mofreak_destroy(ctx);
        return fail(nullptr, MOFREAK_ERR_UNSUPPORTED, "MIP position lists larger than the device tables");
    }
    st.mip_n_cur = (int32_t)t.mip_need_cur.size();
    st.mip_n_prev = (int32_t)t.mip_need_prev.size();
    for (size_t i = 0; i < 64; ++i) st.mip_cur[i] = t.mip_need_cur[std::min(i, t.mip_need_cur.size() - 1)];
    for (size_t i = 0; i < 256; ++i) st.mip_prev[i] = t.mip_need_prev[std::min(i, t.mip_need_prev.size() - 1)];
    CREATE_TRY(hipMemcpy(ctx->d_lut, t.lut.data(), t.lut.size() * sizeof(PatternPoint), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(ctx->d_resize, t.resize.data(), t.resize.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(ctx->d_small, &st, sizeof(st), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemsetAsync(ctx->d_status, 0, 2 * sizeof(int32_t), ctx->stream));
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
#undef CREATE_TRY
    *out = ctx;
    return MOFREAK_OK;
}

void mofreak_destroy(mofreak_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device < 0) {
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    if (ctx->d_resize) (void)hipFree(ctx->d_resize);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_theta) (void)hipFree(ctx->d_theta);
    if (ctx->d_stamps) (void)hipFree(ctx->d_stamps);
    if (ctx->d_mip_samples) (void)hipFree(ctx->d_mip_samples);
    if (ctx->d_mip_pos) (void)hipFree(ctx->d_mip_pos);
    if (ctx->d_tile_lanes) (void)hipFree(ctx->d_tile_lanes);
    if (ctx->d_lut_int) (void)hipFree(ctx->d_lut_int);
    for (int b = 0; b < 2; ++b) {
        release(ctx->pipe.d_frames[b]);
        release(ctx->pipe.d_rows[b]);
        release(ctx->pipe.d_pair_rows[b]);
    }
    release(ctx->pair_label);
    if (ctx->pipe.ready) {
        (void)hipStreamSynchronize(ctx->pipe.s_in);
        (void)hipStreamSynchronize(ctx->pipe.s_out);
        for (int b = 0; b < 2; ++b) {
            (void)hipEventDestroy(ctx->pipe.ev_in[b]);
            (void)hipEventDestroy(ctx->pipe.ev_comp[b]);
            (void)hipEventDestroy(ctx->pipe.ev_out[b]);
            (void)hipHostFree(ctx->pipe.h_count[b]);
            if (ctx->pipe.h_frames[b]) (void)hipHostFree(ctx->pipe.h_frames[b]);
            if (ctx->pipe.h_rows[b]) (void)hipHostFree(ctx->pipe.h_rows[b]);
            if (ctx->pipe.h_pair_rows[b]) (void)hipHostFree(ctx->pipe.h_pair_rows[b]);
        }
        (void)hipStreamDestroy(ctx->pipe.s_in);
        (void)hipStreamDestroy(ctx->pipe.s_out);
    }
    release(ctx->kp_key);
    release(ctx->sorted_idx);
    release(ctx->slow_list);
    release(ctx->slow_count);
    release(ctx->bow_counts);
    release(ctx->bow_expanded);
    release(ctx->fmt_ws);
    for (DeviceBuffer *b : {&ctx->det_img, &ctx->det_score, &ctx->det_touch, &ctx->det_status, &ctx->det_rows, &ctx->det_cand_xy, &ctx->det_cand_flag,
                            &ctx->det_cand_emit, &ctx->det_cand_spec, &ctx->det_cand_asked, &ctx->det_cand_win, &ctx->det_cand_res, &ctx->det_layer_start, &ctx->det_emit_count, &ctx->det_emit_chunks, &ctx->det_geom, &ctx->det_emit_offsets,
                            &ctx->det_out_kps, &ctx->det_out_offsets, &ctx->det_out_resp, &ctx->det_out_layer, &ctx->det_planes_out, &ctx->det_hit_mask, &ctx->det_walk_list, &ctx->det_walk_count, &ctx->det_cand_cells, &ctx->det_tie_list, &ctx->det_tie_count})
        release(*b);
    release(ctx->integral);
    release(ctx->band_totals);
    release(ctx->scratch_desc);
    release(ctx->scratch_valid);
    release(ctx->compact_offsets);
    release(ctx->offsets_dev);
    for (auto &s : ctx->stage) release(s);
    for (auto &sp : ctx->spans)
        for (auto e : sp.ev) (void)hipEventDestroy(e);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mofreak_last_error(const mofreak_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mofreak_set_stream(mofreak_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    if (hip_stream == nullptr) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    } else {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
    }
    return MOFREAK_OK;
}

int mofreak_synchronize(mofreak_ctx *ctx)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOFREAK_OK;
}

int mofreak_reserve(mofreak_ctx *ctx, int W, int H, int chunk_pairs)
{
    if (!ctx || W <= 0 || H <= 0) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    ctx->chunk_pairs_hint = chunk_pairs > 0 ? chunk_pairs : 0;
    // the larger of the two users: the gather path alone (chunk_pairs at a time) and the gather path behind the tile kernel
    const int chunk = std::max(choose_chunk(ctx, W, H, 1 << 30), ctx->path_mode == MOFREAK_PATH_GATHER ? 1 : slow_chunk(ctx, W, H));
    const int pitch = integral_pitch(W);
    const int n_bands = (H + kBandRows * kBandGroup - 1) / (kBandRows * kBandGroup);
    int rc = ensure(ctx, ctx->integral, (size_t)chunk * (H + 1) * pitch * sizeof(int32_t));
    if (rc) return rc;
    return ensure(ctx, ctx->band_totals, (size_t)chunk * n_bands * pitch * sizeof(int32_t));
}

int mofreak_set_path(mofreak_ctx *ctx, int path)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (path != MOFREAK_PATH_AUTO && path != MOFREAK_PATH_GATHER) return fail(ctx, MOFREAK_ERR_BAD_ARG, "unknown path");
    ctx->path_mode = path;
    return MOFREAK_OK;
}

int mofreak_get_tile_stamps(mofreak_ctx *ctx, uint64_t *out, int n, int reset)
{
    if (!ctx || !out || n < 0) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    if (!ctx->d_stamps) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "create the context with MOFREAK_TILE_STAMPS=1 in the environment");
    n = std::min(n, kTileStampSlots);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_stamps, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(ctx->d_stamps, 0, kTileStampSlots * sizeof(unsigned long long), ctx->stream));
    return MOFREAK_OK;
}

int mofreak_set_profiling(mofreak_ctx *ctx, int enable)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    ctx->profiling = enable != 0;
    return MOFREAK_OK;
}

int mofreak_get_profile(mofreak_ctx *ctx, mofreak_profile *out, int reset)
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &sp : ctx->spans) {
        float a = 0, b = 0, c = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&a, sp.ev[0], sp.ev[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, sp.ev[1], sp.ev[2]));
        HIP_TRY(ctx, hipEventElapsedTime(&c, sp.ev[2], sp.ev[3]));
        ctx->prof.bin_ms += a;
        ctx->prof.tile_ms += b;
        ctx->prof.gather_ms += c;
        ctx->prof.calls += 1;
        ctx->prof.pairs += sp.pairs;
        ctx->prof.descriptors += sp.items;
        for (auto e : sp.ev) ctx->event_pool.push_back(e);
    }
    ctx->spans.clear();
    *out = ctx->prof;
    if (reset) ctx->prof = mofreak_profile{};
    return MOFREAK_OK;
}

int mofreak_check_status(mofreak_ctx *ctx)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    int32_t s = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&s, ctx->d_status, sizeof(s), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(s), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (s & 1) return fail(ctx, MOFREAK_ERR_ROI, "a keypoint's MIP ROI left the image (the reference throws there); it was marked invalid");
    if (s & 2) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "a keypoint's ROI side exceeds the resize tables; it was marked invalid");
    if (s & 64) return fail(ctx, MOFREAK_ERR_HIP, "debug build: a tile-kernel access left its bounds (LDS allocation or descriptor output)");
    return MOFREAK_OK;
}

int mofreak_extract_pairs(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, int64_t row_stride,
                          int64_t pair_stride, int n_pairs, const mofreak_keypoint *kps, const int64_t *kp_offsets,
                          int64_t n_kp, uint8_t *out_desc16, uint8_t *out_valid, unsigned flags)
{
    int rc = validate_frames(ctx, cur, prev, W, H, row_stride, pair_stride, n_pairs);
    if (rc) return rc;
    if (n_kp < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n_kp < 0");
    if (n_kp > 0 && (!kps || !out_desc16 || !out_valid)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null keypoint/output pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry g{W, H, row_stride, pair_stride};
    std::vector<int64_t> h_off;
    const int64_t *d_off = nullptr;
    int64_t n_out = (int64_t)n_pairs * n_kp;
    if (kp_offsets) {
        rc = fetch_offsets(ctx, kp_offsets, n_pairs, host, h_off);
        if (rc) return rc;
        if (h_off[n_pairs] != n_kp) return fail(ctx, MOFREAK_ERR_BAD_ARG, "kp_offsets[n_pairs] != n_kp");
        n_out = n_kp;
        d_off = kp_offsets;
    }
    if (!host)
        return extract_device(ctx, cur, prev, g, n_pairs, kps, d_off, kp_offsets ? h_off.data() : nullptr, n_kp,
                              out_desc16, out_valid, nullptr, nullptr);

    // host pointers: stage through device buffers
    const size_t span = frame_span(g, n_pairs);
    if ((rc = upload(ctx, ctx->stage[0], cur, span))) return rc;
    if ((rc = upload(ctx, ctx->stage[1], prev, span))) return rc;
    if ((rc = upload(ctx, ctx->stage[2], kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
    if (kp_offsets) {
        if ((rc = upload(ctx, ctx->offsets_dev, h_off.data(), h_off.size() * sizeof(int64_t)))) return rc;
        d_off = static_cast<const int64_t *>(ctx->offsets_dev.ptr);
    }
    if ((rc = ensure(ctx, ctx->stage[3], (size_t)n_out * 16))) return rc;
    if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_out))) return rc;
    rc = extract_device(ctx, static_cast<const uint8_t *>(ctx->stage[0].ptr), static_cast<const uint8_t *>(ctx->stage[1].ptr),
                        g, n_pairs, static_cast<const mofreak_keypoint *>(ctx->stage[2].ptr), d_off,
                        kp_offsets ? h_off.data() : nullptr, n_kp, static_cast<uint8_t *>(ctx->stage[3].ptr),
                        static_cast<uint8_t *>(ctx->stage[4].ptr), nullptr, nullptr);
    if (rc) return rc;
    if (n_out) {
        HIP_TRY(ctx, hipMemcpyAsync(out_desc16, ctx->stage[3].ptr, (size_t)n_out * 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(out_valid, ctx->stage[4].ptr, (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOFREAK_OK;
}

int mofreak_compact_rows(mofreak_ctx *ctx, const mofreak_keypoint *kps, const int64_t *kp_offsets, int64_t n_kp,
                         int n_pairs, int first_frame_number, const uint8_t *desc16, const uint8_t *valid,
                         mofreak_row *rows_out, int64_t rows_capacity, int64_t *n_rows_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_pairs < 0 || n_kp < 0 || rows_capacity < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const int64_t n_items = kp_offsets ? n_kp : (int64_t)n_pairs * n_kp;
    if (n_rows_out) *n_rows_out = 0;
    if (n_items == 0) return MOFREAK_OK;
    if (!kps || !desc16 || !valid || (!rows_out && rows_capacity > 0)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    if (!host)
        return compact_device(ctx, kps, kp_offsets, n_kp, n_pairs, n_items, first_frame_number, desc16, valid, rows_out,
                              rows_capacity, n_rows_out);
    int rc;
    if ((rc = upload(ctx, ctx->stage[2], kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
    if ((rc = upload(ctx, ctx->stage[3], desc16, (size_t)n_items * 16))) return rc;
    if ((rc = upload(ctx, ctx->stage[4], valid, (size_t)n_items))) return rc;
    const int64_t *d_off = nullptr;
    if (kp_offsets) {
        if ((rc = upload(ctx, ctx->offsets_dev, kp_offsets, ((size_t)n_pairs + 1) * sizeof(int64_t)))) return rc;
        d_off = static_cast<const int64_t *>(ctx->offsets_dev.ptr);
    }
    if ((rc = ensure(ctx, ctx->stage[5], (size_t)std::max<int64_t>(rows_capacity, 1) * sizeof(mofreak_row)))) return rc;
    int64_t total = 0;
    rc = compact_device(ctx, static_cast<const mofreak_keypoint *>(ctx->stage[2].ptr), d_off, n_kp, n_pairs, n_items,
                        first_frame_number, static_cast<const uint8_t *>(ctx->stage[3].ptr),
                        static_cast<const uint8_t *>(ctx->stage[4].ptr), static_cast<mofreak_row *>(ctx->stage[5].ptr),
                        rows_capacity, &total);
    if (n_rows_out) *n_rows_out = total;
    if (rc) return rc;
    if (total) HIP_TRY(ctx, hipMemcpy(rows_out, ctx->stage[5].ptr, (size_t)total * sizeof(mofreak_row), hipMemcpyDeviceToHost));
    return MOFREAK_OK;
}

int mofreak_extract_stream(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H, const mofreak_keypoint *kps,
                           const int64_t *kp_offsets, int64_t n_kp, mofreak_row *rows_out, int64_t rows_capacity,
                           int64_t *n_rows_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows_out) *n_rows_out = 0;
    if (T < 0 || n_kp < 0 || rows_capacity < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count");
    const int gap = ctx->params.gap_for_frame_difference;
    const int n_pairs = T - gap;  // the first `gap` frames only prime the queue (MoFREAKUtilities.cpp:391-399)
    if (n_pairs <= 0 || n_kp == 0) return MOFREAK_OK;
    const int64_t fsz = (int64_t)W * H;
    int rc = validate_frames(ctx, frames, frames, W, H, W, fsz, n_pairs);
    if (rc) return rc;
    if (!kps) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null keypoint pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry g{W, H, W, fsz};

    std::vector<int64_t> h_off;
    const int64_t *d_off = nullptr;
    int64_t n_items = (int64_t)n_pairs * n_kp;
    if (kp_offsets) {
        if ((rc = fetch_offsets(ctx, kp_offsets, n_pairs, host, h_off))) return rc;
        if (h_off[n_pairs] != n_kp) return fail(ctx, MOFREAK_ERR_BAD_ARG, "kp_offsets[T-gap] != n_kp");
        n_items = n_kp;
        d_off = kp_offsets;
    }
    const uint8_t *d_frames = frames;
    const mofreak_keypoint *d_kps = kps;
    mofreak_row *d_rows = rows_out;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], frames, (size_t)T * fsz))) return rc;
        if ((rc = upload(ctx, ctx->stage[2], kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
        if (kp_offsets) {
            if ((rc = upload(ctx, ctx->offsets_dev, h_off.data(), h_off.size() * sizeof(int64_t)))) return rc;
            d_off = static_cast<const int64_t *>(ctx->offsets_dev.ptr);
        }
        if ((rc = ensure(ctx, ctx->stage[5], (size_t)std::max<int64_t>(rows_capacity, 1) * sizeof(mofreak_row)))) return rc;
        d_frames = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        d_kps = static_cast<const mofreak_keypoint *>(ctx->stage[2].ptr);
        d_rows = static_cast<mofreak_row *>(ctx->stage[5].ptr);
    }
    if ((rc = ensure(ctx, ctx->scratch_desc, (size_t)n_items * 16))) return rc;
    if ((rc = ensure(ctx, ctx->scratch_valid, (size_t)n_items))) return rc;
    uint8_t *desc = static_cast<uint8_t *>(ctx->scratch_desc.ptr);
    uint8_t *valid = static_cast<uint8_t *>(ctx->scratch_valid.ptr);
    // pair p: current = frame p+gap, previous = frame p (:398-399, :485-487)
    rc = extract_device(ctx, d_frames + (int64_t)gap * fsz, d_frames, g, n_pairs, d_kps, d_off,
                        kp_offsets ? h_off.data() : nullptr, n_kp, desc, valid, nullptr, nullptr);
    if (rc) return rc;
    int64_t total = 0;
    // the first processed frame is labelled gap-1 (:401) and the label is incremented per frame (:488)
    rc = compact_device(ctx, d_kps, d_off, n_kp, n_pairs, n_items, gap - 1, desc, valid, d_rows, rows_capacity, &total);
    if (n_rows_out) *n_rows_out = total;
    if (rc) return rc;
    if (host && total)
        HIP_TRY(ctx, hipMemcpy(rows_out, d_rows, (size_t)total * sizeof(mofreak_row), hipMemcpyDeviceToHost));
    return MOFREAK_OK;
}

static bool is_device_memory(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'd pointer: not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

int mofreak_format_rows_device(mofreak_ctx *ctx, const mofreak_row *d_rows, int64_t n_rows, char *text, size_t cap, size_t *needed,
                               const int64_t *row_starts, int n_segments, size_t *segment_offsets_out)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows < 0 || (n_rows > 0 && !d_rows) || n_segments < 0 || (n_segments > 0 && (!row_starts || !segment_offsets_out)))
        return fail(ctx, MOFREAK_ERR_BAD_ARG, "mofreak_format_rows_device: bad argument");
    NEED_DEVICE(ctx);
    for (int i = 1; i < n_segments; ++i)
        if (row_starts[i] < row_starts[i - 1]) return fail(ctx, MOFREAK_ERR_BAD_ARG, "mofreak_format_rows_device: row_starts must ascend");
    const size_t ws_bytes = format_workspace_bytes(n_rows, n_segments), tail = (size_t)std::max(n_segments, 1) * sizeof(int64_t) + 64;
    int rc = ensure(ctx, ctx->fmt_ws, ws_bytes + tail);
    if (rc) return rc;
    uint8_t *ws = static_cast<uint8_t *>(ctx->fmt_ws.ptr);
    int64_t *d_starts = reinterpret_cast<int64_t *>(ws + ws_bytes);
    int32_t *d_bits = reinterpret_cast<int32_t *>(ws + ws_bytes + (size_t)std::max(n_segments, 1) * sizeof(int64_t));
    HIP_TRY(ctx, hipMemsetAsync(d_bits, 0, sizeof(int32_t), ctx->stream));
    if (n_segments > 0) HIP_TRY(ctx, hipMemcpyAsync(d_starts, row_starts, (size_t)n_segments * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    uint64_t *d_total = nullptr, *d_seg = nullptr;
    int e = launch_format_measure(d_rows, n_rows, ws, d_starts, n_segments, d_bits, &d_total, &d_seg, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("format (measure) launch: ") + hipGetErrorString((hipError_t)e));
    uint64_t total = 0;
    int32_t bits = 0;
    std::vector<uint64_t> seg((size_t)std::max(n_segments, 1));
    HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&bits, d_bits, sizeof bits, hipMemcpyDeviceToHost, ctx->stream));
    if (n_segments > 0) HIP_TRY(ctx, hipMemcpyAsync(seg.data(), d_seg, (size_t)n_segments * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bits)
        return fail(ctx, MOFREAK_ERR_UNSUPPORTED,
                    "mofreak_format_rows_device: a row holds a float outside [1e-4, 1e6) (or negative / not finite): format these rows with mofreak_format_rows");
    if (needed) *needed = (size_t)total;
    for (int i = 0; i < n_segments; ++i) segment_offsets_out[i] = (size_t)seg[(size_t)i];
    if (n_segments > 0) segment_offsets_out[n_segments] = (size_t)total;
    if (!text) return MOFREAK_OK;
    if (total > cap) return fail(ctx, MOFREAK_ERR_CAPACITY, "mofreak_format_rows_device: the text needs " + std::to_string(total) + " bytes");
    e = launch_format_write(d_rows, n_rows, ws, text, (uint64_t)cap, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("format (write) launch: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOFREAK_OK;
}

int mofreak_device_alloc(mofreak_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    *out = nullptr;
    HIP_TRY(ctx, hipMalloc(out, std::max<size_t>(bytes, 256)));
    return MOFREAK_OK;
}

int mofreak_device_free(mofreak_ctx *ctx, void *ptr)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    if (ptr) HIP_TRY(ctx, hipFree(ptr));
    return MOFREAK_OK;
}

int mofreak_copy_to_host(mofreak_ctx *ctx, void *host_dst, const void *device_src, size_t bytes)
{
    if (!ctx || (bytes > 0 && (!host_dst || !device_src))) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    if (bytes) HIP_TRY(ctx, hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
    return MOFREAK_OK;
}

int mofreak_host_alloc(mofreak_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    NEED_DEVICE(ctx);
    *out = nullptr;
    HIP_TRY(ctx, hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    return MOFREAK_OK;
}

int mofreak_host_free(mofreak_ctx *ctx, void *ptr)
{
    // ctx may be NULL: page-locked memory is not tied to the context that allocated it and may outlive it
    if (ctx) NEED_DEVICE(ctx);
    if (ptr) HIP_TRY(ctx, hipHostFree(ptr));
    return MOFREAK_OK;
}

namespace {

// Every exit of the pipelined frame loop after its first asynchronous operation goes through this: a copy may still be
// reading the caller's page-locked frames or writing the caller's rows, so the three streams are drained before the
// caller gets control (and possibly frees or reuses those buffers) -- on errors as on success.
struct PipeDrain {
    mofreak_ctx *ctx;
    bool armed = false;
    ~PipeDrain()
    {
        if (!armed) return;
        (void)hipStreamSynchronize(ctx->pipe.s_in);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->pipe.s_out);
    }
};

// Copy streams, events and the page-locked count words of the pipeline: created into locals and committed as a whole, so
// that a failure half way leaves nothing behind and nothing half-initialised.
int pipe_setup(mofreak_ctx *ctx)
{
    mofreak_ctx::Pipe &P = ctx->pipe;
    if (P.ready) return MOFREAK_OK;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev[6] = {};
    int64_t *cnt[2] = {};
    hipError_t e = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking);
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipHostMalloc((void **)&cnt[b], sizeof(int64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        for (auto c : cnt)
            if (c) (void)hipHostFree(c);
        for (auto v : ev)
            if (v) (void)hipEventDestroy(v);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        return fail(ctx, e == hipErrorOutOfMemory ? MOFREAK_ERR_OOM : MOFREAK_ERR_HIP, std::string("pipeline setup: ") + hipGetErrorString(e));
    }
    P.s_in = s_in;
    P.s_out = s_out;
    for (int b = 0; b < 2; ++b) {
        P.ev_in[b] = ev[3 * b];
        P.ev_comp[b] = ev[3 * b + 1];
        P.ev_out[b] = ev[3 * b + 2];
        P.h_count[b] = cnt[b];
    }
    P.ready = true;
    return MOFREAK_OK;
}

// (Re)allocate a pair of page-locked staging buffers: the old pointer is forgotten the moment it is freed and the size is
// recorded only after both allocations succeeded.
int pipe_host_pair(mofreak_ctx *ctx, void *(&slot)[2], size_t &have, size_t want)
{
    if (have >= want && slot[0] && slot[1]) return MOFREAK_OK;
    have = 0;
    for (int b = 0; b < 2; ++b) {
        if (slot[b]) {
            void *old = slot[b];
            slot[b] = nullptr;
            HIP_TRY(ctx, hipHostFree(old));
        }
    }
    for (int b = 0; b < 2; ++b) HIP_TRY(ctx, hipHostMalloc(&slot[b], std::max<size_t>(want, 64), hipHostMallocDefault));
    have = want;
    return MOFREAK_OK;
}

// The frame loop of computeMoFREAKFromFile (MoFREAKUtilities.cpp:391-489) over MANY clips in one pipelined pass -- the
// dataset loop of computeMoFREAKFiles (main.cpp:862-921) without a synchronous call per video.  The clips are laid end to
// end in a virtual frame sequence; a window of chunk_frames frames of it is copied down at a time (consecutive windows
// overlap by `gap` frames), every frame q of the window with q + gap inside it is a pair (previous = q, current = q + gap),
// and the compaction drops the pairs whose two frames belong to different clips (label < 0) and numbers the others
// gap - 1, gap, ... inside their clip (:401, :488).  Window k + 1's copies run under window k's kernels while window
// k - 1's rows travel back.
// A clip may come in several pieces (joined[c] != 0: piece c continues the clip of piece c - 1 -- a stream's frames still on
// the device in front of the chunk that was just pushed) and may start at a frame number of its own (label_base[c], read
// at a clip's first piece: the frames of the stream before it); a piece may be device memory (copied device to device).
// The reference's own keypoint source in the pipelined routes: BriskFeatureDetector(threshold, octaves) on every window's
// difference images (MoFREAKUtilities.cpp:420-423) instead of a caller's keypoint list.
struct ClipDetector {
    int threshold = 30, octaves = 3;
    int64_t n_keypoints = 0;  // out: keypoints detected over the whole call
};
int det_geometry(const mofreak_ctx *ctx, int W, int H, int octaves, DetGeom &g);
int det_batch(const mofreak_ctx *ctx, const DetGeom &g, int n_pairs);

struct ClipPieces {
    const uint8_t *joined = nullptr;
    const int64_t *label_base = nullptr;
    int last_window_slot = -1;       // out: which of the pipeline's two frame buffers holds the sequence's last window ...
    int64_t last_window_first = 0;   // ... and which frame of the sequence is its first
};

int extract_clips_impl(mofreak_ctx *ctx, const uint8_t *const *clip_frames, const int32_t *clip_n_frames, int n_clips, int W,
                       int H, int chunk_frames, const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out,
                       int64_t rows_capacity, int64_t *clip_row_offsets_out, int64_t *n_rows_out, bool rows_on_device,
                       ClipPieces *pieces = nullptr, ClipDetector *det = nullptr)
{
    const int gap = ctx->params.gap_for_frame_difference;
    const int64_t fsz = (int64_t)W * H;
    // the virtual sequence: clip c occupies frames [start[c], start[c + 1])
    std::vector<int64_t> start((size_t)n_clips + 1, 0);
    for (int c = 0; c < n_clips; ++c) {
        if (clip_n_frames[c] < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative frame count");
        if (clip_n_frames[c] > 0 && !clip_frames[c]) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null clip pointer");
        start[c + 1] = start[c] + clip_n_frames[c];
    }
    const int64_t T = start[n_clips];
    if (clip_row_offsets_out) std::fill(clip_row_offsets_out, clip_row_offsets_out + n_clips + 1, (int64_t)0);
    if (T - gap <= 0 || (!det && n_kp == 0)) return MOFREAK_OK;
    if (T - gap >= ((int64_t)1 << 31)) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "more than 2^31 frames in one call");
    const int64_t n_pairs_all = T - gap;
    if (chunk_frames <= gap)  // default: windows of about 96 MiB of frames
        chunk_frames = (int)std::min<int64_t>(4096, std::max<int64_t>(gap + 16, ((int64_t)96 << 20) / fsz));
    chunk_frames = (int)std::min<int64_t>(chunk_frames, T);
    if (det) {  // a window is one detector batch: as many pairs as its workspace takes at once
        DetGeom dg;
        const int e = det_geometry(ctx, W, H, det->octaves, dg);
        if (e) return e;
        chunk_frames = std::min(chunk_frames, det_batch(ctx, dg, chunk_frames - gap) + gap);
        n_kp = 4096;  // per pair, to size the first buffers (they grow with what the detector finds)
    }
    const int chunk_pairs = chunk_frames - gap;
    const int n_chunks = (int)((n_pairs_all + chunk_pairs - 1) / chunk_pairs);
    const Geometry g{W, H, W, fsz};
    const uint8_t *some_frame = nullptr;
    for (int c = 0; c < n_clips && !some_frame; ++c)
        if (clip_n_frames[c] > 0) some_frame = clip_frames[c];
    int rc = validate_frames(ctx, some_frame, some_frame, W, H, W, fsz, chunk_pairs);
    if (rc) return rc;

    // frame labels per pair (a clip = a run of joined pieces), and which piece a frame belongs to (for the segment copies)
    std::vector<int32_t> label((size_t)n_pairs_all);
    for (int c = 0; c < n_clips;) {
        int c1 = c + 1;
        while (c1 < n_clips && pieces && pieces->joined && pieces->joined[c1]) ++c1;
        const int64_t base = pieces && pieces->label_base ? pieces->label_base[c] : 0;
        for (int64_t q = start[c]; q < start[c1] && q < n_pairs_all; ++q)
            label[(size_t)q] = q + gap < start[c1] ? (int32_t)(q - start[c] + base) + gap - 1 : -1;
        c = c1;
    }
    std::vector<char> clip_pinned((size_t)n_clips);  // 1: page-locked or device memory (copied in place), 0: through the staging buffers
    bool all_pinned = true;
    for (int c = 0; c < n_clips; ++c) {
        clip_pinned[c] = clip_n_frames[c] == 0 || is_pinned_host(clip_frames[c]) || is_device_memory(clip_frames[c]);
        all_pinned = all_pinned && clip_pinned[c];
    }
    const bool rows_pinned = rows_on_device || is_pinned_host(rows_out);

    if ((rc = pipe_setup(ctx))) return rc;
    mofreak_ctx::Pipe &P = ctx->pipe;
    PipeDrain drain{ctx, true};  // from here on every return drains the three streams
    const size_t chunk_bytes = (size_t)chunk_frames * fsz, rows_bytes = (size_t)chunk_pairs * n_kp * sizeof(mofreak_row);
    for (int b = 0; b < 2; ++b) {
        if ((rc = ensure(ctx, P.d_frames[b], chunk_bytes))) return rc;
        if ((rc = ensure(ctx, P.d_rows[b], rows_bytes))) return rc;
        if ((rc = ensure(ctx, P.d_pair_rows[b], (size_t)chunk_pairs * sizeof(int32_t)))) return rc;
    }
    if (!all_pinned && (rc = pipe_host_pair(ctx, P.h_frames, P.h_frames_bytes, chunk_bytes))) return rc;
    if (!rows_pinned && (rc = pipe_host_pair(ctx, P.h_rows, P.h_rows_bytes, rows_bytes))) return rc;
    if ((rc = pipe_host_pair(ctx, P.h_pair_rows, P.h_pair_rows_bytes, (size_t)chunk_pairs * sizeof(int32_t)))) return rc;
    if (!det && (rc = upload(ctx, ctx->stage[2], kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
    if ((rc = upload(ctx, ctx->pair_label, label.data(), label.size() * sizeof(int32_t)))) return rc;
    const mofreak_keypoint *d_kps = static_cast<const mofreak_keypoint *>(ctx->stage[2].ptr);
    const int32_t *d_label = static_cast<const int32_t *>(ctx->pair_label.ptr);
    const int64_t items_max = (int64_t)chunk_pairs * n_kp;
    if ((rc = ensure(ctx, ctx->scratch_desc, (size_t)items_max * 16))) return rc;
    if ((rc = ensure(ctx, ctx->scratch_valid, (size_t)items_max))) return rc;
    if ((rc = ensure(ctx, ctx->compact_offsets, ((size_t)(items_max + kCompactItemsPerBlock - 1) / kCompactItemsPerBlock + 1) * sizeof(int64_t)))) return rc;
    uint8_t *desc = static_cast<uint8_t *>(ctx->scratch_desc.ptr);
    uint8_t *valid = static_cast<uint8_t *>(ctx->scratch_valid.ptr);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the uploads above (they read `label` and the caller's kps)
    if (det) det->n_keypoints = 0;

    std::vector<int32_t> rows_of_pair(clip_row_offsets_out ? (size_t)n_pairs_all : 0);
    int64_t total_rows = 0;           // rows produced so far (keeps counting past rows_capacity: the caller learns the size)
    bool overflow = false;
    int64_t chunk_total[2] = {0, 0};  // row count of the window in each slot, once known
    int64_t chunk_dst[2] = {0, 0};    // where its rows go in rows_out
    int64_t chunk_q0[2] = {0, 0};     // its first pair
    int chunk_np[2] = {0, 0};
    int state[2] = {0, 0};            // 0 free, 1 computing (count unknown), 2 rows on their way back
    // hand window `b`'s rows over: count known -> copy issued; copy done -> (copied to the caller)
    auto advance = [&](int b, bool finish) -> int {
        if (state[b] == 1) {
            HIP_TRY(ctx, hipEventSynchronize(P.ev_comp[b]));
            chunk_total[b] = *P.h_count[b];
            chunk_dst[b] = total_rows;
            total_rows += chunk_total[b];
            if (!rows_of_pair.empty())
                std::memcpy(rows_of_pair.data() + chunk_q0[b], P.h_pair_rows[b], (size_t)chunk_np[b] * sizeof(int32_t));
            if (total_rows > rows_capacity) overflow = true;  // no further row leaves the device; the counting goes on
            if (chunk_total[b] && !overflow) {
                void *dst = rows_pinned ? static_cast<void *>(rows_out + chunk_dst[b]) : P.h_rows[b];
                HIP_TRY(ctx, hipMemcpyAsync(dst, P.d_rows[b].ptr, (size_t)chunk_total[b] * sizeof(mofreak_row),
                                            rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, P.s_out));
            }
            HIP_TRY(ctx, hipEventRecord(P.ev_out[b], P.s_out));
            state[b] = 2;
        }
        if (state[b] == 2 && finish) {
            HIP_TRY(ctx, hipEventSynchronize(P.ev_out[b]));
            if (!rows_pinned && chunk_total[b] && !overflow)
                std::memcpy(rows_out + chunk_dst[b], P.h_rows[b], (size_t)chunk_total[b] * sizeof(mofreak_row));
            state[b] = 0;
        }
        return MOFREAK_OK;
    };

    int clip_lo = 0;  // first clip that reaches into the window whose copies are issued next
    // the copies of window k's frames into slot k & 1, on the copy stream (the slot's previous user, window k - 2, is through)
    auto issue_copies = [&](int k) -> int {
        const int b = k & 1;
        const int64_t f0 = (int64_t)k * chunk_pairs;
        const int nf = (int)std::min<int64_t>(chunk_frames, T - f0);
        bool staged_wait = false;
        HIP_TRY(ctx, hipStreamWaitEvent(P.s_in, P.ev_comp[b], 0));  // the device frames of window k-2 are no longer read
        while (clip_lo < n_clips && start[clip_lo + 1] <= f0) ++clip_lo;
        for (int c = clip_lo; c < n_clips && start[c] < f0 + nf; ++c) {  // the pieces of clips inside [f0, f0 + nf)
            const int64_t lo = std::max(start[c], f0), hi = std::min(start[c + 1], f0 + nf);
            if (hi <= lo) continue;
            const uint8_t *src = clip_frames[c] + (lo - start[c]) * fsz;
            uint8_t *dst = static_cast<uint8_t *>(P.d_frames[b].ptr) + (lo - f0) * fsz;
            if (!clip_pinned[c]) {
                if (!staged_wait) {
                    HIP_TRY(ctx, hipEventSynchronize(P.ev_in[b]));  // window k-2's copies out of this staging buffer
                    staged_wait = true;
                }
                uint8_t *stage = static_cast<uint8_t *>(P.h_frames[b]) + (lo - f0) * fsz;
                std::memcpy(stage, src, (size_t)(hi - lo) * fsz);
                src = stage;
            }
            HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)(hi - lo) * fsz, hipMemcpyDefault, P.s_in));  // (host or device piece)
        }
        HIP_TRY(ctx, hipEventRecord(P.ev_in[b], P.s_in));
        return MOFREAK_OK;
    };
    // With the detector a window's compute hands counts to the host (it blocks): the NEXT window's copies are issued before it,
    // so that they still run under this window's kernels.
    if (det && (rc = issue_copies(0))) return rc;
    for (int k = 0; k < n_chunks; ++k) {
        const int b = k & 1;
        const int64_t f0 = (int64_t)k * chunk_pairs;
        const int nf = (int)std::min<int64_t>(chunk_frames, T - f0), np = nf - gap;
        if (!det) {
            if ((rc = advance(b, true))) return rc;  // slot b: window k-2 is through (its buffers are free again)
            if ((rc = issue_copies(k))) return rc;
        } else if (k + 1 < n_chunks) {
            if ((rc = advance(b ^ 1, true))) return rc;  // slot of window k+1: window k-1 is through
            if ((rc = issue_copies(k + 1))) return rc;
        }
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, P.ev_in[b], 0));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, P.ev_out[b], 0));  // the device rows of window k-2 have left
        const uint8_t *d_fr = static_cast<const uint8_t *>(P.d_frames[b].ptr);
        const mofreak_keypoint *w_kps = d_kps;   // the window's keypoints: the caller's list, or what the detector finds
        const int64_t *w_off = nullptr;
        int64_t w_nkp = n_kp, n_items = (int64_t)np * n_kp;
        if (det) {
            if ((rc = advance(b, true))) return rc;  // (the first two windows: nothing to finish)
            int64_t found = 0;
            for (int attempt = 0;; ++attempt) {
                const int64_t cap = std::max<int64_t>(ctx->det_kp_capacity, (int64_t)4096 * np);
                if ((rc = ensure(ctx, ctx->det_out_kps, (size_t)cap * sizeof(mofreak_keypoint)))) return rc;
                if ((rc = ensure(ctx, ctx->det_out_offsets, (size_t)(np + 1) * sizeof(int64_t)))) return rc;
                rc = mofreak_detect_pairs(ctx, d_fr + (int64_t)gap * fsz, d_fr, W, H, W, fsz, np, det->threshold, det->octaves,
                                          static_cast<mofreak_keypoint *>(ctx->det_out_kps.ptr), cap, static_cast<int64_t *>(ctx->det_out_offsets.ptr), nullptr, nullptr, &found,
                                          MOFREAK_MEM_DEVICE);
                if (rc == MOFREAK_ERR_CAPACITY && found > cap && attempt == 0) {
                    ctx->det_kp_capacity = found + found / 8;
                    continue;
                }
                if (rc) return rc;
                break;
            }
            det->n_keypoints += found;
            w_kps = static_cast<const mofreak_keypoint *>(ctx->det_out_kps.ptr);
            w_off = static_cast<const int64_t *>(ctx->det_out_offsets.ptr);
            w_nkp = n_items = found;
            if (!rows_pinned && (size_t)found * sizeof(mofreak_row) > P.h_rows_bytes) {
                // more rows than the host staging buffers take (both slots are idle here: this window has not produced yet,
                // the other one is waiting for its frames)
                if ((rc = advance(b ^ 1, true))) return rc;
                if ((rc = pipe_host_pair(ctx, P.h_rows, P.h_rows_bytes, (size_t)(found + found / 4) * sizeof(mofreak_row)))) return rc;
            }
            if ((rc = ensure(ctx, P.d_rows[b], (size_t)std::max<int64_t>(found, 1) * sizeof(mofreak_row)))) return rc;
            if ((rc = ensure(ctx, ctx->scratch_desc, (size_t)std::max<int64_t>(found, 1) * 16))) return rc;
            if ((rc = ensure(ctx, ctx->scratch_valid, (size_t)std::max<int64_t>(found, 1)))) return rc;
            if ((rc = ensure(ctx, ctx->compact_offsets, ((size_t)(found + kCompactItemsPerBlock - 1) / kCompactItemsPerBlock + 1) * sizeof(int64_t)))) return rc;
            desc = static_cast<uint8_t *>(ctx->scratch_desc.ptr);
            valid = static_cast<uint8_t *>(ctx->scratch_valid.ptr);
            if (found > 0) {
                std::vector<int64_t> h_off;
                if ((rc = fetch_offsets(ctx, w_off, np, false, h_off))) return rc;
                struct DiffScope {  // the detector's difference planes serve this window's integral images, and nobody else's
                    mofreak_ctx *c;
                    ~DiffScope()
                    {
                        c->use_det_diff = false;
                        c->det_diff = mofreak_ctx::DetDiff{};
                    }
                } scope{ctx};
                ctx->use_det_diff = true;
                rc = extract_device(ctx, d_fr + (int64_t)gap * fsz, d_fr, g, np, w_kps, w_off, h_off.data(), found, desc, valid, nullptr, nullptr);
                if (rc) return rc;
            }
        } else {
            rc = extract_device(ctx, d_fr + (int64_t)gap * fsz, d_fr, g, np, d_kps, nullptr, nullptr, n_kp, desc, valid, nullptr, nullptr);
            if (rc) return rc;
        }
        {
            const int n_blocks = (int)((n_items + kCompactItemsPerBlock - 1) / kCompactItemsPerBlock);
            CompactArgs c;
            c.kps = w_kps;
            c.kp_offsets = w_off;
            c.n_kp = w_nkp;
            c.n_items = n_items;
            c.n_pairs = np;
            c.first_frame_number = 0;
            c.pair_label = d_label + f0;  // labels restart with every clip (:401, :488); < 0: the pair straddles two clips
            c.pair_rows = rows_of_pair.empty() ? nullptr : static_cast<int32_t *>(P.d_pair_rows[b].ptr);  // only when per-clip offsets are asked for
            c.desc = desc;
            c.valid = valid;
            c.rows = static_cast<mofreak_row *>(P.d_rows[b].ptr);
            c.capacity = n_items;
            c.block_offsets = static_cast<int64_t *>(ctx->compact_offsets.ptr);
            c.n_blocks = n_blocks;
            if (c.pair_rows) HIP_TRY(ctx, hipMemsetAsync(c.pair_rows, 0, (size_t)np * sizeof(int32_t), ctx->stream));
            const int e = launch_compact(c, ctx->stream);
            if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("compact launch: ") + hipGetErrorString((hipError_t)e));
            HIP_TRY(ctx, hipMemcpyAsync(P.h_count[b], c.block_offsets + n_blocks, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            if (!rows_of_pair.empty())
                HIP_TRY(ctx, hipMemcpyAsync(P.h_pair_rows[b], c.pair_rows, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(ctx, hipEventRecord(P.ev_comp[b], ctx->stream));
        state[b] = 1;
        chunk_q0[b] = f0;
        chunk_np[b] = np;
        if (pieces) {
            pieces->last_window_slot = b;
            pieces->last_window_first = f0;
        }
        if ((rc = advance(b ^ 1, false))) return rc;  // window k-1: its count is in (or we wait for it) -> rows start travelling
    }
    for (int k = n_chunks; k < n_chunks + 2; ++k)
        if ((rc = advance(k & 1, true))) return rc;
    if (n_rows_out) *n_rows_out = total_rows;
    if (clip_row_offsets_out) {
        int64_t acc = 0;
        for (int c = 0; c < n_clips; ++c) {
            for (int64_t q = start[c]; q < start[c + 1] && q < n_pairs_all; ++q) acc += rows_of_pair[(size_t)q];
            clip_row_offsets_out[c + 1] = acc;
        }
    }
    if (overflow)
        return fail(ctx, MOFREAK_ERR_CAPACITY, "rows_out too small: " + std::to_string(total_rows) + " rows needed (returned through n_rows_out)");
    return MOFREAK_OK;
}

}  // namespace

int mofreak_extract_clips(mofreak_ctx *ctx, const uint8_t *const *clip_frames, const int32_t *clip_n_frames, int n_clips, int W, int H,
                          int chunk_frames, const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out, int64_t rows_capacity,
                          int64_t *clip_row_offsets_out, int64_t *n_rows_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows_out) *n_rows_out = 0;
    if (n_clips < 0 || n_kp < 0 || rows_capacity < 0 || W <= 0 || H <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count or empty frame");
    if (n_clips == 0) return MOFREAK_OK;
    if (!clip_frames || !clip_n_frames) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null clip table");
    if (n_kp > 0 && !kps) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null keypoint pointer");
    if (rows_capacity > 0 && !rows_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null rows_out");
    NEED_DEVICE(ctx);
    return extract_clips_impl(ctx, clip_frames, clip_n_frames, n_clips, W, H, chunk_frames, kps, n_kp, rows_out, rows_capacity,
                              clip_row_offsets_out, n_rows_out, (flags & MOFREAK_ROWS_DEVICE) != 0);
}

int mofreak_compute_clips(mofreak_ctx *ctx, const uint8_t *const *clip_frames, const int32_t *clip_n_frames, int n_clips, int W, int H, int chunk_frames,
                          int threshold, int octaves, mofreak_row *rows_out, int64_t rows_capacity, int64_t *clip_row_offsets_out, int64_t *n_rows_out,
                          int64_t *n_keypoints_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows_out) *n_rows_out = 0;
    if (n_keypoints_out) *n_keypoints_out = 0;
    if (n_clips < 0 || rows_capacity < 0 || W <= 0 || H <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count or empty frame");
    if (threshold < 1 || threshold > 255) return fail(ctx, MOFREAK_ERR_BAD_ARG, "threshold must be in 1..255");
    if (n_clips == 0) return MOFREAK_OK;
    if (!clip_frames || !clip_n_frames) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null clip table");
    if (rows_capacity > 0 && !rows_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null rows_out");
    NEED_DEVICE(ctx);
    ClipDetector det;
    det.threshold = threshold;
    det.octaves = octaves;
    const int rc = extract_clips_impl(ctx, clip_frames, clip_n_frames, n_clips, W, H, chunk_frames, nullptr, 0, rows_out, rows_capacity, clip_row_offsets_out, n_rows_out,
                                      (flags & MOFREAK_ROWS_DEVICE) != 0, nullptr, &det);
    if (n_keypoints_out) *n_keypoints_out = det.n_keypoints;
    return rc;
}

int mofreak_extract_stream_pipelined(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H, int chunk_frames,
                                     const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out,
                                     int64_t rows_capacity, int64_t *n_rows_out)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows_out) *n_rows_out = 0;
    if (T < 0 || n_kp < 0 || rows_capacity < 0 || W <= 0 || H <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count or empty frame");
    const int gap = ctx->params.gap_for_frame_difference;
    if (T - gap <= 0 || n_kp == 0) return MOFREAK_OK;
    if (!frames || !kps || !rows_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    if (chunk_frames <= gap) chunk_frames = 256;
    const int32_t n_frames = T;  // one clip: the stream
    return extract_clips_impl(ctx, &frames, &n_frames, 1, W, H, chunk_frames, kps, n_kp, rows_out, rows_capacity, nullptr, n_rows_out, false);
}

int mofreak_bgr_to_gray(mofreak_ctx *ctx, const uint8_t *bgr, int W, int H, int64_t row_stride, int64_t frame_stride,
                        int n_frames, uint8_t *gray_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (W <= 0 || H <= 0 || n_frames < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "W, H must be positive and n_frames >= 0");
    if (row_stride < (int64_t)3 * W) return fail(ctx, MOFREAK_ERR_BAD_ARG, "row_stride < 3*W");
    if (n_frames == 0) return MOFREAK_OK;
    if (!bgr || !gray_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    if (n_frames > 1 && frame_stride == 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "frame_stride is 0");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const uint8_t *d_in = bgr;
    uint8_t *d_out = gray_out;
    const size_t out_bytes = (size_t)n_frames * W * H;
    int rc;
    if (host) {
        const size_t span = (size_t)((int64_t)(n_frames - 1) * frame_stride + (int64_t)(H - 1) * row_stride + (int64_t)3 * W);
        if ((rc = upload(ctx, ctx->stage[0], bgr, span))) return rc;
        if ((rc = ensure(ctx, ctx->stage[3], out_bytes))) return rc;
        d_in = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        d_out = static_cast<uint8_t *>(ctx->stage[3].ptr);
    }
    const int e = launch_bgr2gray(d_in, W, H, row_stride, frame_stride, n_frames, d_out, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("bgr2gray launch: ") + hipGetErrorString((hipError_t)e));
    if (host) {
        HIP_TRY(ctx, hipMemcpyAsync(gray_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MOFREAK_OK;
}

// ------------------------------------------------------------------ bag-of-words assignment
static int bow_common(mofreak_ctx *ctx, const uint8_t *desc16, const uint8_t *valid, int64_t n, const uint8_t *codebook16,
                      int n_codewords, int32_t *out_index, float *hist_out, int32_t *success_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n < 0 || n_codewords <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n >= 0 and n_codewords > 0 required");
    if (n_codewords > 10240) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "codebook larger than 10240 codewords (the packed key's index field; the reference's largest is 10100)");
    if (!codebook16 || (n > 0 && !desc16)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    int rc;
    const uint8_t *d_desc = desc16, *d_valid = valid, *d_cb = codebook16;
    int32_t *d_idx = out_index;
    float *d_hist = hist_out;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], desc16, (size_t)n * 16))) return rc;
        d_desc = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        if (valid) {
            if ((rc = upload(ctx, ctx->stage[1], valid, (size_t)n))) return rc;
            d_valid = static_cast<const uint8_t *>(ctx->stage[1].ptr);
        }
        if ((rc = upload(ctx, ctx->stage[2], codebook16, (size_t)n_codewords * 16))) return rc;
        d_cb = static_cast<const uint8_t *>(ctx->stage[2].ptr);
        if (out_index) {
            if ((rc = ensure(ctx, ctx->stage[3], (size_t)std::max<int64_t>(n, 1) * 4))) return rc;
            d_idx = static_cast<int32_t *>(ctx->stage[3].ptr);
        }
        if (hist_out) {
            if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_codewords * 4 + 16))) return rc;
            d_hist = static_cast<float *>(ctx->stage[4].ptr);
        }
    }
    unsigned int *d_counts = nullptr;
    int32_t *d_success = nullptr;
    if (hist_out) {
        if ((rc = ensure(ctx, ctx->bow_counts, (size_t)n_codewords * 4 + 16))) return rc;
        d_counts = static_cast<unsigned int *>(ctx->bow_counts.ptr);
        d_success = reinterpret_cast<int32_t *>(d_counts + n_codewords);
        HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, (size_t)n_codewords * 4 + 16, ctx->stream));
    }
    if (n > 0) {
        if ((rc = ensure(ctx, ctx->bow_expanded, bow_expanded_bytes(n_codewords)))) return rc;
        const int e = launch_bow_assign(d_desc, d_valid, n, d_cb, n_codewords, d_idx, d_counts, ctx->n_cus, ctx->bow_expanded.ptr, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("bow_assign launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (hist_out) {
        const int e = launch_bow_normalize(d_counts, n_codewords, d_hist, d_success, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, "bow_normalize launch failed");
        int32_t ok = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&ok, d_success, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (host) HIP_TRY(ctx, hipMemcpyAsync(hist_out, d_hist, (size_t)n_codewords * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (success_out) *success_out = ok;
    }
    if (host && out_index && n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(out_index, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MOFREAK_OK;
}

int mofreak_bow_assign(mofreak_ctx *ctx, const uint8_t *desc16, const uint8_t *valid, int64_t n, const uint8_t *codebook16,
                       int n_codewords, int32_t *out_index, unsigned flags)
{
    if (ctx && n > 0 && !out_index) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null output");
    return bow_common(ctx, desc16, valid, n, codebook16, n_codewords, out_index, nullptr, nullptr, flags);
}

int mofreak_bow_histogram(mofreak_ctx *ctx, const uint8_t *desc16, const uint8_t *valid, int64_t n, const uint8_t *codebook16,
                          int n_codewords, float *hist_out, int32_t *success_out, unsigned flags)
{
    if (ctx && !hist_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null output");
    return bow_common(ctx, desc16, valid, n, codebook16, n_codewords, nullptr, hist_out, success_out, flags);
}

// ------------------------------------------------------------------ component entry points
int mofreak_diff_integral(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, int64_t row_stride,
                          int64_t pair_stride, int n_pairs, int32_t *out, unsigned flags)
{
    int rc = validate_frames(ctx, cur, prev, W, H, row_stride, pair_stride, n_pairs);
    if (rc) return rc;
    if (n_pairs == 0) return MOFREAK_OK;
    if (!out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null output");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry g{W, H, row_stride, pair_stride};
    const uint8_t *d_cur = cur, *d_prev = prev;
    if (host) {
        const size_t span = frame_span(g, n_pairs);
        if ((rc = upload(ctx, ctx->stage[0], cur, span))) return rc;
        if ((rc = upload(ctx, ctx->stage[1], prev, span))) return rc;
        d_cur = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        d_prev = static_cast<const uint8_t *>(ctx->stage[1].ptr);
    }
    const size_t per_pair = (size_t)(W + 1) * (H + 1);
    int32_t *d_out = out;
    if (host) {
        if ((rc = ensure(ctx, ctx->stage[3], per_pair * sizeof(int32_t)))) return rc;
        d_out = static_cast<int32_t *>(ctx->stage[3].ptr);
    }
    for (int p = 0; p < n_pairs; ++p) {
        if ((rc = run_integral(ctx, d_cur, d_prev, g, p, 1))) return rc;
        int32_t *dst = host ? d_out : out + (size_t)p * per_pair;
        const int e = launch_unpack_integral(static_cast<const int32_t *>(ctx->integral.ptr), integral_pitch(W), W, H, 1, dst, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, "unpack launch failed");
        if (host) {
            HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)p * per_pair, d_out, per_pair * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return MOFREAK_OK;
}

int mofreak_mip19(mofreak_ctx *ctx, const uint8_t *cur19, const uint8_t *prev19, int64_t n, uint8_t *out_motion8,
                  unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n < 0");
    if (n == 0) return MOFREAK_OK;
    if (!cur19 || !prev19 || !out_motion8) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    int rc;
    const uint8_t *dc = cur19, *dp = prev19;
    uint8_t *dout = out_motion8;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], cur19, (size_t)n * 361))) return rc;
        if ((rc = upload(ctx, ctx->stage[1], prev19, (size_t)n * 361))) return rc;
        if ((rc = ensure(ctx, ctx->stage[3], (size_t)n * 8))) return rc;
        dc = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        dp = static_cast<const uint8_t *>(ctx->stage[1].ptr);
        dout = static_cast<uint8_t *>(ctx->stage[3].ptr);
    }
    const int e = launch_mip19(dc, dp, n, ctx->params.mip_theta, dout, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, "mip19 launch failed");
    if (host) {
        HIP_TRY(ctx, hipMemcpyAsync(out_motion8, dout, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MOFREAK_OK;
}

static int one_pair_debug(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H,
                          const mofreak_keypoint *kps, int64_t n_kp, int32_t *out_info, uint8_t *out_roi, unsigned flags)
{
    int rc = validate_frames(ctx, cur, prev, W, H, W, (int64_t)W * H, 1);
    if (rc) return rc;
    if (n_kp < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n_kp < 0");
    if (n_kp == 0) return MOFREAK_OK;
    if (!kps || (!out_info && !out_roi)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry g{W, H, W, (int64_t)W * H};
    const uint8_t *dc = cur, *dp = prev;
    const mofreak_keypoint *dk = kps;
    const size_t out_bytes = out_info ? (size_t)n_kp * 16 : (size_t)n_kp * 722;
    void *dout = out_info ? (void *)out_info : (void *)out_roi;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], cur, (size_t)W * H))) return rc;
        if ((rc = upload(ctx, ctx->stage[1], prev, (size_t)W * H))) return rc;
        if ((rc = upload(ctx, ctx->stage[2], kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
        if ((rc = ensure(ctx, ctx->stage[5], out_bytes))) return rc;
        dc = static_cast<const uint8_t *>(ctx->stage[0].ptr);
        dp = static_cast<const uint8_t *>(ctx->stage[1].ptr);
        dk = static_cast<const mofreak_keypoint *>(ctx->stage[2].ptr);
        dout = ctx->stage[5].ptr;
    }
    if ((rc = ensure(ctx, ctx->scratch_desc, (size_t)n_kp * 16))) return rc;
    if ((rc = ensure(ctx, ctx->scratch_valid, (size_t)n_kp))) return rc;
    rc = extract_device(ctx, dc, dp, g, 1, dk, nullptr, nullptr, n_kp, static_cast<uint8_t *>(ctx->scratch_desc.ptr),
                        static_cast<uint8_t *>(ctx->scratch_valid.ptr), out_info ? static_cast<int32_t *>(dout) : nullptr,
                        out_info ? nullptr : static_cast<uint8_t *>(dout));
    if (rc) return rc;
    if (host) {
        HIP_TRY(ctx, hipMemcpyAsync(out_info ? (void *)out_info : (void *)out_roi, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MOFREAK_OK;
}

int mofreak_roi19(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, const mofreak_keypoint *kps,
                  int64_t n_kp, uint8_t *out, unsigned flags)
{
    return one_pair_debug(ctx, cur, prev, W, H, kps, n_kp, nullptr, out, flags);
}

int mofreak_freak_info(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, const mofreak_keypoint *kps,
                       int64_t n_kp, int32_t *out_info, unsigned flags)
{
    return one_pair_debug(ctx, cur, prev, W, H, kps, n_kp, out_info, nullptr, flags);
}

int mofreak_theta_index(mofreak_ctx *ctx, const int32_t *dirs, int64_t n, int32_t *out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n < 0");
    if (n == 0) return MOFREAK_OK;
    if (!dirs || !out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    int rc;
    const int32_t *dd = dirs;
    int32_t *dout = out;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], dirs, (size_t)n * 8))) return rc;
        if ((rc = ensure(ctx, ctx->stage[3], (size_t)n * 4))) return rc;
        dd = static_cast<const int32_t *>(ctx->stage[0].ptr);
        dout = static_cast<int32_t *>(ctx->stage[3].ptr);
    }
    const int e = launch_theta(ctx->d_theta, dd, n, dout, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, "theta launch failed");
    if (host) {
        HIP_TRY(ctx, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MOFREAK_OK;
}

int mofreak_pattern_sizes(const mofreak_ctx *ctx, int32_t out[64])
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    std::memcpy(out, ctx->tables.pattern_sizes, sizeof(int32_t) * 64);
    return MOFREAK_OK;
}

int mofreak_scale_index(const mofreak_ctx *ctx, float size, int32_t *out)
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    if (!ctx->params.freak_scale_normalized) {
        *out = ctx->tables.fixed_scale_index;
        return MOFREAK_OK;
    }
    int idx = 0;  // the device's rule: number of thresholds <= size
    for (int k = 0; k < kNbScales - 1; ++k) idx += size >= ctx->tables.scale_thresholds[k];
    *out = idx;
    return MOFREAK_OK;
}

int mofreak_table_pattern(const mofreak_ctx *ctx, int scale, int rot, float out[43 * 3])
{
    if (!ctx || !out || scale < 0 || scale >= kNbScales || rot < 0 || rot >= kNbOrientation) return MOFREAK_ERR_BAD_ARG;
    const PatternPoint *row = &ctx->tables.lut[((size_t)scale * kNbOrientation + rot) * kNbPoints];
    for (int i = 0; i < kNbPoints; ++i) {
        out[3 * i] = row[i].x;
        out[3 * i + 1] = row[i].y;
        out[3 * i + 2] = row[i].sigma;
    }
    return MOFREAK_OK;
}

int mofreak_table_orientation(const mofreak_ctx *ctx, int32_t out[45 * 4])
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    for (int m = 0; m < kNbOrientPairs; ++m) {
        out[4 * m] = ctx->tables.orient[m].i;
        out[4 * m + 1] = ctx->tables.orient[m].j;
        out[4 * m + 2] = ctx->tables.orient[m].weight_dx;
        out[4 * m + 3] = ctx->tables.orient[m].weight_dy;
    }
    return MOFREAK_OK;
}

int mofreak_table_bit_pairs(const mofreak_ctx *ctx, uint8_t out[128])
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    for (int b = 0; b < 64; ++b) {
        out[2 * b] = ctx->tables.bit_pair_i[b];
        out[2 * b + 1] = ctx->tables.bit_pair_j[b];
    }
    return MOFREAK_OK;
}

int mofreak_table_mip_positions(const mofreak_ctx *ctx, int L, uint16_t out[320], int32_t *n_out)
{
    if (!ctx || !out || L < 1 || L > kTileMaxRoi) return MOFREAK_ERR_BAD_ARG;
    const Tables &t = ctx->tables;
    if (t.mip_n > 320) return MOFREAK_ERR_UNSUPPORTED;
    for (int j = 0; j < t.mip_n; ++j) out[j] = t.mip_pos[(size_t)L * t.mip_stride + j];
    if (n_out) *n_out = t.mip_n;
    return MOFREAK_OK;
}

int mofreak_table_resize(const mofreak_ctx *ctx, int L, int16_t out[2 * 19 * 4])
{
    if (!ctx || !out || L < 1 || L > kMaxRoiSide) return MOFREAK_ERR_BAD_ARG;
    const ResizeTap *t = &ctx->tables.resize[(size_t)L * 2 * kPatch];
    for (int i = 0; i < 2 * kPatch; ++i) {
        out[4 * i] = t[i].ofs;
        out[4 * i + 1] = t[i].ofs1;
        out[4 * i + 2] = t[i].c0;
        out[4 * i + 3] = t[i].c1;
    }
    return MOFREAK_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ keypoint detector
namespace {

// BriskScaleSpace(octaves) + the layer constructors (brisk.cpp:561-567, 1646-1674): sizes, scale and offset per layer
int det_geometry(const mofreak_ctx *ctx, int W, int H, int octaves, DetGeom &g)
{
    if (octaves < 0 || octaves > kDetMaxLayers / 2) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "octaves must be in 0..4");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(ctx, MOFREAK_ERR_BAD_ARG, "W, H must be in 1..65535");
    std::memset(&g, 0, sizeof(g));
    g.n_layers = octaves == 0 ? 1 : 2 * octaves;
    int64_t off = 0;
    int rows = 0;
    for (int i = 0; i < g.n_layers; ++i) {
        DetLayer &L = g.L[i];
        if (i == 0) {
            L.w = W;
            L.h = H;
            L.scale = 1.0f;
            L.offset = 0.0f;
        } else if (i == 1) {
            L.w = 2 * (g.L[0].w / 3);
            L.h = 2 * (g.L[0].h / 3);
            L.scale = (float)((double)g.L[0].scale * 1.5);
            L.offset = (float)(0.5 * (double)L.scale - 0.5);
        } else {
            L.w = g.L[i - 2].w / 2;
            L.h = g.L[i - 2].h / 2;
            L.scale = g.L[i - 2].scale * 2;
            L.offset = (float)(0.5 * (double)L.scale - 0.5);
        }
        L.off = off;
        L.row_base = rows;
        off += (((int64_t)L.w * L.h + 63) / 64) * 64;
        rows += L.h;
    }
    g.plane_bytes = off + 64;
    g.total_rows = rows;
    // the corner kernel's tiles and hit-mask words (one 64-bit word per tile row and layer row)
    int64_t tiles = 0, words = 0;
    for (int i = 0; i < g.n_layers; ++i) {
        g.tiles_x[i] = (g.L[i].w + kDetTileW - 1) / kDetTileW;
        g.tile_start[i] = (int32_t)tiles;
        g.mask_off[i] = words;
        tiles += (int64_t)g.tiles_x[i] * ((g.L[i].h + kDetTileH - 1) / kDetTileH);
        words += (int64_t)g.tiles_x[i] * g.L[i].h;
    }
    if (tiles >= ((int64_t)1 << 31)) return fail(ctx, MOFREAK_ERR_UNSUPPORTED, "frame too large for the detector's tile list");
    for (int i = g.n_layers; i <= kDetMaxLayers; ++i) g.tile_start[i] = (int32_t)tiles;
    g.mask_words = words;
    int groups = 0;
    for (int i = 0; i < g.n_layers; ++i) {
        g.cand_rows_per_wave[i] = std::max(1, 64 / std::max(1, g.tiles_x[i]));
        g.cand_group_start[i] = groups;
        groups += (g.L[i].h + g.cand_rows_per_wave[i] - 1) / g.cand_rows_per_wave[i];
    }
    for (int i = g.n_layers; i <= kDetMaxLayers; ++i) g.cand_group_start[i] = groups;
    return MOFREAK_OK;
}

// Candidates (corners of all layers) per pair the workspace is laid out for: an eighth of the frame's pixels -- difference
// images have 1-2 % -- at least 4096, grown by calls that met more, never above the context's limit.  The grids of the
// per-candidate kernels and the pairs a batch takes follow from it: a 320 x 240 pair does not pay for a full-HD pair's lists.
int det_cand_eff(const mofreak_ctx *ctx, const DetGeom &g)
{
    if (!ctx->det_cand_auto) return ctx->det_cand_cap;
    const int64_t px = (int64_t)g.L[0].w * g.L[0].h;
    const int64_t want = ((std::max<int64_t>(4096, px / 8) + 1023) / 1024 * 1024) << std::min(ctx->det_cand_shift, 16);
    return (int)std::min<int64_t>(ctx->det_cand_cap, want);
}

int det_workspace(mofreak_ctx *ctx, const DetGeom &g, int batch, DetArgs &a)
{
    int rc;
    const int cand_cap = det_cand_eff(ctx, g);
    const size_t planes = (size_t)batch * g.plane_bytes, cands = (size_t)batch * cand_cap;
    if ((rc = ensure(ctx, ctx->det_img, planes))) return rc;
    if ((rc = ensure(ctx, ctx->det_score, planes))) return rc;
    // The two bookkeeping maps of the tie logic are all zero between calls: the emission takes back every byte a call set.
    // They are filled once when they are (re)allocated, and again after a call that did not run to its end.
    // (A map that has to grow is a new buffer with whatever it holds -- and it may well come back at the address the old one
    // had, so the address says nothing: round 4's first version compared addresses, and once in some ten thousand calls of
    // growing frame sizes a tie met a stale byte.)
    if (!ctx->det_touch.ptr || ctx->det_touch.bytes < planes || !ctx->det_status.ptr || ctx->det_status.bytes < planes) ctx->det_maps_dirty = true;
    if ((rc = ensure(ctx, ctx->det_touch, planes))) return rc;
    if ((rc = ensure(ctx, ctx->det_status, planes))) return rc;
    if (ctx->det_maps_dirty) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->det_touch.ptr, 0, ctx->det_touch.bytes, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->det_status.ptr, 0, ctx->det_status.bytes, ctx->stream));
        ctx->det_maps_dirty = false;
    }
    if ((rc = ensure(ctx, ctx->det_hit_mask, (size_t)batch * g.mask_words * sizeof(unsigned long long)))) return rc;
    a.walk_chunks = (cand_cap + 511) / 512;
    if ((rc = ensure(ctx, ctx->det_walk_list, cands * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_walk_count, (size_t)batch * a.walk_chunks * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_cells, cands * 64))) return rc;
    if ((rc = ensure(ctx, ctx->det_tie_list, cands * 8))) return rc;
    if ((rc = ensure(ctx, ctx->det_tie_count, (size_t)batch * a.walk_chunks * sizeof(int32_t)))) return rc;
    // one buffer, one fill: the call's running keypoint total (16 bytes) and the per-row counts
    const size_t n_row_counts = (size_t)batch * (g.total_rows + 1);
    if ((rc = ensure(ctx, ctx->det_rows, 16 + n_row_counts * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_xy, cands * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_flag, cands + 64))) return rc;  // (read sixteen at a time)
    if ((rc = ensure(ctx, ctx->det_cand_emit, cands + 64))) return rc;  // (read sixteen at a time)
    if ((rc = ensure(ctx, ctx->det_cand_spec, cands))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_asked, cands * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_win, cands * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_cand_res, cands * sizeof(DetResult)))) return rc;
    if ((rc = ensure(ctx, ctx->det_layer_start, (size_t)batch * (kDetMaxLayers + 1) * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_emit_count, (size_t)batch * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->det_emit_offsets, (size_t)(batch + 1) * sizeof(int64_t)))) return rc;
    a.emit_chunk_cap = (cand_cap + 1023) / 1024;
    if ((rc = ensure(ctx, ctx->det_emit_chunks, (size_t)batch * a.emit_chunk_cap * sizeof(int32_t)))) return rc;
    a.g = g;
    // the device copy of the geometry: sent when it changes (a stream of equal-sized frames: once)
    if (!ctx->det_geom.ptr || !ctx->det_geom_sent_valid || std::memcmp(&ctx->det_geom_sent, &g, sizeof(DetGeom)) != 0) {
        ctx->det_geom_sent_valid = false;
        ctx->det_geom_sent = g;  // (a member: the copy below may still be reading it when this function returns)
        if ((rc = upload(ctx, ctx->det_geom, &ctx->det_geom_sent, sizeof(DetGeom)))) return rc;
        ctx->det_geom_sent_valid = true;
    }
    a.dg = static_cast<const DetGeom *>(ctx->det_geom.ptr);
    a.img = static_cast<uint8_t *>(ctx->det_img.ptr);
    a.score = static_cast<uint8_t *>(ctx->det_score.ptr);
    a.touch = static_cast<uint8_t *>(ctx->det_touch.ptr);
    a.status = static_cast<uint8_t *>(ctx->det_status.ptr);
    a.row_count = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(ctx->det_rows.ptr) + 16);
    a.hit_mask = static_cast<unsigned long long *>(ctx->det_hit_mask.ptr);
    a.walk_list = static_cast<int32_t *>(ctx->det_walk_list.ptr);
    a.walk_count = static_cast<int32_t *>(ctx->det_walk_count.ptr);
    a.cand_cells = static_cast<uint8_t *>(ctx->det_cand_cells.ptr);
    a.tie_list = static_cast<DetTie *>(ctx->det_tie_list.ptr);
    a.tie_count = static_cast<int32_t *>(ctx->det_tie_count.ptr);
    ctx->det_counter_bytes = n_row_counts * sizeof(int32_t);
    a.cand_cap = cand_cap;
    a.cand_xy = static_cast<uint32_t *>(ctx->det_cand_xy.ptr);
    a.cand_flag = static_cast<uint8_t *>(ctx->det_cand_flag.ptr);
    a.cand_emit = static_cast<uint8_t *>(ctx->det_cand_emit.ptr);
    a.cand_spec = static_cast<uint8_t *>(ctx->det_cand_spec.ptr);
    a.cand_asked = static_cast<unsigned long long *>(ctx->det_cand_asked.ptr);
    a.cand_win = static_cast<uint32_t *>(ctx->det_cand_win.ptr);
    a.cand_res = static_cast<DetResult *>(ctx->det_cand_res.ptr);
    a.layer_start = static_cast<int32_t *>(ctx->det_layer_start.ptr);
    a.emit_count = static_cast<int32_t *>(ctx->det_emit_count.ptr);
    a.emit_offsets = static_cast<int64_t *>(ctx->det_emit_offsets.ptr);
    a.emit_chunks = static_cast<int32_t *>(ctx->det_emit_chunks.ptr);
    // the detector's own status word sits behind the call's running total: one fill clears both, one copy fetches both
    a.status_word = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(ctx->det_rows.ptr) + 8);
    a.fp_x87 = ctx->params.brisk_fp_model == MOFREAK_FP_X87 ? 1 : 0;
    return MOFREAK_OK;
}

// pairs per batch: about 8 GiB of planes and candidate records (a small share of the 288 GB): the tie rounds are a
// chain of short latency-bound launches per batch, so the more pairs share them the better
int det_batch(const mofreak_ctx *ctx, const DetGeom &g, int n_pairs)
{
    const size_t per_pair = 4 * (size_t)g.plane_bytes + (size_t)g.mask_words * 8 +
                            (size_t)det_cand_eff(ctx, g) * (sizeof(uint32_t) + 3 + 12 + sizeof(DetResult) + sizeof(int32_t) + 64 + 8);
    const size_t b = std::max<size_t>(1, ((size_t)8 << 30) / per_pair);
    return (int)std::min<size_t>({b, (size_t)std::max(n_pairs, 1), (size_t)16384});
}

}  // namespace

extern "C" {

int mofreak_detect_set_capacity(mofreak_ctx *ctx, int candidates_per_pair)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (candidates_per_pair == 0) {  // back to the default: room by frame size, at most 131072 per pair
        ctx->det_cand_auto = true;
        ctx->det_cand_cap = 131072;
        return MOFREAK_OK;
    }
    if (candidates_per_pair < 256 || candidates_per_pair > (1 << 24)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "candidates_per_pair must be 0 (automatic) or in 256..2^24");
    ctx->det_cand_cap = candidates_per_pair;
    ctx->det_cand_auto = false;
    return MOFREAK_OK;
}

int mofreak_detect_pairs(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, int64_t row_stride, int64_t pair_stride,
                         int n_pairs, int threshold, int octaves, mofreak_keypoint *out_kps, int64_t capacity, int64_t *out_offsets,
                         float *out_response, int32_t *out_layer, int64_t *n_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_out) *n_out = 0;
    if (n_pairs < 0 || capacity < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "n_pairs and capacity must not be negative");
    if (threshold < 1 || threshold > 255) return fail(ctx, MOFREAK_ERR_BAD_ARG, "threshold must be in 1..255");
    if (row_stride < W) return fail(ctx, MOFREAK_ERR_BAD_ARG, "row_stride < W");
    if (n_pairs > 1 && pair_stride == 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "pair_stride is 0");
    if (!out_offsets || (capacity > 0 && !out_kps) || (n_pairs > 0 && !cur)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    DetGeom g;
    int rc = det_geometry(ctx, W, H, octaves, g);
    if (rc) return rc;
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry geo{W, H, row_stride, pair_stride};
    const uint8_t *d_cur = cur, *d_prev = prev;
    mofreak_keypoint *d_kps = out_kps;
    int64_t *d_off = out_offsets;
    float *d_resp = out_response;
    int32_t *d_layer = out_layer;
    if (host) {
        if (n_pairs > 0) {
            if ((rc = upload(ctx, ctx->stage[0], cur, frame_span(geo, n_pairs)))) return rc;
            d_cur = static_cast<const uint8_t *>(ctx->stage[0].ptr);
            if (prev) {
                if ((rc = upload(ctx, ctx->stage[1], prev, frame_span(geo, n_pairs)))) return rc;
                d_prev = static_cast<const uint8_t *>(ctx->stage[1].ptr);
            }
        }
        if ((rc = ensure(ctx, ctx->det_out_kps, (size_t)capacity * sizeof(mofreak_keypoint)))) return rc;
        if ((rc = ensure(ctx, ctx->det_out_offsets, (size_t)(n_pairs + 1) * sizeof(int64_t)))) return rc;
        d_kps = static_cast<mofreak_keypoint *>(ctx->det_out_kps.ptr);
        d_off = static_cast<int64_t *>(ctx->det_out_offsets.ptr);
        if (out_response) {
            if ((rc = ensure(ctx, ctx->det_out_resp, (size_t)capacity * sizeof(float)))) return rc;
            d_resp = static_cast<float *>(ctx->det_out_resp.ptr);
        }
        if (out_layer) {
            if ((rc = ensure(ctx, ctx->det_out_layer, (size_t)capacity * sizeof(int32_t)))) return rc;
            d_layer = static_cast<int32_t *>(ctx->det_out_layer.ptr);
        }
    }
    int64_t total = 0;
    int32_t st = 0;
    for (;;) {  // (again from the start, with room for four times the candidates, when a pair had more than the workspace's share)
    const int batch = det_batch(ctx, g, n_pairs);
    DetArgs a{};
    if ((rc = det_workspace(ctx, g, batch, a))) return rc;
    int64_t *running = static_cast<int64_t *>(ctx->det_rows.ptr);
    if (n_pairs == 0) HIP_TRY(ctx, hipMemsetAsync(running, 0, 16, ctx->stream));
    if (n_pairs == 0) HIP_TRY(ctx, hipMemsetAsync(d_off, 0, sizeof(int64_t), ctx->stream));
    a.threshold = threshold;
    a.safe_threshold = (int)(uint8_t)((float)threshold * 1.0f);  // safeThreshold_ = threshold_ * safetyFactor_ (brisk.cpp:58, 597)
    a.out_kps = d_kps;
    a.out_response = d_resp;
    a.out_layer = d_layer;
    a.out_capacity = capacity;
    a.out_offsets = d_off;
    ctx->det_maps_dirty = true;  // until the call has run to its end (every early return below leaves it set)
    ctx->det_diff = mofreak_ctx::DetDiff{};
    if (d_prev && n_pairs <= batch) ctx->det_diff = mofreak_ctx::DetDiff{a.img + g.L[0].off, g.plane_bytes, W, H, n_pairs};
    for (int p0 = 0; p0 < n_pairs; p0 += batch) {
        const int np = std::min(batch, n_pairs - p0);
        a.n_pairs = np;
        a.first_pair = p0;
        a.f = FrameArgs{d_cur + (int64_t)p0 * pair_stride, d_prev ? d_prev + (int64_t)p0 * pair_stride : nullptr, W, H, row_stride, pair_stride};
        // (the counters of the batch and, in front of them, the call's running total and status word before its first batch are
        // zeroed by the batch's first kernel; the touch and status maps are zero already (det_workspace) and are left zero by
        // the emission)
        int e = launch_det_pyramid(a, ctx->stream);
        if (!e) e = launch_det_corners(a, ctx->stream);
        if (!e) e = launch_det_keypoints(a, running, ctx->stream);
        if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("detector launch: ") + hipGetErrorString((hipError_t)e));
    }
    int64_t head[2] = {0, 0};  // the running total and, behind it, the status word
    HIP_TRY(ctx, hipMemcpyAsync(head, running, sizeof(head), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    total = head[0];
    st = (int32_t)(uint32_t)(uint64_t)head[1];
    ctx->det_maps_dirty = (st & (16 | 32 | 64)) != 0;  // (else) the emission of every batch has taken its bytes back
    if ((st & 4) && !(st & (16 | 32 | 64)) && det_cand_eff(ctx, g) < ctx->det_cand_cap) {
        ctx->det_cand_shift += 2;    // kept for the context's later calls
        ctx->det_maps_dirty = true;  // (candidates past the workspace's share were not followed: nothing is taken for granted)
        continue;
    }
    break;
    }
    if (n_out) *n_out = total;
    if (st & 64) return fail(ctx, MOFREAK_ERR_HIP, "detector: the bookkeeping maps were not clean when the call started (internal error; bounds-checking build)");
    if (st & 16) return fail(ctx, MOFREAK_ERR_HIP, "detector: a refinement walk left its staged window (internal error)");
    if (st & 32) return fail(ctx, MOFREAK_ERR_HIP, "detector: a chain of tied scores did not resolve within its pass budget (internal error)");
    if (st & 4) return fail(ctx, MOFREAK_ERR_CAPACITY, "more corner candidates in one pair than the detector reserved (mofreak_detect_set_capacity)");
    if (host) {
        const int64_t n_copy = std::min(total, capacity);
        HIP_TRY(ctx, hipMemcpy(out_offsets, d_off, (size_t)(n_pairs + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (n_copy) {
            HIP_TRY(ctx, hipMemcpy(out_kps, d_kps, (size_t)n_copy * sizeof(mofreak_keypoint), hipMemcpyDeviceToHost));
            if (out_response) HIP_TRY(ctx, hipMemcpy(out_response, d_resp, (size_t)n_copy * sizeof(float), hipMemcpyDeviceToHost));
            if (out_layer) HIP_TRY(ctx, hipMemcpy(out_layer, d_layer, (size_t)n_copy * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
    }
    if ((st & 8) || total > capacity) return fail(ctx, MOFREAK_ERR_CAPACITY, "more keypoints than out_kps holds");
    return MOFREAK_OK;
}

}  // extern "C"

extern "C" {

int mofreak_compute_stream(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H, int threshold, int octaves, mofreak_row *rows_out,
                           int64_t rows_capacity, int64_t *n_rows_out, int64_t *n_keypoints_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (n_rows_out) *n_rows_out = 0;
    if (n_keypoints_out) *n_keypoints_out = 0;
    if (T < 0 || rows_capacity < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count");
    const int gap = ctx->params.gap_for_frame_difference;
    const int n_pairs = T - gap;
    if (n_pairs <= 0) return MOFREAK_OK;
    if (!frames || (rows_capacity > 0 && !rows_out)) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const int64_t fsz = (int64_t)W * H;
    int rc;
    const uint8_t *d_frames = frames;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], frames, (size_t)T * fsz))) return rc;
        d_frames = static_cast<const uint8_t *>(ctx->stage[0].ptr);
    }
    mofreak_row *d_rows = rows_out;
    if (host) {
        if ((rc = ensure(ctx, ctx->stage[5], (size_t)std::max<int64_t>(rows_capacity, 1) * sizeof(mofreak_row)))) return rc;
        d_rows = static_cast<mofreak_row *>(ctx->stage[5].ptr);
    }
    // The stack in detector batches (as many pairs as its workspace takes at once: a few hundred full-HD pairs): detector,
    // then the descriptors of the batch's keypoints -- their integral images read |cur - prev| where the detector has just
    // left it -- then the batch's rows behind those of the batches before.
    DetGeom g;
    if ((rc = det_geometry(ctx, W, H, octaves, g))) return rc;
    const int batch = det_batch(ctx, g, n_pairs);
    const Geometry geo{W, H, W, fsz};
    int64_t total_rows = 0, total_kp = 0;
    bool rows_overflow = false;
    struct DiffScope {  // the planes are this call's: no later extract may take them for its own pairs' difference images
        mofreak_ctx *c;
        ~DiffScope()
        {
            c->use_det_diff = false;
            c->det_diff = mofreak_ctx::DetDiff{};
        }
    } scope{ctx};
    for (int p0 = 0; p0 < n_pairs; p0 += batch) {
        const int np = std::min(batch, n_pairs - p0);
        const uint8_t *cur = d_frames + (int64_t)(gap + p0) * fsz, *prev = d_frames + (int64_t)p0 * fsz;
        // keypoints: room for what the last call needed (at least 4096 per pair), grown once if this stream has more
        int64_t n_kp = 0;
        for (int attempt = 0;; ++attempt) {
            const int64_t cap = std::max<int64_t>(ctx->det_kp_capacity, (int64_t)4096 * np);
            if ((rc = ensure(ctx, ctx->det_out_kps, (size_t)cap * sizeof(mofreak_keypoint)))) return rc;
            if ((rc = ensure(ctx, ctx->det_out_offsets, (size_t)(np + 1) * sizeof(int64_t)))) return rc;
            rc = mofreak_detect_pairs(ctx, cur, prev, W, H, W, fsz, np, threshold, octaves, static_cast<mofreak_keypoint *>(ctx->det_out_kps.ptr), cap,
                                      static_cast<int64_t *>(ctx->det_out_offsets.ptr), nullptr, nullptr, &n_kp, MOFREAK_MEM_DEVICE);
            if (rc == MOFREAK_ERR_CAPACITY && n_kp > cap && attempt == 0) {
                ctx->det_kp_capacity = n_kp + n_kp / 8;
                continue;
            }
            if (rc) return rc;
            break;
        }
        total_kp += n_kp;
        if (n_keypoints_out) *n_keypoints_out = total_kp;
        if (n_kp == 0) continue;
        const mofreak_keypoint *kps = static_cast<const mofreak_keypoint *>(ctx->det_out_kps.ptr);
        const int64_t *d_off = static_cast<const int64_t *>(ctx->det_out_offsets.ptr);
        std::vector<int64_t> h_off;
        if ((rc = fetch_offsets(ctx, d_off, np, false, h_off))) return rc;
        if ((rc = ensure(ctx, ctx->scratch_desc, (size_t)n_kp * 16))) return rc;
        if ((rc = ensure(ctx, ctx->scratch_valid, (size_t)n_kp))) return rc;
        uint8_t *desc = static_cast<uint8_t *>(ctx->scratch_desc.ptr), *valid = static_cast<uint8_t *>(ctx->scratch_valid.ptr);
        ctx->use_det_diff = true;
        rc = extract_device(ctx, cur, prev, geo, np, kps, d_off, h_off.data(), n_kp, desc, valid, nullptr, nullptr);
        ctx->use_det_diff = false;
        if (rc) return rc;
        int64_t got = 0;
        const int64_t room = rows_overflow ? 0 : rows_capacity - total_rows;
        // pair p of the batch is frame gap + p0 + p of the stack: labelled gap - 1 + p0 + p (:401, :488)
        rc = compact_device(ctx, kps, d_off, n_kp, np, n_kp, gap - 1 + p0, desc, valid, d_rows + (rows_overflow ? 0 : total_rows), room, &got);
        total_rows += got;
        if (n_rows_out) *n_rows_out = total_rows;
        if (rc == MOFREAK_ERR_CAPACITY) {
            rows_overflow = true;  // no further row leaves the kernels; the counting goes on: the caller learns the size a retry needs
            continue;
        }
        if (rc) return rc;
    }
    if (rows_overflow) return fail(ctx, MOFREAK_ERR_CAPACITY, "rows_out too small: need " + std::to_string(total_rows));
    if (host && total_rows) HIP_TRY(ctx, hipMemcpy(rows_out, d_rows, (size_t)total_rows * sizeof(mofreak_row), hipMemcpyDeviceToHost));
    return MOFREAK_OK;
}

struct mofreak_stream {
    mofreak_ctx *ctx;
    int W, H, use_detector, threshold, octaves;
    int64_t n_seen;
    DeviceBuffer ring, in, kps, offs, desc, valid, rows;
};

int mofreak_stream_open(mofreak_ctx *ctx, int W, int H, int use_detector, int threshold, int octaves, mofreak_stream **out)
{
    if (!ctx || !out) return MOFREAK_ERR_BAD_ARG;
    *out = nullptr;
    if (W <= 0 || H <= 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "W, H must be positive");
    if (use_detector) {
        DetGeom g;
        const int rc = det_geometry(ctx, W, H, octaves, g);
        if (rc) return rc;
        if (threshold < 1 || threshold > 255) return fail(ctx, MOFREAK_ERR_BAD_ARG, "threshold must be in 1..255");
    }
    NEED_DEVICE(ctx);
    mofreak_stream *s = new (std::nothrow) mofreak_stream();
    if (!s) return fail(ctx, MOFREAK_ERR_OOM, "out of host memory");
    s->ctx = ctx;
    s->W = W;
    s->H = H;
    s->use_detector = use_detector;
    s->threshold = threshold;
    s->octaves = octaves;
    s->n_seen = 0;
    const int rc = ensure(ctx, s->ring, (size_t)(ctx->params.gap_for_frame_difference + 1) * W * H);
    if (rc) {
        delete s;
        return rc;
    }
    *out = s;
    return MOFREAK_OK;
}

int64_t mofreak_stream_frames(const mofreak_stream *s) { return s ? s->n_seen : 0; }

void mofreak_stream_close(mofreak_stream *s)
{
    if (!s) return;
    if (s->ctx && s->ctx->device >= 0) {
        (void)hipSetDevice(s->ctx->device);
        if (s->ctx->stream) (void)hipStreamSynchronize(s->ctx->stream);
    }
    for (DeviceBuffer *b : {&s->ring, &s->in, &s->kps, &s->offs, &s->desc, &s->valid, &s->rows}) release(*b);
    delete s;
}

int mofreak_stream_push(mofreak_stream *s, const uint8_t *frame, int channels, int64_t row_stride, const mofreak_keypoint *kps, int64_t n_kp,
                        mofreak_row *rows_out, int64_t rows_capacity, int64_t *n_rows_out, unsigned flags)
{
    if (!s || !s->ctx) return MOFREAK_ERR_BAD_ARG;
    mofreak_ctx *ctx = s->ctx;
    if (n_rows_out) *n_rows_out = 0;
    if (!frame) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null frame");
    if (channels != 1 && channels != 3) return fail(ctx, MOFREAK_ERR_BAD_ARG, "channels must be 1 (gray) or 3 (BGR)");
    if (row_stride < (int64_t)channels * s->W) return fail(ctx, MOFREAK_ERR_BAD_ARG, "row_stride too small");
    if (rows_capacity < 0 || n_kp < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count");
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const int W = s->W, H = s->H, gap = ctx->params.gap_for_frame_difference, slots = gap + 1;
    const int64_t fsz = (int64_t)W * H;
    uint8_t *ring = static_cast<uint8_t *>(s->ring.ptr);
    uint8_t *slot = ring + (s->n_seen % slots) * fsz;
    int rc;
    // the frame into its ring slot, gray
    if (channels == 1) {
        HIP_TRY(ctx, hipMemcpy2DAsync(slot, (size_t)W, frame, (size_t)row_stride, (size_t)W, (size_t)H,
                                      host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        const uint8_t *d_in = frame;
        if (host) {
            const size_t span = (size_t)((int64_t)(H - 1) * row_stride + (int64_t)3 * W);
            if ((rc = upload(ctx, s->in, frame, span))) return rc;
            d_in = static_cast<const uint8_t *>(s->in.ptr);
        }
        if ((rc = mofreak_bgr_to_gray(ctx, d_in, W, H, row_stride, 0, 1, slot, MOFREAK_MEM_DEVICE))) return rc;
    }
    const int64_t index = s->n_seen++;
    if (index < gap) {  // the first gap frames only fill the queue (:391-399)
        if (host) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the caller may reuse its frame buffer
        return MOFREAK_OK;
    }
    const uint8_t *cur = slot, *prev = ring + ((index - gap) % slots) * fsz;
    const mofreak_keypoint *d_kps = kps;
    if (s->use_detector) {
        if ((rc = ensure(ctx, s->offs, 2 * sizeof(int64_t)))) return rc;
        for (int attempt = 0;; ++attempt) {
            const int64_t cap = std::max<int64_t>((int64_t)(s->kps.bytes / sizeof(mofreak_keypoint)), 16384);
            if ((rc = ensure(ctx, s->kps, (size_t)cap * sizeof(mofreak_keypoint)))) return rc;
            rc = mofreak_detect_pairs(ctx, cur, prev, W, H, W, fsz, 1, s->threshold, s->octaves, static_cast<mofreak_keypoint *>(s->kps.ptr), cap,
                                      static_cast<int64_t *>(s->offs.ptr), nullptr, nullptr, &n_kp, MOFREAK_MEM_DEVICE);
            if (rc == MOFREAK_ERR_CAPACITY && n_kp > cap && attempt == 0) {
                if ((rc = ensure(ctx, s->kps, (size_t)(n_kp + n_kp / 8) * sizeof(mofreak_keypoint)))) return rc;
                continue;
            }
            if (rc) return rc;
            break;
        }
        d_kps = static_cast<const mofreak_keypoint *>(s->kps.ptr);
    } else if (n_kp > 0) {
        if (!kps) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null keypoint pointer");
        if (host) {
            if ((rc = upload(ctx, s->kps, kps, (size_t)n_kp * sizeof(mofreak_keypoint)))) return rc;
            d_kps = static_cast<const mofreak_keypoint *>(s->kps.ptr);
        }
    }
    if (n_kp == 0) {
        if (host) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return MOFREAK_OK;
    }
    if ((rc = ensure(ctx, s->desc, (size_t)n_kp * 16))) return rc;
    if ((rc = ensure(ctx, s->valid, (size_t)n_kp))) return rc;
    if ((rc = mofreak_extract_pairs(ctx, cur, prev, W, H, W, fsz, 1, d_kps, nullptr, n_kp, static_cast<uint8_t *>(s->desc.ptr),
                                    static_cast<uint8_t *>(s->valid.ptr), MOFREAK_MEM_DEVICE)))
        return rc;
    mofreak_row *d_rows = rows_out;
    if (host) {
        if ((rc = ensure(ctx, s->rows, (size_t)std::max<int64_t>(rows_capacity, 1) * sizeof(mofreak_row)))) return rc;
        d_rows = static_cast<mofreak_row *>(s->rows.ptr);
    }
    int64_t total = 0;
    // frame `index` carries the label index - 1: the first processed frame (index gap) is gap - 1 (:401, :488)
    rc = mofreak_compact_rows(ctx, d_kps, nullptr, n_kp, 1, (int)(index - 1), static_cast<const uint8_t *>(s->desc.ptr),
                              static_cast<const uint8_t *>(s->valid.ptr), d_rows, rows_capacity, &total, MOFREAK_MEM_DEVICE);
    if (n_rows_out) *n_rows_out = total;
    if (rc) return rc;
    if (host && total) HIP_TRY(ctx, hipMemcpy(rows_out, d_rows, (size_t)total * sizeof(mofreak_row), hipMemcpyDeviceToHost));
    return MOFREAK_OK;
}

int mofreak_stream_push_frames(mofreak_stream *s, const uint8_t *frames, int n_frames, int chunk_frames, const mofreak_keypoint *kps, int64_t n_kp,
                               mofreak_row *rows_out, int64_t rows_capacity, int64_t *n_rows_out)
{
    if (!s || !s->ctx) return MOFREAK_ERR_BAD_ARG;
    mofreak_ctx *ctx = s->ctx;
    if (n_rows_out) *n_rows_out = 0;
    if (n_frames < 0 || n_kp < 0 || rows_capacity < 0) return fail(ctx, MOFREAK_ERR_BAD_ARG, "negative count");
    if (n_frames == 0) return MOFREAK_OK;
    if (!frames) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null frames");
    if (!s->use_detector && n_kp > 0 && !kps) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null keypoint pointer");
    if (rows_capacity > 0 && !rows_out) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null rows_out");
    NEED_DEVICE(ctx);
    const int W = s->W, H = s->H, gap = ctx->params.gap_for_frame_difference, slots = gap + 1;
    const int64_t fsz = (int64_t)W * H;
    uint8_t *ring = static_cast<uint8_t *>(s->ring.ptr);
    const int tail = (int)std::min<int64_t>(s->n_seen, gap);  // frames of the stream still in the device ring that new pairs look back to
    const int64_t first_new = s->n_seen;
    int rc = MOFREAK_OK;
    ClipPieces pc;
    ClipDetector det;  // a detector stream: the keypoints of every pair are found on the device, window by window (kps is ignored)
    det.threshold = s->threshold;
    det.octaves = s->octaves;
    if ((int64_t)tail + n_frames > gap && (n_kp > 0 || s->use_detector)) {
        // one clip in tail + 1 pieces: the ring's frames (one slot each, oldest first), then the caller's chunk
        std::vector<const uint8_t *> ptr;
        std::vector<int32_t> len;
        std::vector<uint8_t> joined;
        std::vector<int64_t> base;
        for (int j = 0; j < tail; ++j) {
            ptr.push_back(ring + ((first_new - tail + j) % slots) * fsz);
            len.push_back(1);
        }
        ptr.push_back(frames);
        len.push_back(n_frames);
        joined.assign(ptr.size(), 1);
        joined[0] = 0;
        base.assign(ptr.size(), first_new - tail);  // the clip's first frame is frame number first_new - tail of the stream (:401, :488)
        pc.joined = joined.data();
        pc.label_base = base.data();
        rc = extract_clips_impl(ctx, ptr.data(), len.data(), (int)ptr.size(), W, H, chunk_frames, kps, n_kp, rows_out, rows_capacity, nullptr, n_rows_out, false, &pc,
                                s->use_detector ? &det : nullptr);
        if (rc != MOFREAK_OK && rc != MOFREAK_ERR_CAPACITY) return rc;  // (rows that do not fit: the frames are consumed all the same, like mofreak_stream_push)
    }
    // the stream's last gap frames into their ring slots: from the pipeline's last window where they already are on the
    // device, else (a chunk too short for a pair, or no keypoints) from the caller's memory
    const int keep = std::min(gap, n_frames);
    for (int j = 0; j < keep; ++j) {
        const int64_t idx = first_new + n_frames - keep + j;
        uint8_t *dst = ring + (idx % slots) * fsz;
        if (pc.last_window_slot >= 0) {
            const int64_t q = tail + (idx - first_new);  // its place in the clip
            const uint8_t *src = static_cast<const uint8_t *>(ctx->pipe.d_frames[pc.last_window_slot].ptr) + (q - pc.last_window_first) * fsz;
            HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)fsz, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(dst, frames + (idx - first_new) * fsz, (size_t)fsz, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the caller may reuse its frame buffer
    s->n_seen += n_frames;
    return rc;
}

int mofreak_brisk_pyramid(mofreak_ctx *ctx, const uint8_t *img, int W, int H, int64_t row_stride, int octaves, uint8_t *layers_out,
                          uint8_t *scores_out, int32_t *dims_out, float *scale_offset_out, int *n_layers_out, unsigned flags)
{
    if (!ctx) return MOFREAK_ERR_BAD_ARG;
    if (!img) return fail(ctx, MOFREAK_ERR_BAD_ARG, "null pointer");
    if (row_stride < W) return fail(ctx, MOFREAK_ERR_BAD_ARG, "row_stride < W");
    DetGeom g;
    int rc = det_geometry(ctx, W, H, octaves, g);
    if (rc) return rc;
    if (n_layers_out) *n_layers_out = g.n_layers;
    for (int i = 0; i < g.n_layers; ++i) {
        if (dims_out) {
            dims_out[2 * i] = g.L[i].w;
            dims_out[2 * i + 1] = g.L[i].h;
        }
        if (scale_offset_out) {
            scale_offset_out[2 * i] = g.L[i].scale;
            scale_offset_out[2 * i + 1] = g.L[i].offset;
        }
    }
    if (!layers_out && !scores_out) return MOFREAK_OK;
    NEED_DEVICE(ctx);
    const bool host = (flags & MOFREAK_MEM_HOST) != 0;
    const Geometry geo{W, H, row_stride, 0};
    const uint8_t *d_img = img;
    if (host) {
        if ((rc = upload(ctx, ctx->stage[0], img, frame_span(geo, 1)))) return rc;
        d_img = static_cast<const uint8_t *>(ctx->stage[0].ptr);
    }
    DetArgs a{};
    if ((rc = det_workspace(ctx, g, 1, a))) return rc;
    a.n_pairs = 1;
    a.threshold = a.safe_threshold = 255;
    a.f = FrameArgs{d_img, nullptr, W, H, row_stride, 0};
    HIP_TRY(ctx, hipMemsetAsync(a.row_count, 0, (size_t)(g.total_rows + 1) * sizeof(int32_t), ctx->stream));
    int e = launch_det_pyramid(a, ctx->stream);
    if (!e && scores_out) e = launch_det_scores(a, ctx->stream);
    if (e) return fail(ctx, MOFREAK_ERR_HIP, std::string("detector launch: ") + hipGetErrorString((hipError_t)e));
    const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    size_t o = 0;
    for (int i = 0; i < g.n_layers; ++i) {
        const size_t n = (size_t)g.L[i].w * g.L[i].h;
        if (n && layers_out) HIP_TRY(ctx, hipMemcpyAsync(layers_out + o, a.img + g.L[i].off, n, kind, ctx->stream));
        if (n && scores_out) HIP_TRY(ctx, hipMemcpyAsync(scores_out + o, a.score + g.L[i].off, n, kind, ctx->stream));
        o += n;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOFREAK_OK;
}

}  // extern "C"
