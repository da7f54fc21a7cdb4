// Argument blocks shared by the host launcher (capi.cpp) and the gfx950 kernels (kernels.hip).
#pragma once

#include <cstdint>

#include "../../include/mofreak_hip.h"
#include "tables.h"

namespace mofreak {

// Small per-context tables; every workgroup copies them from global memory into LDS once.
struct SmallTables {
    int32_t pattern_sizes[kNbScales];
    float scale_thresholds[kNbScales];
    OrientPair orient[kNbOrientPairs];
    uint8_t bit_pair_i[64];
    uint8_t bit_pair_j[64];
    int32_t fixed_scale_index;
    int32_t orientation_normalized;
    int32_t scale_normalized;
    int32_t bit_mode;
    int32_t mip_theta;
    int32_t mip_n_cur, mip_n_prev;  // entries of the two lists below
    int32_t pad[1];
    // which of the 43 pattern points the orientation pairs / the 64 description pairs read (bit = point): the gather path leaves the
    // others' boxes alone (the default tables: all but point 42; 39 points)
    unsigned long long need_orient, need_bits;
    // the 19x19 positions the MIP reads of the current / previous buffer (ascending): the gather path resamples only these
    uint16_t mip_cur[64];
    uint16_t mip_prev[256];
};

// Integral image of one chunk of pairs, as the kernels lay it out in HBM:
// (H+1) rows of `pitch` int32 per pair; logical column c (0..W) lives at physical column c+3, so the
// 4-pixel group starting at pixel x = 4g (logical columns x+1..x+4) is one aligned 16-byte store.
constexpr int kIntegralColOffset = 3;
constexpr int kBandRows = 4;   // rows of the banded integral kernels' LDS buffer: a wave per row; 31 KB per workgroup at 1080p, five per CU
constexpr int kBandGroup = 4;  // such slabs per band: a workgroup takes them one after the other; band totals are per band
constexpr int kBandColItersMax = 10;  // 1024 columns per iteration of the column pass: pitch <= 10240 (what 160 KB of LDS allow)

struct FrameArgs {
    const uint8_t *cur;
    const uint8_t *prev;
    int32_t W, H;
    int64_t row_stride;
    int64_t pair_stride;
};

struct IntegralArgs {
    const int32_t *gate;  // optional device word: the kernels return at once when *gate == 0 (no slow keypoints)
    FrameArgs f;          // cur/prev already point at the first pair of the chunk
    int32_t *integral;    // [n_pairs][H+1][pitch]
    int32_t *band_totals; // [n_pairs][n_bands][pitch]
    int32_t pitch;        // int32 per integral row, multiple of 4
    int32_t n_bands;
    int32_t n_pairs;
};

constexpr int kBinHeaderInts = 64;  // slow_count at int 0, max_ps at int 16 of the binning pass's counter buffer

struct DescribeArgs {
    FrameArgs f;               // cur/prev point at the first pair of the chunk
    const int32_t *integral;   // chunk-local
    int32_t pitch;
    int32_t n_pairs;           // pairs in this chunk
    const PatternPoint *lut;
    const ResizeTap *resize;
    const SmallTables *small;
    const ThetaBound *theta;
    const mofreak_keypoint *kps;
    const int64_t *kp_offsets; // device CSR offsets of the WHOLE call (nullptr: shared list)
    int64_t n_kp;              // shared list length (kp_offsets == nullptr)
    int64_t first_pair;        // index of the chunk's first pair within the call
    int64_t item_base;         // CSR: kp_offsets[first_pair]; shared: first_pair * n_kp
    int64_t n_items;           // keypoint instances in this chunk
    uint8_t *out_desc;         // whole-call output arrays
    uint8_t *out_valid;
    int32_t *out_info;         // optional: 4 x int32 per instance
    uint8_t *out_roi19;        // optional: 722 bytes per instance
    int32_t *status;           // device status word (bit 0: ROI left the image, bit 1: ROI side too large)
    // gather ("slow") path behind the tile kernel: instances = the keypoints the binning pass could not give to a
    // tile (slow_list, *slow_count of them), in every pair of the chunk (shared list) or in their own pair (CSR)
    const int32_t *slow_list;
    const int32_t *slow_count;
    const int32_t *band_start;  // CSR: where each (pair, band of rows) starts in the list (BinArgs::tile_start + n_keys): a chunk of pairs owns one piece of it
    int32_t bands_per_pair;
    int64_t n_pairs_total;     // pairs of the whole call (CSR pair search in slow-list mode)
};

// Binning output: the keypoint itself next to its index, grouped by tile (one dwordx4 per entry).
struct SortedKp {
    float x, y;
    uint32_t packed;  // ROI side L = ceil(size) | (int)size/2 << 8 | FREAK scale index << 16 (all derived once, at binning)
    int32_t g;
};

// Binning of keypoints by image tile for the fused tile kernel (tile_kernel.hip).
// key = tile (shared keypoint list) or pair * n_tiles + tile (CSR).
struct BinArgs {
    const mofreak_keypoint *kps;
    const int64_t *kp_offsets;  // nullptr: shared list
    int64_t n_kp;
    int32_t n_pairs;
    int32_t W, H;
    int32_t tiles_x, tiles_y;
    int32_t force_slow;         // every valid keypoint goes to the gather path (tests, roi19 dumps)
    const SmallTables *small;
    int32_t *kp_key;            // [n_kp] key >= 0: tile key; -1: erased; <= -3: gather path, band key b (pair * tiles_y + tile row) as -3 - b
    uint8_t *kp_scale;          // [n_kp] FREAK scale index (0 for keypoints that fail the size / finiteness tests)
    int32_t *tile_start;        // [n_keys + n_bkeys + 1] counts, then exclusive starts: the tiles' keys, behind them the gather path's band keys
    int32_t *tile_cursor;       // [n_keys + n_bkeys]
    uint32_t *tile_lmin_c;      // [n_keys] smallest (as its complement ~L) / largest ROI side among the tile's keypoints: equal in
    uint32_t *tile_lmax;        // the usual case, and then the tile kernel can fetch its sample table before it has seen a keypoint
    size_t counter_bytes;       // slow_count .. the end of tile_lmax are one buffer (kBinHeaderInts in front of tile_start): one fill
    SortedKp *sorted_kp;        // [n_kp] keypoints grouped by key
    int32_t *slow_list;         // [n_kp] the gather path's keypoints, band after band (tile_start[n_keys + b] - tile_start[n_keys] is where band b starts)
    int32_t *slow_count;        // [1]
    int32_t *max_ps;            // [1] largest patternSizes[] among the tile-path keypoints: sizes the tile kernel's halo
    int32_t *wg_slow, *wg_maxps;  // [ceil(n_kp / 256)] pass 1's per-workgroup gather-path count / largest tile-path pattern
    uint8_t *out_desc;          // erased keypoints are finalised by the binning pass (zeros, valid = 0)
    uint8_t *out_valid;
    int32_t *out_info;
    int64_t n_keys;             // tile keys: tiles_x * tiles_y (shared list) or that per pair (CSR)
    int64_t n_bkeys;            // band keys of the gather path: tiles_y, or that per pair
};

// Per-lane constants of the tile kernel (64 entries, host-built): one lane's share of the tables of stage 3.
struct TileLane {
    uint16_t task[4];        // box-mean tasks of a group of four keypoints, passes 0..2: keypoint-in-group | point << 8
    uint8_t pair_i, pair_j;  // the description pair of descriptor bit `lane`
    uint8_t opi[3], opj[3];  // orientation pairs (lane & 15) + 16 k, k < 3 (the 45 pairs over 16 lanes; unused ones: 0, 0)
    float owx[3], owy[3];    // their weights / 2048 (exact in float); 0 for unused slots
    uint32_t pad[2];
};
static_assert(sizeof(TileLane) == 48, "three dwordx4 per lane");

struct TileArgs {
    FrameArgs f;
    int32_t n_pairs;
    int32_t tiles_x, tiles_y;
    const PatternPoint *lut;
    const BoxInt *lut_int;      // boxes of lut for keypoints at integer coordinates
    float box_margin;           // BoxInt entries with a larger margin are exact for this frame size (2.0: none)
    const SmallTables *small;
    const ThetaBound *theta;
    const MipSample *mip_samples;
    const uint16_t *mip_pos;
    const TileLane *lanes;
    int32_t mip_n_cur, mip_n, mip_stride;
    const int64_t *kp_offsets;  // nullptr: shared list (tile lists are the same for every pair)
    int64_t n_kp;
    const int32_t *tile_start;
    const uint32_t *tile_lmin_c, *tile_lmax;  // smallest ROI side stored complemented (BinArgs)
    const SortedKp *sorted_kp;
    const int32_t *max_ps;      // device word written by the binning pass (see BinArgs)
    int32_t *status;            // device status word (debug builds: bit 6 = an access left its bounds)
    int64_t out_items;          // descriptors the output arrays of this launch hold (debug builds check stores against it)
    uint8_t *out_desc;
    uint8_t *out_valid;
    int32_t *out_info;          // optional
    unsigned long long *stamps; // optional [kTileStampSlots]: diagnostic per-phase tick sums (tile_kernel<true>)
};
constexpr int kTileStampSlots = 32;

struct CompactArgs {
    const mofreak_keypoint *kps;
    const int64_t *kp_offsets;
    int64_t n_kp;
    int64_t n_items;
    int32_t n_pairs;
    int32_t first_frame_number;
    const uint8_t *desc;
    const uint8_t *valid;
    mofreak_row *rows;
    int64_t capacity;
    int64_t *block_offsets;  // [n_blocks + 1]; last = total
    int32_t n_blocks;
    // Many clips in one call (mofreak_extract_clips; shared keypoint list only): pair p carries frame number pair_label[p]
    // instead of first_frame_number + p, and a pair whose label is negative (its two frames belong to different clips)
    // yields no rows.  pair_rows[p] (zeroed by the caller) receives the number of rows pair p produced.  Both may be null.
    const int32_t *pair_label = nullptr;
    int32_t *pair_rows = nullptr;
    // Rows appended behind the rows of earlier launches without a host round trip in between (the pipelined frame loop):
    // *row_base (device) is added to every row position of this launch and advanced by its total.  May be null.
    int64_t *row_base = nullptr;
};

constexpr int kCompactItemsPerBlock = 1024;

// ---- keypoint detector (detect_kernel.hip): BRISK scale space over |cur - prev|
constexpr int kDetMaxLayers = 8;
struct DetLayer {
    int32_t w, h;
    int64_t off;       // byte offset of the layer inside one pair's plane
    float scale, offset;
    int32_t row_base;  // index of the layer's first row in the pair's concatenated row list
    int32_t pad;
};
struct DetGeom {
    int32_t n_layers, total_rows;
    int64_t plane_bytes;  // one pair, all layers
    DetLayer L[kDetMaxLayers];
    // the corner kernel's 64 x 64-pixel tiles: tiles per row of a layer, first tile of a layer in the pair's tile list
    // (tile_start[n_layers] = tiles per pair); the hit masks -- one 64-bit word per tile row and layer row: bit b of word
    // (y, tx) = pixel (64 tx + b, y) is a detected corner -- start at word mask_off[l] of the pair's mask_words words
    int32_t tiles_x[kDetMaxLayers], tile_start[kDetMaxLayers + 1];
    int64_t mask_off[kDetMaxLayers], mask_words;
    // the candidate kernel: a wave takes cand_rows_per_wave[l] whole rows of layer l (as many as give it <= 64 mask words; 1
    // for wider layers); cand_group_start[l] = first such group of layer l, [n_layers] = groups per pair
    int32_t cand_rows_per_wave[kDetMaxLayers], cand_group_start[kDetMaxLayers + 1];
};
constexpr int kDetTileW = 64, kDetTileH = 64;
// candidate flags
enum : uint8_t { kDetNotMax = 0, kDetMax = 1, kDetTie = 2 };
// status-map values (one byte per pixel, written only by the thread that owns that candidate)
enum : uint8_t { kStNone = 0, kStPending = 1, kStReached = 2, kStDone = 3 };
struct DetResult {
    float x, y, size, response;
};
struct DetTie {
    int32_t cand;   // candidate index within the pair
    uint32_t xy;    // x | y << 16 in its layer
};

struct DetArgs {
    DetGeom g;          // host copy for the launchers; device code reads *dg (indexing a by-value kernel argument with a
    const DetGeom *dg;  // per-thread layer number would spill the whole argument block to scratch)
    FrameArgs f;  // cur / prev of the batch's first pair (prev == nullptr: cur already is the image to search)
    int32_t n_pairs;
    int32_t threshold, safe_threshold;
    int32_t fp_x87;     // 1: float expressions as the reference's x87 build evaluates them (mofreak_params.brisk_fp_model)
    // [n_pairs][plane_bytes] each.  score: the reference's score cache as far as anyone reads it -- the corner score where a
    // pixel is a detected corner (>= safe_threshold), elsewhere 0 or, once a refinement walk has computed it, the cell's
    // true score (< safe_threshold; meaningful only where touch / status say the reference would have cached it).
    // touch / status are all zero between calls: whoever sets a byte clears it again (det_emit_scatter_kernel).
    uint8_t *img, *score, *touch, *status;
    unsigned long long *hit_mask;           // [n_pairs][mask_words]: DetGeom::mask_off
    int32_t *row_count;                     // [n_pairs][total_rows + 1]: counts, then exclusive offsets
    int32_t cand_cap;                       // per pair
    uint32_t *cand_xy;                      // [n_pairs][cand_cap]  x | y << 16
    uint8_t *cand_flag;                     // [n_pairs][cand_cap]
    uint8_t *cand_emit;                     // [n_pairs][cand_cap]
    uint8_t *cand_spec;                     // [n_pairs][cand_cap] ties: emit / reached bits of the refinement run ahead of the decision
    unsigned long long *cand_asked;         // [n_pairs][cand_cap] cells a walk asked for in the layer above (bit mask over a 6 x 6 window; 0: none)
    uint32_t *cand_win;                     // [n_pairs][cand_cap] that window's origin, x | y << 16
    // refinement: chunk c of 512 candidates has walk_count[c] walkers (maxima and ties); walker k of the chunk is candidate
    // walk_list[512 c + k] and owns the 64-byte record cand_cells[512 c + k] (the cells its walks can read)
    int32_t *walk_list;                     // [n_pairs][cand_cap]
    int32_t *walk_count;                    // [n_pairs][walk_chunks]
    int32_t walk_chunks;
    uint8_t *cand_cells;                    // [n_pairs][cand_cap][64]
    // the ties among a chunk's walkers, in candidate order: tie k of chunk c is tie_list[512 c + k]
    DetTie *tie_list;                       // [n_pairs][cand_cap]
    int32_t *tie_count;                     // [n_pairs][walk_chunks]
    DetResult *cand_res;                    // [n_pairs][cand_cap]
    int32_t *layer_start;                   // [n_pairs][kDetMaxLayers + 1]
    int32_t *emit_count;                    // [n_pairs]
    int32_t *emit_chunks;                   // [n_pairs][emit_chunk_cap]: emitted candidates per chunk of 1024
    int32_t emit_chunk_cap;
    int64_t *emit_offsets;                  // [n_pairs + 1], relative to out_base
    mofreak_keypoint *out_kps;              // whole-call outputs
    float *out_response;                    // optional
    int32_t *out_layer;                     // optional
    int64_t out_capacity, out_base;
    int64_t *out_offsets;                   // whole-call CSR offsets; this batch fills [first_pair .. first_pair + n_pairs]
    int64_t first_pair;
    int32_t *status_word;                   // bit 2: more candidates than cand_cap; bit 3: more keypoints than out_capacity; bit 4: internal (a walk left its window); bit 5: internal (ties); bit 6: the bounds-checking build found the bookkeeping maps not clean at the start
};

// launchers (kernels.hip); all asynchronous on `stream`, return a hipError_t value as int
int launch_integral(const IntegralArgs &a, void *stream);
int launch_describe(const DescribeArgs &a, int n_blocks, void *stream);
int launch_mip19(const uint8_t *cur19, const uint8_t *prev19, int64_t n, int mip_theta, uint8_t *out, void *stream);
int launch_theta(const ThetaBound *tb, const int32_t *dirs, int64_t n, int32_t *out, void *stream);
int launch_compact(const CompactArgs &a, void *stream);
// .mofreak text on the device (format_kernel.hip): lengths + scan (+ segment offsets), then the text
size_t format_workspace_bytes(int64_t n_rows, int n_segments);
int launch_format_measure(const mofreak_row *d_rows, int64_t n_rows, void *ws, const int64_t *d_row_starts, int n_segments, int32_t *d_status_bits,
                          uint64_t **d_total_out, uint64_t **d_seg_off_out, void *stream);
int launch_format_write(const mofreak_row *d_rows, int64_t n_rows, void *ws, char *d_text, uint64_t cap, void *stream);
int launch_bin(const BinArgs &a, void *stream);   // zeroes the counters, classifies, scans, scatters
int launch_tile(const TileArgs &a, void *stream);
int launch_bgr2gray(const uint8_t *bgr, int W, int H, int64_t row_stride, int64_t frame_stride, int n_frames, uint8_t *gray,
                    void *stream);
size_t bow_expanded_bytes(int n_codewords);  // workspace for the codebook expanded to the matrix instruction's operand bytes
int launch_bow_assign(const uint8_t *desc, const uint8_t *valid, int64_t n, const uint8_t *codebook, int n_codewords,
                      int32_t *out_index, unsigned int *counts, int n_cus, void *expanded_ws, void *stream);
int launch_bow_normalize(const unsigned int *counts, int n_codewords, float *hist, int32_t *success, void *stream);
int launch_unpack_integral(const int32_t *src, int pitch, int W, int H, int n_pairs, int32_t *dst, void *stream);
int launch_det_pyramid(const DetArgs &a, void *stream);   // difference image + the resampled layers
int launch_det_corners(const DetArgs &a, void *stream);   // corner scores where a pixel is a corner (0 elsewhere), hit masks, per-row detection counts
int launch_det_scores(const DetArgs &a, void *stream);    // component entry point only: the dense corner score of every pixel
int launch_det_keypoints(const DetArgs &a, int64_t *running, void *stream);  // candidates, maxima, refinement, ordered emission

}  // namespace mofreak
