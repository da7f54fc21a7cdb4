// Host-built lookup tables of the MoFREAK path (uploaded once per context).
//
// Everything here is computed on the HOST in the evaluation order of the code it replaces, so the
// device only ever does integer work and IEEE float/double add/mul on these values:
//   - the cv::FREAK pattern LUT, patternSizes, orientation weights and description pairs
//     (OpenCV 2.4.x features2d/src/freak.cpp buildPattern(); constructed per frame by the reference at
//     MoFREAKUtilities.cpp:427),
//   - thresholds that turn a keypoint size into FREAK's scale index without calling log() on the device,
//   - the fixed-point coefficient tables of cv::resize(ROI -> 19x19, INTER_LINEAR) per ROI side L
//     (imgproc/src/imgwarp.cpp; called at MoFREAKUtilities.cpp:303-304).
#pragma once

#include <cstdint>
#include <vector>

namespace mofreak {

constexpr int kNbScales = 64;
constexpr int kNbOrientation = 256;
constexpr int kNbPoints = 43;
constexpr int kNbPairs = 512;
constexpr int kNbOrientPairs = 45;
constexpr int kSmallestKpSize = 7;
constexpr int kPatch = 19;         // MoFREAKUtilities.cpp:300
constexpr int kMaxRoiSide = 2048;  // largest ceil(keypoint size) the resize tables cover

struct PatternPoint {  // 16 bytes on the device so a lane fetches one point with a single dwordx4 load
    float x, y, sigma;
    // Tile kernel (integral kept modulo 2^16): rows of this point's sampling box that are certain to hold at most 257
    // pixels, floor(257 / (floor(2 sigma) + 3)) -- the box is at most floor(2 sigma) + 2 pixels wide -- and at least 1.
    int32_t rows_per_slice;
};

// The sampling box of one pattern point for a keypoint at INTEGER coordinates, where the box is a fixed offset from the
// keypoint: meanIntensity's corners are int(fl(fl(P.x + kx) - r) + 0.5) etc., and for an integral kx the two float
// roundings move the argument by at most one ulp of the coordinate's binade, so the corner is kx + floor(P.x - r + 0.5)
// whenever that argument is further than `margin` from a rounding boundary (the tile kernel compares margin with the
// bound that goes with the frame size and falls back to the float expressions otherwise).
struct BoxInt {
    int16_t off_tl;           // byte offset of the top-left corner from the keypoint's own corner in the u16 integral rows
    uint16_t w2;              // 2 * box width: byte distance of the right corners
    uint16_t step1;           // byte distance to the bottom corners of the first slice (min(rows_per_slice, h) rows)
    uint8_t left, rps;        // rows after the first slice; rows per slice
    float inv_area;           // 1 / (w * h), correctly rounded
    float margin;             // smallest distance of the four corner arguments from a rounding boundary
};
static_assert(sizeof(BoxInt) == 16, "one dwordx4 per lane");

struct OrientPair {
    int32_t i, j, weight_dx, weight_dy;
};

// One row of the resize tables: for output index d of a 19-wide axis.
struct ResizeTap {
    int16_t ofs;    // source index (already clamped for the x axis)
    int16_t ofs1;   // second source index: x: ofs+1 (or ofs when d >= xmax); y: clip(ofs+1)
    int16_t c0, c1; // fixed-point weights (11 bits); x axis at d >= xmax: {2048, 0}
};

// ---- geometry of the fused tile kernel (tile_kernel.hip); the MIP sample table below bakes in kTileStagePitch
constexpr int kTileW = 128, kTileH = 64;  // pixels a workgroup owns: 128 keypoints of an 8-pixel grid, whole groups and whole pairs of
                                          // keypoints for every wave of the tile kernel (96 x 64: 86 of them, 2.7 groups per wave)
constexpr int kTileHalo = 40;            // FREAK: keypoints whose patternSizes[scale] <= kTileHalo take the tile path (size < ~12.7;
                                          // 48 with 96-pixel tiles: what two workgroups' LDS per CU allows for the wider region)
constexpr int kTileHaloX = (kTileHalo + 15) & ~15;  // 16-byte pieces of a region row start on 16 bytes of the frame row
constexpr int kTileMipHalo = 8;          // MIP: ROI reach beyond the tile (the smallest halo, 24, covers it)
constexpr int kTileRW = kTileW + 2 * kTileHaloX, kTileRH = kTileH + 2 * kTileHalo;          // largest integral region 224 x 144
// One LDS row of the tile kernel: first the staged gray bytes of a region row (current at byte 0, previous at byte
// kTileRW), later, in place, the row of the u16 integral (kTileRW + 8 entries).
constexpr int kTileStagePitch = 2 * (kTileRW + 8);
constexpr int kTileMaxRoi = 16;          // largest ROI side the tile path samples
constexpr int kP19Pad = 368;             // bytes reserved per 19x19 buffer

// One of the 19x19 output pixels the MIP actually reads, for one ROI side L: the four source bytes (offsets from
// the ROI's top-left inside the kTileStagePitch-pitch staged rows) and the fixed-point weights of cv::resize.
struct MipSample {   // 16 bytes: one dwordx4 per lane and pass
    uint16_t off_row0, off_row1;  // first source byte of the two rows; the frame's offset inside a staged row is included
    uint32_t cx;                  // x weights, c0 | c1 << 16 (11-bit fixed point); the second byte of a row is the next one
    uint32_t c0y_s12, c1y_s12;    // y weights << 12: (w * (t >> 4)) >> 16 == mul_hi_u24(t & ~15, w << 12)
};

// One step of the thetaIdx staircase: the direction (cos, sin) of the exact angle at which the index changes.
struct ThetaBound {
    double c, s;
};
constexpr int kThetaBounds = 256;  // [0..127] upper half plane beta_1..beta_128, [128..254] lower half mu_1..mu_127

struct FreakParams {
    float pattern_scale = 22.0f;
    int n_octaves = 4;
    bool orientation_normalized = true;
    bool scale_normalized = true;
    int bit_mode = 0;
};

struct Tables {
    std::vector<PatternPoint> lut;        // [64][256][43]
    std::vector<BoxInt> lut_int;          // same indexing: the boxes of lut for integer keypoint coordinates
    int32_t pattern_sizes[kNbScales];
    OrientPair orient[kNbOrientPairs];
    // The 64 description pairs that land in descriptor bytes 0..7, indexed by output bit
    // (bit b of byte B = index 8*B + b), for the selected bit mode.
    uint8_t bit_pair_i[64], bit_pair_j[64];
    // scale index = number of thresholds <= size (scale_normalized); 63 entries used
    float scale_thresholds[kNbScales];
    int fixed_scale_index;                // used when !scale_normalized
    float min_sigma;                      // smallest sigma in the LUT (the box sampler needs >= 0.5)
    int max_abs_direction;                // bound on |direction0|, |direction1|
    // resize taps: [L][axis(0=x,1=y)][19], L = 0..kMaxRoiSide (row 0 unused)
    std::vector<ResizeTap> resize;
    // The bytes of a (cur19 | prev19) buffer pair (kP19Pad bytes each) that the tile kernel resamples: every aligned
    // dword holding a 19x19 position motionInterchangePattern reads at the 8 patch centres (MoFREAKUtilities.cpp:56-70,
    // 79-88, 308-316), per ROI side L (which lane takes which dword is chosen per L for LDS bank spread):
    // mip_pos[L * mip_stride + 64 * u + lane] for u < 4 is byte u of the lane's dword; the rest follow byte by byte.
    // value = frame * kP19Pad + row * 19 + col.  mip_samples[L][j] is entry j's MipSample for ROI side L (L <= kTileMaxRoi).
    // thetaIdx steps (see device_helpers.h theta_index): found by bisection over float angles through the chain
    // angle = (float)(a * (180.0/CV_PI)); thetaIdx = int(256*angle*(1/360.0)+0.5)
    std::vector<ThetaBound> theta_bounds;
    std::vector<uint16_t> mip_pos;
    // The 19x19 positions (row * 19 + col) motionInterchangePattern reads of the current / the previous buffer at the 8
    // patch centres, ascending: what the gather path resamples (51 and 225 of the 361 positions).
    std::vector<uint16_t> mip_need_cur, mip_need_prev;
    int mip_n_cur = 0, mip_n = 0, mip_stride = 0;  // mip_stride: entries per L in mip_samples (mip_n rounded up to 64)
    std::vector<MipSample> mip_samples;
};

// Scale index by the reference expression (freak.cpp computeImpl) -- the chain the thresholds are derived from.
int scale_index_from_size(float size, int n_octaves);

void build_theta_bounds(std::vector<ThetaBound> &out);
void build_tables(const FreakParams &p, Tables &t);

}  // namespace mofreak
