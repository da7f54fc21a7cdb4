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
    float x, y, sigma, pad;
};

struct OrientPair {
    int32_t i, j, weight_dx, weight_dy;
};

// One row of the resize tables: for output index d of a 19-wide axis.
struct ResizeTap {
    int16_t ofs;    // source index (already clamped for the x axis)
    int16_t ofs1;   // second source index: x: ofs+1 (or ofs when d >= xmax); y: clip(ofs+1)
    int16_t c0, c1; // fixed-point weights (11 bits); x axis at d >= xmax: {2048, 0}
};

struct FreakParams {
    float pattern_scale = 22.0f;
    int n_octaves = 4;
    bool orientation_normalized = true;
    bool scale_normalized = true;
    int bit_mode = 0;
};

struct Tables {
    std::vector<PatternPoint> lut;        // [64][256][43]
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
};

// Scale index by the reference expression (freak.cpp computeImpl) -- the chain the thresholds are derived from.
int scale_index_from_size(float size, int n_octaves);

void build_tables(const FreakParams &p, Tables &t);

}  // namespace mofreak
