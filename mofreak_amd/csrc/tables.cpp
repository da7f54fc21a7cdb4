// Host-side table construction for the MoFREAK path.  See tables.h.
// Compile with -ffp-contract=off: the float/double expression order below is part of the contract.
#include "tables.h"

#include "resize_axis.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace mofreak {

namespace {

#include "mip_lane_order.inc"

constexpr double kCvPi = 3.1415926535897932384626433832795;  // OpenCV's CV_PI
constexpr double kFreakLog2 = 0.693147180559945;             // freak.cpp FREAK_LOG2

// FREAK_DEF_PAIRS (freak.cpp): which of the 903 point pairs (i > j, enumerated i-major) make up the
// 512 descriptor bits.  MoFREAK keeps descriptor bytes 0..7 only (MoFREAKUtilities.cpp:453-456).
const uint16_t kDefPairs[kNbPairs] = {
    404, 431, 818, 511, 181, 52,  311, 874, 774, 543, 719, 230, 417, 205, 11,  560, 149, 265, 39,  306, 165, 857,
    250, 8,   61,  15,  55,  717, 44,  412, 592, 134, 761, 695, 660, 782, 625, 487, 549, 516, 271, 665, 762, 392,
    178, 796, 773, 31,  672, 845, 548, 794, 677, 654, 241, 831, 225, 238, 849, 83,  691, 484, 826, 707, 122, 517,
    583, 731, 328, 339, 571, 475, 394, 472, 580, 381, 137, 93,  380, 327, 619, 729, 808, 218, 213, 459, 141, 806,
    341, 95,  382, 568, 124, 750, 193, 749, 706, 843, 79,  199, 317, 329, 768, 198, 100, 466, 613, 78,  562, 783,
    689, 136, 838, 94,  142, 164, 679, 219, 419, 366, 418, 423, 77,  89,  523, 259, 683, 312, 555, 20,  470, 684,
    123, 458, 453, 833, 72,  113, 253, 108, 313, 25,  153, 648, 411, 607, 618, 128, 305, 232, 301, 84,  56,  264,
    371, 46,  407, 360, 38,  99,  176, 710, 114, 578, 66,  372, 653, 129, 359, 424, 159, 821, 10,  323, 393, 5,
    340, 891, 9,   790, 47,  0,   175, 346, 236, 26,  172, 147, 574, 561, 32,  294, 429, 724, 755, 398, 787, 288,
    299, 769, 565, 767, 722, 757, 224, 465, 723, 498, 467, 235, 127, 802, 446, 233, 544, 482, 800, 318, 16,  532,
    801, 441, 554, 173, 60,  530, 713, 469, 30,  212, 630, 899, 170, 266, 799, 88,  49,  512, 399, 23,  500, 107,
    524, 90,  194, 143, 135, 192, 206, 345, 148, 71,  119, 101, 563, 870, 158, 254, 214, 276, 464, 332, 725, 188,
    385, 24,  476, 40,  231, 620, 171, 258, 67,  109, 844, 244, 187, 388, 701, 690, 50,  7,   850, 479, 48,  522,
    22,  154, 12,  659, 736, 655, 577, 737, 830, 811, 174, 21,  237, 335, 353, 234, 53,  270, 62,  182, 45,  177,
    245, 812, 673, 355, 556, 612, 166, 204, 54,  248, 365, 226, 242, 452, 700, 685, 573, 14,  842, 481, 468, 781,
    564, 416, 179, 405, 35,  819, 608, 624, 367, 98,  643, 448, 2,   460, 676, 440, 240, 130, 146, 184, 185, 430,
    65,  807, 377, 82,  121, 708, 239, 310, 138, 596, 730, 575, 477, 851, 797, 247, 27,  85,  586, 307, 779, 326,
    494, 856, 324, 827, 96,  748, 13,  397, 125, 688, 702, 92,  293, 716, 277, 140, 112, 4,   80,  855, 839, 1,
    413, 347, 584, 493, 289, 696, 19,  751, 379, 76,  73,  115, 6,   590, 183, 734, 197, 483, 217, 344, 330, 400,
    186, 243, 587, 220, 780, 200, 793, 246, 824, 41,  735, 579, 81,  703, 322, 760, 720, 139, 480, 490, 91,  814,
    813, 163, 152, 488, 763, 263, 425, 410, 576, 120, 319, 668, 150, 160, 302, 491, 515, 260, 145, 428, 97,  251,
    395, 272, 252, 18,  106, 358, 854, 485, 144, 550, 131, 133, 378, 68,  102, 104, 58,  361, 275, 209, 697, 582,
    338, 742, 589, 325, 408, 229, 28,  304, 191, 189, 110, 126, 486, 211, 547, 533, 70,  215, 670, 249, 36,  581,
    389, 605, 331, 518, 442, 822};

// The 45 orientation pairs: within each of the six outer rings the 3 diameters and 6 next-but-one
// chords, then the 3 diameters of rings 5/6 and ring 7 (freak.cpp buildPattern()).
void orientation_pair_indices(int (&ij)[kNbOrientPairs][2])
{
    int m = 0;
    for (int ring = 0; ring < 4; ++ring) {
        const int b = 6 * ring;
        const int local[9][2] = {{0, 3}, {1, 4}, {2, 5}, {0, 2}, {1, 3}, {2, 4}, {3, 5}, {4, 0}, {5, 1}};
        for (const auto &l : local) {
            ij[m][0] = b + l[0];
            ij[m][1] = b + l[1];
            ++m;
        }
    }
    for (int b : {24, 30, 36})
        for (int k = 0; k < 3; ++k) {
            ij[m][0] = b + k;
            ij[m][1] = b + k + 3;
            ++m;
        }
}

inline int clip_index(int v, int n) { return v < 0 ? 0 : (v < n ? v : n - 1); }

inline int16_t saturate_short_round(float v)
{
    // cv::saturate_cast<short>(float) == cvRound (round-half-to-even) then clamp
    const long r = std::lrintf(v);
    return static_cast<int16_t>(r > 32767 ? 32767 : (r < -32768 ? -32768 : r));
}

// Coefficient row of cv::resize's INTER_LINEAR pass for a source axis of length L -> 19 outputs.
void build_resize_axis(int L, bool is_x, ResizeTap *out)
{
    const double inv_scale = static_cast<double>(kPatch) / L;
    const double scale = 1. / inv_scale;
    int dmax = kPatch;
    int s_arr[kPatch];
    float f_arr[kPatch];
    for (int d = 0; d < kPatch; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = static_cast<int>(std::floor(static_cast<double>(f)));
        f -= s;
        if (is_x) {
            if (s < 0) {
                f = 0;
                s = 0;
            }
            if (s + 1 >= L) {
                dmax = std::min(dmax, d);
                if (s >= L - 1) {
                    f = 0;
                    s = L - 1;
                }
            }
        }
        s_arr[d] = s;
        f_arr[d] = f;
    }
    for (int d = 0; d < kPatch; ++d) {
        ResizeTap t;
        const int16_t c0 = saturate_short_round((1.f - f_arr[d]) * 2048);
        const int16_t c1 = saturate_short_round(f_arr[d] * 2048);
        if (is_x) {
            if (d < dmax) {
                t.ofs = static_cast<int16_t>(s_arr[d]);
                t.ofs1 = static_cast<int16_t>(s_arr[d] + 1);
                t.c0 = c0;
                t.c1 = c1;
            } else {  // HResizeLinear's tail: D[dx] = S[sx] * ONE
                t.ofs = t.ofs1 = static_cast<int16_t>(s_arr[d]);
                t.c0 = 2048;
                t.c1 = 0;
            }
        } else {  // rows are clamped when fetched; the weights keep the unclamped fraction
            t.ofs = static_cast<int16_t>(clip_index(s_arr[d], L));
            t.ofs1 = static_cast<int16_t>(clip_index(s_arr[d] + 1, L));
            t.c0 = c0;
            t.c1 = c1;
        }
        out[d] = t;
    }
}

// thetaIdx before wrapping, from the float angle in radians (freak.cpp computeImpl; the chain after atan2)
int theta_raw_from_float(float a)
{
    const float angle = static_cast<float>(static_cast<double>(a) * (180.0 / kCvPi));
    return static_cast<int>(static_cast<double>(256.0f * angle) * (1 / 360.0) + 0.5);
}

inline float float_from_bits(uint32_t b)
{
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

}  // namespace

void build_theta_bounds(std::vector<ThetaBound> &out)
{
    out.assign(kThetaBounds, ThetaBound{1.0, 0.0});
    const float pi_f = static_cast<float>(M_PI);
    uint32_t pi_bits;
    std::memcpy(&pi_bits, &pi_f, 4);
    // upper half plane: beta_k = the real angle above which the rounded float angle gives an index >= k.
    // The float atan2 is the correctly rounded one, so the step sits at the midpoint of two adjacent floats.
    for (int k = 1; k <= 128; ++k) {
        uint32_t lo = 0, hi = pi_bits;  // raw(+0) = 0 < k <= raw(pi_f) = 128; positive floats order like their bits
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (theta_raw_from_float(float_from_bits(mid)) >= k)
                hi = mid;
            else
                lo = mid;
        }
        const double beta = (static_cast<double>(float_from_bits(hi - 1)) + static_cast<double>(float_from_bits(hi))) / 2;
        out[k - 1] = ThetaBound{std::cos(beta), std::sin(beta)};
    }
    // lower half plane: the index is <= -m once |angle| exceeds mu_m
    for (int m = 1; m <= 127; ++m) {
        uint32_t lo = 0, hi = pi_bits;  // raw(-0) = 0 > -m; raw(-pi_f) = -127 <= -m
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (theta_raw_from_float(-float_from_bits(mid)) <= -m)
                hi = mid;
            else
                lo = mid;
        }
        const double mu = (static_cast<double>(float_from_bits(hi - 1)) + static_cast<double>(float_from_bits(hi))) / 2;
        out[128 + m - 1] = ThetaBound{std::cos(mu), std::sin(mu)};
    }
}

int scale_index_from_size(float size, int n_octaves)
{
    // kpScaleIdx[k] = max((int)(log(keypoints[k].size/FREAK_SMALLEST_KP_SIZE)*sizeCst+0.5), 0), capped at 63;
    // log(float) on the reference's MSVC x86 CRT is (float)log((double)).
    const float size_cst = static_cast<float>(kNbScales / (kFreakLog2 * n_octaves));
    const float ratio = size / kSmallestKpSize;
    const float lg = static_cast<float>(std::log(static_cast<double>(ratio)));
    int idx = static_cast<int>(lg * size_cst + 0.5);
    idx = std::max(idx, 0);
    return std::min(idx, kNbScales - 1);
}

void build_tables(const FreakParams &p, Tables &t)
{
    // ---- pattern LUT (buildPattern)
    t.lut.assign(static_cast<size_t>(kNbScales) * kNbOrientation * kNbPoints, PatternPoint{0, 0, 0, 1});
    const double scale_step = std::pow(2.0, static_cast<double>(p.n_octaves) / kNbScales);
    const int ring_points[8] = {6, 6, 6, 6, 6, 6, 6, 1};
    const double big_r = 2.0 / 3.0, small_r = 2.0 / 24.0;
    const double unit = (big_r - small_r) / 21.0;
    const double radius[8] = {big_r,           big_r - 6 * unit,  big_r - 11 * unit, big_r - 15 * unit,
                              big_r - 18 * unit, big_r - 20 * unit, small_r,           0.0};
    double sigma[8];
    for (int i = 0; i < 7; ++i) sigma[i] = radius[i] / 2.0;
    sigma[7] = radius[6] / 2.0;

    t.min_sigma = FLT_MAX;
    for (int sc = 0; sc < kNbScales; ++sc) {
        t.pattern_sizes[sc] = 0;
        const double scaling = std::pow(scale_step, sc);
        for (int rot = 0; rot < kNbOrientation; ++rot) {
            const double theta = double(rot) * 2 * kCvPi / double(kNbOrientation);
            PatternPoint *row = &t.lut[(static_cast<size_t>(sc) * kNbOrientation + rot) * kNbPoints];
            int pt = 0;
            for (int ring = 0; ring < 8; ++ring) {
                const int n = ring_points[ring];
                for (int k = 0; k < n; ++k, ++pt) {
                    const double beta = M_PI / n * (ring % 2);
                    const double alpha = double(k) * 2 * M_PI / double(n) + beta + theta;
                    row[pt].x = static_cast<float>(radius[ring] * std::cos(alpha) * scaling * p.pattern_scale);
                    row[pt].y = static_cast<float>(radius[ring] * std::sin(alpha) * scaling * p.pattern_scale);
                    row[pt].sigma = static_cast<float>(sigma[ring] * scaling * p.pattern_scale);
                    row[pt].rows_per_slice = std::max(1, 257 / (static_cast<int>(2.0f * row[pt].sigma) + 3));
                    t.min_sigma = std::min(t.min_sigma, row[pt].sigma);
                    const int size_max =
                        static_cast<int>(std::ceil((radius[ring] + sigma[ring]) * scaling * p.pattern_scale)) + 1;
                    t.pattern_sizes[sc] = std::max(t.pattern_sizes[sc], size_max);
                }
            }
        }
    }

    // ---- the same boxes for keypoints at integer coordinates (BoxInt)
    t.lut_int.assign(t.lut.size(), BoxInt{0, 0, 0, 0, 1, 0.0f, 0.0f});
    for (size_t i = 0; i < t.lut.size(); ++i) {
        const PatternPoint &P = t.lut[i];
        const double args[4] = {static_cast<double>(P.x) - P.sigma + 0.5, static_cast<double>(P.y) - P.sigma + 0.5,
                                static_cast<double>(P.x) + P.sigma + 0.5, static_cast<double>(P.y) + P.sigma + 0.5};
        double margin = 1.0;
        int fl[4];
        for (int k = 0; k < 4; ++k) {
            const double f = std::floor(args[k]);
            fl[k] = static_cast<int>(f);
            margin = std::min(margin, std::min(args[k] - f, f + 1.0 - args[k]));
        }
        const int dxl = fl[0], dyt = fl[1], w = fl[2] + 1 - fl[0], h = fl[3] + 1 - fl[1];
        BoxInt &b = t.lut_int[i];
        const int off = 2 * (dyt * (kTileStagePitch / 2) + dxl);
        if (w < 1 || h < 1 || w > 127 || h > 127 || off < -32768 || off > 32767) {
            b.margin = 0.0f;  // never taken: the float path handles it (patterns far beyond the tile path's reach)
            continue;
        }
        const int rps = std::max(1, 257 / w), first = std::min(rps, h);
        b.off_tl = static_cast<int16_t>(off);
        b.w2 = static_cast<uint16_t>(2 * w);
        b.step1 = static_cast<uint16_t>(first * kTileStagePitch);
        b.left = static_cast<uint8_t>(h - first);
        b.rps = static_cast<uint8_t>(std::min(rps, 255));
        b.inv_area = 1.0f / static_cast<float>(w * h);
        b.margin = static_cast<float>(margin);
    }

    // ---- orientation pairs and their fixed-point weights (from scale 0, orientation 0)
    int ij[kNbOrientPairs][2];
    orientation_pair_indices(ij);
    long bound0 = 0, bound1 = 0;
    for (int m = 0; m < kNbOrientPairs; ++m) {
        OrientPair &o = t.orient[m];
        o.i = ij[m][0];
        o.j = ij[m][1];
        const float dx = t.lut[o.i].x - t.lut[o.j].x;
        const float dy = t.lut[o.i].y - t.lut[o.j].y;
        const float norm_sq = (dx * dx + dy * dy);
        o.weight_dx = static_cast<int>((dx / (norm_sq)) * 4096.0 + 0.5);
        o.weight_dy = static_cast<int>((dy / (norm_sq)) * 4096.0 + 0.5);
        bound0 += 255L * std::labs(o.weight_dx) / 2048;
        bound1 += 255L * std::labs(o.weight_dy) / 2048;
    }
    t.max_abs_direction = static_cast<int>(std::max(bound0, bound1));

    // ---- description pairs -> the 64 bits of descriptor bytes 0..7
    uint8_t all_i[903], all_j[903];
    {
        int c = 0;
        for (int i = 1; i < kNbPoints; ++i)
            for (int j = 0; j < i; ++j, ++c) {
                all_i[c] = static_cast<uint8_t>(i);
                all_j[c] = static_cast<uint8_t>(j);
            }
    }
    for (int B = 0; B < 8; ++B)
        for (int b = 0; b < 8; ++b) {
            // SSE layout: within the first 128-pair block, group b (mask bit b) holds pairs 16b..16b+15 and
            // _mm_set_epi8 puts pair 16b+t in byte 15-t  =>  byte B, bit b  <-  pair 16*b + (15-B).
            // Natural (std::bitset) layout: byte B, bit b  <-  pair 8*B + b.
            const int pair = (p.bit_mode == 1) ? (8 * B + b) : (16 * b + (15 - B));
            t.bit_pair_i[8 * B + b] = all_i[kDefPairs[pair]];
            t.bit_pair_j[8 * B + b] = all_j[kDefPairs[pair]];
        }

    // ---- scale index thresholds: thresholds[k] = smallest float size whose index is >= k+1
    for (int k = 0; k < kNbScales; ++k) t.scale_thresholds[k] = FLT_MAX;
    for (int k = 1; k < kNbScales; ++k) {
        uint32_t lo = 0x00800000u;  // FLT_MIN: index 0
        uint32_t hi = 0x7f7fffffu;  // FLT_MAX: index 63
        while (hi - lo > 1) {       // positive floats order like their bit patterns
            const uint32_t mid = lo + (hi - lo) / 2;
            float f;
            std::memcpy(&f, &mid, 4);
            if (scale_index_from_size(f, p.n_octaves) >= k)
                hi = mid;
            else
                lo = mid;
        }
        std::memcpy(&t.scale_thresholds[k - 1], &hi, 4);
    }
    {
        // !scaleNormalized: const int scIdx = max((int)(1.0986122886681*sizeCst+0.5), 0)
        const float size_cst = static_cast<float>(kNbScales / (kFreakLog2 * p.n_octaves));
        int idx = std::max(static_cast<int>(1.0986122886681 * size_cst + 0.5), 0);
        t.fixed_scale_index = std::min(idx, kNbScales - 1);
    }

    // ---- resize coefficient tables for every ROI side
    t.resize.assign(static_cast<size_t>(kMaxRoiSide + 1) * 2 * kPatch, ResizeTap{0, 0, 0, 0});
    for (int L = 1; L <= kMaxRoiSide; ++L) {
        build_resize_axis(L, true, &t.resize[(static_cast<size_t>(L) * 2 + 0) * kPatch]);
        build_resize_axis(L, false, &t.resize[(static_cast<size_t>(L) * 2 + 1) * kPatch]);
    }

    // The tile kernel's lane-per-keypoint MIP (mip_lane.h) is compiled per ROI side against the compile-time construction of
    // the same rows (resize_axis.h): the two must agree entry by entry, or that kernel would resample with other taps than
    // every other path.
    for (int L = 1; L <= kTileMaxRoi; ++L)
        for (int axis = 0; axis < 2; ++axis) {
            const ResizeAxisC c = make_resize_axis(L, axis == 0);
            const ResizeTap *r = &t.resize[(static_cast<size_t>(L) * 2 + axis) * kPatch];
            for (int d = 0; d < kPatch; ++d)
                if (r[d].ofs != c.ofs[d] || r[d].ofs1 != c.ofs1[d] || r[d].c0 != c.c0[d] || r[d].c1 != c.c1[d])
                    throw std::logic_error("resize taps: the compile-time rows of resize_axis.h differ from build_resize_axis at ROI side " +
                                           std::to_string(L));
        }

    build_theta_bounds(t.theta_bounds);

    // ---- the subset of the two 19x19 buffers the MIP reads, and its per-L sample table for the tile kernel
    {
        const int centers[8][2] = {{5, 5}, {5, 9}, {5, 13}, {9, 5}, {9, 13}, {13, 5}, {13, 9}, {13, 13}};  // (x, y)
        const int offs[8][2] = {{-4, 0}, {-3, 3}, {0, 4}, {3, 3}, {4, 0}, {3, -3}, {0, -4}, {-3, -3}};     // (dx, dy)
        std::vector<char> need_cur(kPatch * kPatch, 0), need_prev(kPatch * kPatch, 0);
        for (const auto &c : centers) {
            const int bc = (c[1] - 1) * kPatch + (c[0] - 1);
            for (int k = 0; k < 9; ++k) need_cur[bc + k] = 1;
            for (const auto &o : offs) {
                const int bp = (c[1] + o[1] - 1) * kPatch + (c[0] + o[0] - 1);
                for (int k = 0; k < 9; ++k) need_prev[bp + k] = 1;
            }
        }
        t.mip_need_cur.clear();
        t.mip_need_prev.clear();
        for (int i = 0; i < kPatch * kPatch; ++i) {
            if (need_cur[i]) t.mip_need_cur.push_back(static_cast<uint16_t>(i));
            if (need_prev[i]) t.mip_need_prev.push_back(static_cast<uint16_t>(i));
        }
        // Sample order for the tile kernel: the needed bytes are covered by aligned dwords of the (cur19 | prev19)
        // buffer pair; the first 64 dwords go one per lane with byte u in pass u (so that a lane packs its four
        // results into one 32-bit store), the remaining dwords byte by byte in the last pass.  Bytes of a covering
        // dword that the MIP never reads are resampled as well (harmless); bytes in a buffer's padding take the last
        // pixel's sample.
        std::vector<int> dwords;
        for (int d = 0; d < 2 * kP19Pad / 4; ++d) {
            bool any = false;
            for (int b = 0; b < 4; ++b) {
                const int byte = 4 * d + b, fr = byte / kP19Pad, i = byte % kP19Pad;
                if (i < kPatch * kPatch && (fr ? need_prev[i] : need_cur[i])) any = true;
            }
            if (any) dwords.push_back(d);
        }
        // Which dword a lane takes is chosen per ROI side (kMipLaneOrder, generated by mofreak_amd/tools/mip_lane_order.py): in
        // ascending order every sampling pass pays a 2-way LDS bank conflict in its first lane group (`current` rows against
        // `previous` rows staged 48 dwords further); the generated orders are conflict-free at all four ROI alignments.
        const int n_dw = static_cast<int>(dwords.size());
        if (n_dw < 65 || n_dw > 80)  // the kernel's five passes assume 64 full dwords plus a partial pass
            throw std::logic_error("MIP sample table: " + std::to_string(n_dw) + " dwords, the tile kernel expects 65..80");
        t.mip_n = 4 * n_dw;
        t.mip_stride = (t.mip_n + 63) / 64 * 64;
        t.mip_pos.assign(static_cast<size_t>(kTileMaxRoi + 1) * t.mip_stride, 0);
        for (int L = 0; L <= kTileMaxRoi; ++L) {
            std::vector<int> order(n_dw);
            for (int i = 0; i < n_dw; ++i) order[i] = n_dw == kMipOrderDwords ? kMipLaneOrder[L][i] : i;
            {  // (a permutation, or the table is corrupt)
                std::vector<char> seen(n_dw, 0);
                for (int i : order) {
                    if (i < 0 || i >= n_dw || seen[i]) throw std::logic_error("MIP lane order: not a permutation");
                    seen[i] = 1;
                }
            }
            uint16_t *pos = &t.mip_pos[static_cast<size_t>(L) * t.mip_stride];
            int n = 0;
            for (int u = 0; u < 4; ++u)
                for (int lane = 0; lane < 64; ++lane) pos[n++] = static_cast<uint16_t>(4 * dwords[order[lane]] + u);
            for (int d = 64; d < n_dw; ++d)
                for (int b = 0; b < 4; ++b) pos[n++] = static_cast<uint16_t>(4 * dwords[order[d]] + b);
            for (; n < t.mip_stride; ++n) pos[n] = pos[t.mip_n - 1];
        }
        t.mip_n_cur = 0;
        for (int j = 0; j < t.mip_n; ++j) t.mip_n_cur += t.mip_pos[j] < kP19Pad ? 1 : 0;
        t.mip_samples.assign(static_cast<size_t>(kTileMaxRoi + 1) * t.mip_stride, MipSample{0, 0, 0, 0, 0});
        for (int L = 1; L <= kTileMaxRoi; ++L) {
            const ResizeTap *tx = &t.resize[(static_cast<size_t>(L) * 2 + 0) * kPatch];
            const ResizeTap *ty = &t.resize[(static_cast<size_t>(L) * 2 + 1) * kPatch];
            const uint16_t *pos_L = &t.mip_pos[static_cast<size_t>(L) * t.mip_stride];
            for (int j = 0; j < t.mip_stride; ++j) {
                const int frame = pos_L[j] / kP19Pad;  // 0: current, 1: previous (staged kTileRW bytes further)
                const int pos = std::min(pos_L[j] % kP19Pad, kPatch * kPatch - 1), dy = pos / kPatch, dx = pos % kPatch;
                MipSample &m = t.mip_samples[static_cast<size_t>(L) * t.mip_stride + j];
                // the tile kernel reads a row pair as (off, off + 1): a clamped column has to carry a zero weight
                if (tx[dx].ofs1 != tx[dx].ofs + 1 && tx[dx].c1 != 0)
                    throw std::logic_error("MIP sample table: a clamped resize column carries a weight");
                m.off_row0 = static_cast<uint16_t>(frame * kTileRW + ty[dy].ofs * kTileStagePitch + tx[dx].ofs);
                m.off_row1 = static_cast<uint16_t>(frame * kTileRW + ty[dy].ofs1 * kTileStagePitch + tx[dx].ofs);
                m.cx = static_cast<uint32_t>(static_cast<uint16_t>(tx[dx].c0)) | static_cast<uint32_t>(static_cast<uint16_t>(tx[dx].c1)) << 16;
                m.c0y_s12 = static_cast<uint32_t>(static_cast<uint16_t>(ty[dy].c0)) << 12;
                m.c1y_s12 = static_cast<uint32_t>(static_cast<uint16_t>(ty[dy].c1)) << 12;
            }
        }
    }
}

}  // namespace mofreak
