// CPU check of the lane-per-keypoint MIP (mofreak_amd/csrc/mip_lane.h) against the oracle: the header is compiled for the
// host (MOFREAK_MIP_LANE_HOST: the gfx950 instructions replaced by plain C++), every ROI side it is instantiated for, ROIs at
// every byte alignment and at the image borders, random and structured frames.  Test infrastructure only
// (tests/test_mip_lane_host.py builds and runs it); prints "ok <cases>" or the first mismatches.
#define MOFREAK_MIP_LANE_HOST 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../mofreak_amd/csrc/mip_lane.h"

extern "C" int mo_mip_descriptor(const uint8_t *cur, const uint8_t *prev, int W, int H, float size, int x, int y, uint8_t out[8]);

using namespace mofreak;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

template <int L, int CM>
static long run_side(const uint8_t *cur, const uint8_t *prev, int W, int H, int theta, long &cases)
{
    long bad = 0;
    // sizes with ceil(size) == L: the integer one (half = L / 2) and a fractional one (half = (L - 1) / 2)
    const float sizes[2] = {(float)L, (float)L - 0.4f};
    for (float size : sizes) {
        const int half = (int)size / 2;
        for (int t = 0; t < 1500; ++t) {
            int x, y;
            if (t < 64) {  // the four corners' neighbourhoods: ROIs that touch the image borders
                x = (t & 1) ? W - (L - half) - (t >> 4 & 3) : half + (t >> 4 & 3);
                y = (t & 2) ? H - (L - half) - (t >> 2 & 3) : half + (t >> 2 & 3);
            } else {
                x = half + (int)(rnd() % (uint32_t)(W - L + 1));
                y = half + (int)(rnd() % (uint32_t)(H - L + 1));
            }
            uint8_t want[8];
            if (mo_mip_descriptor(cur, prev, W, H, size, x, y, want) != 0) continue;
            // (the last rows' fetch may pass the frame's end: the harness' buffers carry 32 spare bytes, as the kernel's
            // callers keep such tiles off this path)
            const int64_t roi = (int64_t)(y - half) * W + (x - half);
            const MipUint2 got = mip_lane_keypoint<L, CM>(cur, prev, (uint32_t)roi, W, theta);
            uint8_t packed[8], g[8];
            for (int k = 0; k < 4; ++k) {
                packed[k] = (uint8_t)(got.x >> (8 * k));
                packed[4 + k] = (uint8_t)(got.y >> (8 * k));
            }
            ++cases;
            bool same = true;
            for (int c = 0, k = 0; c < 8; ++c) {  // the mask's centres are packed in ascending order; the other bytes stay zero
                g[c] = ((CM >> c) & 1) ? packed[k++] : want[c];
                same = same && g[c] == want[c];
            }
            for (int k = __builtin_popcount(CM); k < 8; ++k) same = same && packed[k] == 0;
            if (!same && bad++ < 5) {
                std::printf("L=%d size=%.1f x=%d y=%d: got", L, size, x, y);
                for (int k = 0; k < 8; ++k) std::printf(" %02x", g[k]);
                std::printf(" want");
                for (int k = 0; k < 8; ++k) std::printf(" %02x", want[k]);
                std::printf("\n");
            }
        }
    }
    return bad;
}

int main(int argc, char **argv)
{
    const int W = 148, H = 96;  // a multiple of 4 (the path's precondition), not of 8
    // both frames in one allocation, 4-byte aligned, with slack behind each
    std::vector<uint32_t> store((2 * (W * H + 64)) / 4 + 4);
    uint8_t *cur = reinterpret_cast<uint8_t *>(store.data()), *prev = cur + W * H + 64;
    long bad = 0, cases = 0;
    for (int kind = 0; kind < 3; ++kind) {
        for (int i = 0; i < W * H; ++i) {
            const int x = i % W, y = i / W;
            if (kind == 0) {  // noise around a gradient: SSDs on both sides of the threshold
                cur[i] = (uint8_t)((x * 3 + y * 2 + (int)(rnd() % 9)) & 0xff);
                prev[i] = (uint8_t)((x * 3 + y * 2 + 4 + (int)(rnd() % 17)) & 0xff);
            } else if (kind == 1) {  // full-range noise
                cur[i] = (uint8_t)rnd();
                prev[i] = (uint8_t)rnd();
            } else {  // extremes: rounding at 0 / 255, large SSDs
                cur[i] = (rnd() & 1) ? 255 : 0;
                prev[i] = (rnd() & 3) ? 255 : 0;
            }
        }
        bad += run_side<7, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<7, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<7, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<8, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<8, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<8, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<9, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<9, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<9, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<10, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<10, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<10, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<11, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<11, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<11, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<12, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<12, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<12, kMipMaskB>(cur, prev, W, H, 288, cases);
        bad += run_side<13, kMipMaskAll>(cur, prev, W, H, 288, cases);
        bad += run_side<13, kMipMaskA>(cur, prev, W, H, 288, cases);
        bad += run_side<13, kMipMaskB>(cur, prev, W, H, 288, cases);
    }
    if (bad) {
        std::printf("MISMATCH %ld of %ld\n", bad, cases);
        return 1;
    }
    std::printf("ok %ld\n", cases);
    return 0;
}
