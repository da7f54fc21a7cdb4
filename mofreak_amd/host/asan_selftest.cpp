// Host-side sanitizer build of the C ABI's CPU code (tables, row text, argument checks) and of the C++ facade:
//   make -C mofreak_amd/host asan   ->  ./asan_selftest   (AddressSanitizer + UBSan, no GPU involved)
// The kernel launchers are replaced by stubs that report "no device": nothing here computes a descriptor -- the device
// path has no CPU fallback -- but every byte the host code touches on its own is checked.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../csrc/device_types.h"
#include "MoFREAKUtilities.h"

namespace mofreak {
#define STUB return 100 /* hipErrorNoDevice */
int launch_integral(const IntegralArgs &, void *) { STUB; }
int launch_describe(const DescribeArgs &, int, void *) { STUB; }
int launch_mip19(const uint8_t *, const uint8_t *, int64_t, int, uint8_t *, void *) { STUB; }
int launch_theta(const ThetaBound *, const int32_t *, int64_t, int32_t *, void *) { STUB; }
int launch_compact(const CompactArgs &, void *) { STUB; }
int launch_bin(const BinArgs &, void *) { STUB; }
int launch_tile(const TileArgs &, void *) { STUB; }
int launch_bgr2gray(const uint8_t *, int, int, int64_t, int64_t, int, uint8_t *, void *) { STUB; }
int launch_bow_assign(const uint8_t *, const uint8_t *, int64_t, const uint8_t *, int, int32_t *, unsigned int *, int, void *, void *) { STUB; }
size_t bow_expanded_bytes(int) { return 0; }
int launch_bow_normalize(const unsigned int *, int, float *, int32_t *, void *) { STUB; }
int launch_unpack_integral(const int32_t *, int, int, int, int, int32_t *, void *) { STUB; }
int launch_det_pyramid(const DetArgs &, void *) { STUB; }
int launch_det_scores(const DetArgs &, void *) { STUB; }
int launch_det_corners(const DetArgs &, void *) { STUB; }
int launch_det_keypoints(const DetArgs &, int64_t *, void *) { STUB; }
size_t format_workspace_bytes(int64_t, int) { return 256; }
int launch_format_measure(const mofreak_row *, int64_t, void *, const int64_t *, int, int32_t *, uint64_t **, uint64_t **, void *) { STUB; }
int launch_format_write(const mofreak_row *, int64_t, void *, char *, uint64_t, void *) { STUB; }
}  // namespace mofreak

#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                  \
        }                                                              \
    } while (0)

int main()
{
    mofreak_params p;
    CHECK(mofreak_default_params(&p) == MOFREAK_OK);
    mofreak_ctx *ctx = nullptr;
    CHECK(mofreak_create(MOFREAK_TABLES_ONLY, &p, &ctx) == MOFREAK_OK && ctx);
    // tables: every getter over its whole range
    int32_t sizes[64];
    CHECK(mofreak_pattern_sizes(ctx, sizes) == MOFREAK_OK && sizes[0] == 23 && sizes[12] == 38);
    for (float s : {0.0f, 6.9f, 7.0f, 12.0f, 40.5f, 1e9f}) {
        int idx = -1;
        CHECK(mofreak_scale_index(ctx, s, &idx) == MOFREAK_OK && idx >= 0 && idx < 64);
    }
    std::vector<float> pat(43 * 3);
    for (int sc : {0, 12, 63})
        for (int rot : {0, 77, 255}) CHECK(mofreak_table_pattern(ctx, sc, rot, pat.data()) == MOFREAK_OK);
    CHECK(mofreak_table_pattern(ctx, 64, 0, pat.data()) != MOFREAK_OK);
    std::vector<int32_t> orient(45 * 4);
    CHECK(mofreak_table_orientation(ctx, orient.data()) == MOFREAK_OK);
    std::vector<uint8_t> pairs(64 * 2);
    CHECK(mofreak_table_bit_pairs(ctx, pairs.data()) == MOFREAK_OK);
    std::vector<int16_t> taps(2 * 19 * 4);
    for (int L : {1, 7, 12, 19, 38, 2048}) CHECK(mofreak_table_resize(ctx, L, taps.data()) == MOFREAK_OK);
    CHECK(mofreak_table_resize(ctx, 2049, taps.data()) != MOFREAK_OK);
    // rows -> text -> rows, exact buffer sizes and one byte short
    std::vector<mofreak_row> rows(257);
    for (size_t i = 0; i < rows.size(); ++i) {
        std::memset(&rows[i], 0, sizeof(mofreak_row));
        rows[i].x = 1.25f * (float)i;
        rows[i].y = 1080.0f - (float)i / 3;
        rows[i].frame_number = 4 + (int)i / 7;
        rows[i].scale = i % 2 ? 12.0f : 14.4f;
        for (int k = 0; k < 8; ++k) {
            rows[i].appearance[k] = (uint8_t)(i * 31 + k);
            rows[i].motion[k] = (uint8_t)(i * 17 + 3 * k);
        }
    }
    size_t need = 0;
    CHECK(mofreak_format_rows(rows.data(), (int64_t)rows.size(), nullptr, 0, &need) == MOFREAK_OK && need > 0);
    std::vector<char> text(need);
    size_t wrote = 0;
    CHECK(mofreak_format_rows(rows.data(), (int64_t)rows.size(), text.data(), need, &wrote) == MOFREAK_OK && wrote == need);
    std::vector<char> small(need - 1);  // a buffer one byte short: filled to its end, not beyond
    CHECK(mofreak_format_rows(rows.data(), (int64_t)rows.size(), small.data(), need - 1, &wrote) == MOFREAK_ERR_CAPACITY && wrote == need);
    CHECK(std::memcmp(small.data(), text.data(), need - 1) == 0);
    std::vector<mofreak_row> back(rows.size());
    int64_t n_back = 0;
    CHECK(mofreak_parse_rows(text.data(), need, nullptr, 0, &n_back) == MOFREAK_OK && n_back == (int64_t)rows.size());
    CHECK(mofreak_parse_rows(text.data(), need, back.data(), (int64_t)back.size(), &n_back) == MOFREAK_OK && n_back == (int64_t)rows.size());
    for (size_t i = 0; i < rows.size(); ++i)
        CHECK(std::memcmp(rows[i].appearance, back[i].appearance, 8) == 0 && std::memcmp(rows[i].motion, back[i].motion, 8) == 0 &&
              rows[i].frame_number == back[i].frame_number);
    CHECK(mofreak_parse_rows(text.data(), need, back.data(), 10, &n_back) != MOFREAK_OK);  // capacity
    (void)mofreak_parse_rows(text.data(), need / 2 + 3, back.data(), (int64_t)back.size(), &n_back);  // a cut line: any status, no overrun
    // compute entry points refuse a tables-only context instead of touching memory
    uint8_t frame[64] = {0}, desc[16], valid[1];
    mofreak_keypoint kp{4.0f, 4.0f, 7.0f};
    CHECK(mofreak_extract_pairs(ctx, frame, frame, 8, 8, 8, 64, 1, &kp, nullptr, 1, desc, valid, MOFREAK_MEM_HOST) == MOFREAK_ERR_NO_DEVICE);
    CHECK(std::strlen(mofreak_last_error(ctx)) > 0);
    // the many-clips call: argument checks first, then the same refusal; the clip table is only read, the offsets zeroed
    {
        const uint8_t *clips[3] = {frame, nullptr, frame};
        const int32_t lens[3] = {1, 0, 1};
        int64_t offs[4] = {7, 7, 7, 7}, n_rows = 7;
        mofreak_row row;
        CHECK(mofreak_extract_clips(ctx, clips, lens, -1, 8, 8, 0, &kp, 1, &row, 1, offs, &n_rows, 0) == MOFREAK_ERR_BAD_ARG);
        CHECK(mofreak_extract_clips(ctx, nullptr, lens, 3, 8, 8, 0, &kp, 1, &row, 1, offs, &n_rows, 0) == MOFREAK_ERR_BAD_ARG);
        CHECK(mofreak_extract_clips(ctx, clips, lens, 3, 8, 8, 0, &kp, 1, nullptr, 1, offs, &n_rows, 0) == MOFREAK_ERR_BAD_ARG);
        CHECK(mofreak_extract_clips(ctx, clips, lens, 0, 8, 8, 0, &kp, 1, &row, 1, offs, &n_rows, 0) == MOFREAK_OK && n_rows == 0);
        CHECK(mofreak_extract_clips(ctx, clips, lens, 3, 8, 8, 0, &kp, 1, &row, 1, offs, &n_rows, 0) == MOFREAK_ERR_NO_DEVICE);
        CHECK(mofreak_extract_stream_pipelined(ctx, frame, 1, 8, 8, 0, &kp, 1, &row, 1, &n_rows) == MOFREAK_OK && n_rows == 0);  // T <= gap: no rows, no device needed
        uint16_t pos[320];
        int32_t n_pos = 0;
        for (int L = 1; L <= 16; ++L) CHECK(mofreak_table_mip_positions(ctx, L, pos, &n_pos) == MOFREAK_OK && n_pos == 300);
        CHECK(mofreak_table_mip_positions(ctx, 17, pos, &n_pos) == MOFREAK_ERR_BAD_ARG);
        p.brisk_fp_model = 2;
        mofreak_ctx *bad = nullptr;
        CHECK(mofreak_create(MOFREAK_TABLES_ONLY, &p, &bad) == MOFREAK_ERR_BAD_ARG && bad == nullptr);
        p.brisk_fp_model = MOFREAK_FP_SSE;
        CHECK(mofreak_create(MOFREAK_TABLES_ONLY, &p, &bad) == MOFREAK_OK && bad);
        mofreak_destroy(bad);
    }
    mofreak_destroy(ctx);
    // the facade's file format code (the reader reverses, like the reference's)
    {
        const char *path = "/tmp/mofreak_asan_selftest.mofreak";
        FILE *f = std::fopen(path, "wb");
        CHECK(f);
        std::fwrite(text.data(), 1, (size_t)need, f);
        std::fclose(f);
        MoFREAKUtilities m(MoFREAKUtilities::KTH);  // reading and writing files needs no device
        m.readMoFREAKFeatures(path, 0);
        std::deque<MoFREAKFeature> feats = m.getMoFREAKFeatures();
        CHECK(feats.size() == rows.size() && feats.front().frame_number == rows.back().frame_number);
        m.writeMoFREAKFeaturesToFile(path);
        std::remove(path);
    }
    std::puts("asan selftest ok");
    return 0;
}
