// The Motion Interchange Pattern of one keypoint computed by ONE LANE (MoFREAKUtilities.cpp:288-325 -> :46-99): 64 keypoints
// per wave, no LDS, no cross-lane traffic.  Used by the tile kernel for tiles whose keypoints all have the same ROI side L
// (tile_kernel.hip); compiled per L.
//
// What makes a lane-per-keypoint form cheap is that everything that is not data is the same in all 64 lanes and known at
// compile time: which source byte feeds which of the 19x19 resampled pixels (cv::resize's taps for ROI side L:
// resize_axis.h), which of those pixels motionInterchangePattern reads at all (51 of the current buffer, 225 of the
// previous one), where the sixty-four 9-byte strips start.  So a lane fetches its ROI rows from the frames with 16-byte
// loads, keeps them in registers, runs cv::resize's horizontal pass once per source row (byte pairs by v_perm with constant
// selectors, weights as scalar operands) and its vertical pass per needed pixel, packs the results into dwords in the order
// of the 19-byte-pitch buffer and takes the strips out of those with constant shifts; the previous buffer is consumed row by
// row as it is produced, so only a few of its dwords are alive at a time.  A wave-per-keypoint form (the tile kernel's
// other MIP path) spends 20 one-byte LDS reads per lane and keypoint on the same taps.
//
// Every table read below happens in a constant expression (static_for hands the loop index over as a type), so no table
// exists in device memory and no register array is ever indexed at run time.
#pragma once

#include <utility>

#include "resize_axis.h"

// The handful of gfx950 instructions the algorithm is written in.  With MOFREAK_MIP_LANE_HOST defined the same header compiles
// for the host with plain C++ in their place (tests/helpers/mip_lane_host.cpp: the per-ROI-side code paths checked against
// the oracle on the CPU, every tap and selector included, before a GPU sees them).
#ifdef MOFREAK_MIP_LANE_HOST
#include <cstring>
#define MIP_FN inline
namespace mofreak {
namespace {
struct MipUint2 {
    uint32_t x, y;
};
MIP_FN uint32_t mip_perm(uint32_t hi, uint32_t lo, uint32_t sel)  // v_perm_b32: selector byte 0..3 = lo's bytes, 4..7 = hi's, 0x0c = 0
{
    uint32_t out = 0;
    for (int b = 0; b < 4; ++b) {
        const uint32_t s = (sel >> (8 * b)) & 0xff;
        const uint32_t v = s == 0x0c ? 0u : s < 4 ? (lo >> (8 * s)) & 0xff : (hi >> (8 * (s - 4))) & 0xff;
        out |= v << (8 * b);
    }
    return out;
}
MIP_FN uint32_t mip_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (shift & 3))); }
MIP_FN uint32_t mip_udot2(uint32_t a, uint32_t b) { return (a & 0xffff) * (b & 0xffff) + (a >> 16) * (b >> 16); }
MIP_FN uint32_t mip_udot4(uint32_t a, uint32_t b, uint32_t c)
{
    for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xff) * ((b >> (8 * k)) & 0xff);
    return c;
}
MIP_FN uint32_t mip_umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
MIP_FN void mip_and_or(uint32_t &word, uint32_t v, uint32_t mask) { word |= v & mask; }
MIP_FN void mip_sched_barrier() {}
// byte B of `word` = s >> 2 (a value below 256); FIRST: the other bytes become zero, else they stay
template <int B, bool FIRST>
MIP_FN void mip_put_quarter(uint32_t &word, uint32_t s)
{
    const uint32_t v = ((s >> 2) & 0xffu) << (8 * B);
    word = FIRST ? v : ((word & ~(0xffu << (8 * B))) | v);
}
template <int N>
MIP_FN void mip_load_dwords(const uint8_t *frame, uint32_t off4, uint32_t *d)  // N dwords from frame + off4 (4-byte aligned)
{
    std::memcpy(d, frame + off4, 4 * N);
}
MIP_FN MipUint2 mip_make_uint2(uint32_t x, uint32_t y) { return MipUint2{x, y}; }
}  // namespace
}  // namespace mofreak
#else
#include "device_helpers.h"
#define MIP_FN __device__ __forceinline__
namespace mofreak {
namespace {
typedef uint2 MipUint2;
MIP_FN uint32_t mip_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
MIP_FN uint32_t mip_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
MIP_FN uint32_t mip_udot2(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), 0u, false); }
MIP_FN uint32_t mip_udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
MIP_FN uint32_t mip_umulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
// word |= v & mask as ONE instruction that depends on `word`: written as plain ORs the compiler keeps every strip's bit in a
// register of its own and joins them in trees later
MIP_FN void mip_and_or(uint32_t &word, uint32_t v, uint32_t mask) { asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(word) : "v"(v), "s"(mask)); }
MIP_FN void mip_sched_barrier() { __builtin_amdgcn_sched_barrier(0); }
// byte B of `word` = s >> 2 (a value below 256); FIRST: the other bytes become zero, else they stay.  The shift writes its
// result straight into the byte (SDWA destination select): packing four cells into a dword costs no instruction of its own.
template <int B, bool FIRST>
MIP_FN void mip_put_quarter(uint32_t &word, uint32_t s)
{
    static_assert(B >= 0 && B < 4, "a byte of a dword");
    if constexpr (FIRST) {
        if constexpr (B == 0) asm("v_lshrrev_b32_e32 %0, 2, %1" : "=v"(word) : "v"(s));
        if constexpr (B == 1) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(word) : "v"(s));
        if constexpr (B == 2) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(word) : "v"(s));
        if constexpr (B == 3) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(word) : "v"(s));
    } else {
        if constexpr (B == 0) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(s));
        if constexpr (B == 1) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(s));
        if constexpr (B == 2) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(s));
        if constexpr (B == 3) asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(s));
    }
}
typedef uint32_t MipU4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t MipU3 __attribute__((ext_vector_type(3), aligned(4)));
// N dwords from frame + off4: `frame` is the same in every lane (a scalar base), off4 the lane's 32-bit byte offset, a
// multiple of 4 -- the address costs no vector arithmetic; one 12- or 16-byte load (+ a dword)
template <int N>
MIP_FN void mip_load_dwords(const uint8_t *frame, uint32_t off4, uint32_t *d)
{
    static_assert(N >= 1 && N <= 5, "ROI sides up to 16");
    const uint8_t *p4 = frame + off4;
    if constexpr (N <= 3) {
        const MipU3 v = *reinterpret_cast<const MipU3 *>(p4);
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
    } else {
        const MipU4 v = *reinterpret_cast<const MipU4 *>(p4);
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
        d[3] = v.w;
        if constexpr (N > 4) d[4] = *reinterpret_cast<const uint32_t *>(p4 + 16);
    }
}
MIP_FN MipUint2 mip_make_uint2(uint32_t x, uint32_t y) { return make_uint2(x, y); }
}  // namespace
}  // namespace mofreak
#endif

namespace mofreak {
namespace {

template <class F, int... Is>
MIP_FN void static_for_impl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
MIP_FN void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- geometry of motionInterchangePattern: patch centres (:308-316) and the 8 offsets (:56-70), as in mip_bits()
constexpr int kMipCX[8] = {5, 5, 5, 9, 9, 13, 13, 13};
constexpr int kMipCY[8] = {5, 9, 13, 5, 13, 5, 9, 13};
constexpr int kMipDX[8] = {-4, -3, 0, 3, 4, 3, 0, -3};
constexpr int kMipDY[8] = {0, 3, 4, 3, 0, -3, -4, -3};
constexpr int kMipCells = kAxisOut * kAxisOut;        // 361
constexpr int kMipDwords = (kMipCells + 3) / 4;       // dwords of a 19x19 buffer at its 19-byte pitch

// first byte of the 9 CONTIGUOUS bytes the reference walks with p++ (not a 3x3 patch)
constexpr int mip_cur_start(int c) { return (kMipCY[c] - 1) * kAxisOut + (kMipCX[c] - 1); }
constexpr int mip_prev_start(int c, int o) { return (kMipCY[c] + kMipDY[o] - 1) * kAxisOut + (kMipCX[c] + kMipDX[o] - 1); }

// What one frame's resample has to produce for ROI side L, and from what.
struct MipPlan {
    bool need[kMipCells];         // the MIP reads this cell of the 19x19 buffer
    uint32_t row_mask[kAxisOut];  // per output row: bit dx = cell (dy, dx) is needed
    uint32_t src_mask[16];        // per source row r: the output columns whose horizontal sums T[r][dx] some needed cell takes
    int n_rows;                   // source rows with a non-empty mask, ascending
    int rows[16];
};

template <int L, bool PREV, int CM>
constexpr MipPlan make_mip_plan()
{
    MipPlan p{};
    for (int c = 0; c < 8; ++c) {
        if (!((CM >> c) & 1)) continue;
        if (!PREV) {
            for (int k = 0; k < 9; ++k) p.need[mip_cur_start(c) + k] = true;
        } else {
            for (int o = 0; o < 8; ++o)
                for (int k = 0; k < 9; ++k) p.need[mip_prev_start(c, o) + k] = true;
        }
    }
    const ResizeAxisC Y = make_resize_axis(L, false);
    for (int dy = 0; dy < kAxisOut; ++dy) {
        for (int dx = 0; dx < kAxisOut; ++dx)
            if (p.need[dy * kAxisOut + dx]) p.row_mask[dy] |= 1u << dx;
        p.src_mask[Y.ofs[dy]] |= p.row_mask[dy];
        p.src_mask[Y.ofs1[dy]] |= p.row_mask[dy];
    }
    for (int r = 0; r < L; ++r)
        if (p.src_mask[r]) p.rows[p.n_rows++] = r;
    return p;
}

template <int L, bool PREV, int CM>
struct MipTables {
    static constexpr ResizeAxisC X = make_resize_axis(L, true);
    static constexpr ResizeAxisC Y = make_resize_axis(L, false);
    static constexpr MipPlan P = make_mip_plan<L, PREV, CM>();
};

// Cells are produced source row by source row, inside one column by column, inside a column output row by output row: is
// `pos` the first needed cell of its dword to be produced?  (It assigns the dword, the others OR into it; two output rows that
// hang on the same source row share dwords at the 19-byte pitch, and there the higher position can come first.)
template <int L, bool PREV, int CM>
constexpr bool mip_first_of_dword(int pos)
{
    const ResizeAxisC &Y = MipTables<L, PREV, CM>::Y;
    const MipPlan &P = MipTables<L, PREV, CM>::P;
    const int dy = pos / kAxisOut, dx = pos % kAxisOut;
    const int key = ((Y.ofs[dy] > Y.ofs1[dy] ? Y.ofs[dy] : Y.ofs1[dy]) * kAxisOut + dx) * kAxisOut + dy;
    for (int q = pos - (pos & 3); q < pos - (pos & 3) + 4 && q < kMipCells; ++q) {
        if (q == pos || !P.need[q]) continue;
        const int qy = q / kAxisOut, qx = q % kAxisOut;
        const int qkey = ((Y.ofs[qy] > Y.ofs1[qy] ? Y.ofs[qy] : Y.ofs1[qy]) * kAxisOut + qx) * kAxisOut + qy;
        if (qkey < key) return false;
    }
    return true;
}

constexpr int mip_row_dwords(int L) { return (L + 3) / 4; }      // aligned dwords that hold the L bytes of a ROI row
constexpr int mip_raw_dwords(int L) { return (L + 3 + 3) / 4; }  // dwords fetched for it: the row starts at any byte of the first

struct MipRawRow {
    uint32_t d[6];  // at least mip_raw_dwords(L) of them are fetched; the aligning step names one dword beyond the last needed one and
                    // never uses its bytes
};

// the dwords that cover one ROI row: `row` = the frame row's address (the same in every lane), off4 = the ROI's byte offset
// in it rounded down to 4 bytes
template <int L>
MIP_FN MipRawRow mip_load_row(const uint8_t *row, uint32_t off4)
{
    MipRawRow r;
    r.d[0] = r.d[1] = r.d[2] = r.d[3] = r.d[4] = r.d[5] = 0;
    mip_load_dwords<(mip_raw_dwords(L) < 3 ? 3 : mip_raw_dwords(L))>(row, off4, r.d);
    return r;
}

// cv::resize's two passes over one frame's ROI for the cells the MIP reads.  OUT[k] receives bytes 4k .. 4k+3 of the 19x19
// buffer (only needed cells are written; a dword's first needed cell assigns it).  row_done(dy) is called when output row dy
// is complete (all needed cells of rows <= dy are in OUT).
//   frame: the frame (a scalar); off4: byte offset of the ROI's top-left pixel in it, rounded down to 4 bytes; shift: the
//   bytes dropped by that (the same in every row: the row stride is a multiple of 4)
template <int L, bool PREV, int CM, class RowDone>
MIP_FN void mip_resample(const uint8_t *frame, uint32_t off4, uint32_t shift, int64_t row_stride, uint32_t (&OUT)[kMipDwords], RowDone &&row_done)
{
    using TB = MipTables<L, PREV, CM>;
    constexpr int kAhead = 2;  // source rows requested ahead of the one being worked on
    constexpr int kRing = kAhead + 1;
    constexpr int n_rows = TB::P.n_rows;
    MipRawRow raw[kRing];
    uint32_t T[kAxisOut];  // horizontal sums (low four bits cleared) of the source row above the one being worked on
    static_for<(kAhead < n_rows ? kAhead : n_rows)>([&](auto I) {
        constexpr int i = decltype(I)::value;
        raw[i % kRing] = mip_load_row<L>(frame + (int64_t)TB::P.rows[i] * row_stride, off4);
    });
    static_for<n_rows>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int r = TB::P.rows[i];
        if constexpr (i + kAhead < n_rows) raw[(i + kAhead) % kRing] = mip_load_row<L>(frame + (int64_t)TB::P.rows[i + kAhead] * row_stride, off4);
        // the row's bytes 4k .. 4k+3
        uint32_t W[mip_row_dwords(L)];
        static_for<mip_row_dwords(L)>([&](auto K) {
            constexpr int k = decltype(K)::value;
            W[k] = mip_alignbyte(raw[i % kRing].d[k + 1], raw[i % kRing].d[k], shift);
        });
        // Column by column: the horizontal sum of this source row (HResizeLinear: T = S[sx] * a0 + S[sx1] * a1, one packed dot
        // product on a byte pair picked by a constant selector), then at once the cells of this column in the output rows whose
        // lower source row is r (VResizeLinear: ((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2, where
        // (T & ~15) * (b << 12) >> 32 is (b * (T >> 4)) >> 16), then the sum replaces that of the row above: one row of sums alive.
        constexpr uint32_t um = TB::P.src_mask[r];
        static_for<kAxisOut>([&](auto DX) {
            constexpr int dx = decltype(DX)::value;
            if constexpr ((um >> dx) & 1u) {
                constexpr int sx = TB::X.ofs[dx], sx1 = TB::X.ofs1[dx];
                constexpr int kl = sx >> 2, kh = sx1 >> 2;
                static_assert(kh == kl || kh == kl + 1, "a tap pair spans at most two dwords");
                constexpr uint32_t sel = (uint32_t)(sx & 3) | 0x0c00u | (uint32_t)((sx1 & 3) + (kh != kl ? 4 : 0)) << 16 | 0x0c000000u;
                const uint32_t pair = mip_perm(W[kh], W[kl], sel);  // S[sx] | S[sx1] << 16
                constexpr uint32_t wx = (uint32_t)TB::X.c0[dx] | (uint32_t)TB::X.c1[dx] << 16;
                const uint32_t t = mip_udot2(pair, wx) & 0x00fffff0u;
                static_for<kAxisOut>([&](auto DY) {
                    constexpr int dy = decltype(DY)::value;
                    constexpr int r0 = TB::Y.ofs[dy], r1 = TB::Y.ofs1[dy];
                    if constexpr (((TB::P.row_mask[dy] >> dx) & 1u) && (r0 > r1 ? r0 : r1) == r) {
                        static_assert(r1 == r && (r0 == r || r0 + 1 == r), "the two source rows of an output row are this one and the one above");
                        constexpr uint32_t b0 = (uint32_t)TB::Y.c0[dy] << 12, b1 = (uint32_t)TB::Y.c1[dy] << 12;
                        uint32_t s = 2u;
                        if constexpr (b0 != 0) s += mip_umulhi(r0 == r ? t : T[dx], b0);
                        if constexpr (b1 != 0) s += mip_umulhi(t, b1);
                        constexpr int pos = dy * kAxisOut + dx, k = pos >> 2, b = pos & 3;
                        mip_put_quarter<b, mip_first_of_dword<L, PREV, CM>(pos)>(OUT[k], s);  // the cell: s >> 2
                    }
                });
                T[dx] = t;
                if constexpr (dx % 2 == 1) mip_sched_barrier();  // (two columns at a time: more in flight means more registers)
            }
        });
        static_for<kAxisOut>([&](auto DY) {
            constexpr int dy = decltype(DY)::value;
            constexpr int r0 = TB::Y.ofs[dy], r1 = TB::Y.ofs1[dy];
            if constexpr (TB::P.row_mask[dy] != 0 && (r0 > r1 ? r0 : r1) == r) row_done(DY);
        });
        mip_sched_barrier();  // a source row at a time: work hoisted across rows only lengthens live ranges
    });
}

// The motion bytes of one keypoint (byte = patch centre, bit = offset: MoFREAKUtilities.cpp:79-96, 308-316) for the patch
// centres in the mask CM (bit c = centre c), packed in ascending order of c: the k-th centre of the mask is byte k & 3 of .x
// (k < 4) or .y.  CM = 0xff: .x = centres 0..3, .y = centres 4..7 -- the uint2 the tile kernel stores behind the appearance
// bytes; the tile kernel gives the centres {0, 1, 3, 5} and {2, 4, 6, 7} of a keypoint to two waves (kMipMaskA / kMipMaskB: the
// split that shares the fewest cells of the previous buffer, 137 each of the 225).
//   cur / prev: the two frames (scalars); roi: byte offset of the ROI's top-left pixel (:293-295, 303-304) in both, below
//   2^32.  Frames and row stride must be multiples of 4 bytes; mip_raw_dwords(L) dwords (at least 3) are read from each
//   row's start rounded down to 4 bytes (the caller keeps ROIs whose last row could take that read past the end of the frame
//   off this path).
constexpr int kMipMaskAll = 0xff, kMipMaskA = 0x2b, kMipMaskB = 0xd4;
constexpr int mip_rank_in_mask(int cm, int c)
{
    int k = 0;
    for (int i = 0; i < c; ++i) k += (cm >> i) & 1;
    return k;
}
template <int L, int CM = kMipMaskAll>
MIP_FN MipUint2 mip_lane_keypoint(const uint8_t *cur, const uint8_t *prev, uint32_t roi, int64_t row_stride, int mip_theta)
{
    const uint32_t shift = roi & 3u, off4 = roi - shift;

    // ---- the current buffer: the eight 9-byte strips start on dwords of the 19-byte-pitch buffer (19 * 4k and x - 1 = 4, 8,
    // 12 are multiples of 4): strip c is CW[s / 4], CW[s / 4 + 1] and the low byte of CW[s / 4 + 2]
    uint32_t CW[kMipDwords];
    mip_resample<L, false, CM>(cur, off4, shift, row_stride, CW, [](auto) {});
    uint32_t CC[8];  // sum of squares of each strip
    static_for<8>([&](auto C) {
        constexpr int c = decltype(C)::value;
        constexpr int s = mip_cur_start(c);
        static_assert((s & 3) == 0, "current strips are dword-aligned");
        if constexpr ((CM >> c) & 1) {
            const uint32_t c8 = CW[s / 4 + 2] & 0xffu;
            CC[c] = mip_udot4(CW[s / 4], CW[s / 4], mip_udot4(CW[s / 4 + 1], CW[s / 4 + 1], c8 * c8));
        }
    });

    // ---- the previous buffer, row by row; a strip's SSD as soon as its last byte exists:
    // SSD = sum c^2 + sum p^2 - 2 sum c p (packed u8 dot products over the first eight bytes + the ninth byte's terms)
    uint32_t PW[kMipDwords];
    uint32_t mot[2] = {0u, 0u};
    mip_resample<L, true, CM>(prev, off4, shift, row_stride, PW, [&](auto DY) {
        constexpr int dy = decltype(DY)::value;
        static_for<64>([&](auto CO) {
            constexpr int c = decltype(CO)::value >> 3, o = decltype(CO)::value & 7;
            constexpr int s = mip_prev_start(c, o);
            if constexpr (((CM >> c) & 1) && (s + 8) / kAxisOut == dy) {
                constexpr int k = s >> 2, sh = s & 3, q = s + 8;
                uint32_t p0, p1;
                if constexpr (sh == 0) {
                    p0 = PW[k];
                    p1 = PW[k + 1];
                } else {
                    p0 = mip_alignbyte(PW[k + 1], PW[k], (uint32_t)sh);
                    p1 = mip_alignbyte(PW[k + 2], PW[k + 1], (uint32_t)sh);
                }
                const uint32_t p8 = (PW[q >> 2] >> (8 * (q & 3))) & 0xffu;
                constexpr int cs = mip_cur_start(c) / 4;
                const uint32_t c8 = CW[cs + 2] & 0xffu;
                const uint32_t sq = mip_udot4(p0, p0, mip_udot4(p1, p1, CC[c] + p8 * p8));
                const uint32_t cross = mip_udot4(CW[cs], p0, mip_udot4(CW[cs + 1], p1, c8 * p8));
                const int d = mip_theta - (int)sq + 2 * (int)cross;  // negative <=> SSD > theta (:93)
                // the sign bit goes to bit o of the centre's byte
                constexpr int rank = mip_rank_in_mask(CM, c);
                constexpr int bit = 8 * (rank & 3) + o;
                const uint32_t moved = (uint32_t)d >> (31 - bit);
                uint32_t word = mot[rank >> 2];  // (a copy: an asm operand does not capture)
                mip_and_or(word, moved, 1u << bit);
                mot[rank >> 2] = word;
            }
        });
    });
    return mip_make_uint2(mot[0], mot[1]);
}

}  // namespace
}  // namespace mofreak
