// gfx950 (CDNA4, wave64) kernels of the MoFREAK descriptor path.
//
// Integer / byte work, bandwidth- and gather-bound; no MFMA.  Compile with -ffp-contract=off: the few
// float/double expressions below restate reference expressions whose rounding is part of the result.
//
//   band_kernel<false|true> + band_scan_kernel   cv::absdiff + cv::integral (MoFREAKUtilities.cpp:413-414,
//                                                 and the integral cv::FREAK::computeImpl builds)
//   describe_kernel                               one wavefront per keypoint: cv::FREAK bytes 0..7 on the
//                                                 difference image (:427-428, :453-456) and the Motion
//                                                 Interchange Pattern bytes (:460 -> :288-325 -> :46-99)
//   compact_*                                     the stable "erase + push_back" of :436-483
#include <algorithm>

#include "device_helpers.h"

namespace mofreak {
namespace {

// ------------------------------------------------------------------------------------------------
// Integral image of |cur - prev|, banded: a workgroup owns a band of kBandGroup x kBandRows rows of one pair.
//   pass A (FINAL=false): band_totals[band] = column sums of the band's row-prefix sums
//   pass B: exclusive scan of band_totals over bands (band_scan_kernel)
//   pass C (FINAL=true):  integral rows = band carry + running column sums, 16-byte stores
// The u8 frames are read twice (1 B/px each time); the int32 integral is written once.
// ------------------------------------------------------------------------------------------------
// COLS: column-pass iterations of 1024 columns the running sums are kept for (registers): 2 covers full HD
template <bool FINAL, int COLS>
__global__ __launch_bounds__(256) void band_kernel(IntegralArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int32_t rp[];  // [kBandRows][pitch] row-prefix sums
    if (a.gate != nullptr && *a.gate == 0) return;  // nothing was left to the gather path
    // a bounded grid walks the (pair, band) items: when the gate is shut -- dense grids of small keypoints -- only a few
    // thousand workgroups come and go instead of one per band
    for (int item = blockIdx.x; item < a.n_bands * a.n_pairs; item += gridDim.x) {
    const int band = item % a.n_bands, pair = item / a.n_bands;
    const int W = a.f.W, pitch = a.pitch;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();  // the wave index: the same in every lane, a scalar
    const uint8_t *cur = a.f.cur + (int64_t)pair * a.f.pair_stride;
    const uint8_t *prev = a.f.prev ? a.f.prev + (int64_t)pair * a.f.pair_stride : nullptr;  // null: `cur` already is the difference image
    const int64_t bt = ((int64_t)pair * a.n_bands + band) * pitch;
    int32_t *integ = a.integral + (int64_t)pair * (a.f.H + 1) * pitch;
    // a band = kBandGroup slabs of kBandRows rows, one after the other through the same LDS buffer, the running column
    // sums staying in registers: the totals that travel through memory (and the scan over them) are per band
    int4 acc[COLS];
#pragma unroll
    for (int u = 0; u < COLS; ++u) {
        const int c4 = (threadIdx.x + 256 * u) * 4;
        acc[u] = make_int4(0, 0, 0, 0);
        if (FINAL && c4 < pitch) {
            acc[u] = *reinterpret_cast<const int4 *>(a.band_totals + bt + c4);
            if (band == 0) *reinterpret_cast<int4 *>(integ + c4) = make_int4(0, 0, 0, 0);  // integral row 0
        }
    }
    for (int slab = 0; slab < kBandGroup; ++slab) {
    const int y0 = (band * kBandGroup + slab) * kBandRows;
    if (y0 >= a.f.H) break;
    const int rows = min(kBandRows, a.f.H - y0);

    for (int r = wave; r < rows; r += 4) {
        const uint8_t *c = cur + (int64_t)(y0 + r) * a.f.row_stride;
        const uint8_t *p = prev ? prev + (int64_t)(y0 + r) * a.f.row_stride : c;  // (without a previous frame: an address that is there; its bytes are not used)
        const bool diff_given = prev == nullptr;
        int32_t *out = rp + r * pitch;
        if (lane < 4) out[lane] = 0;  // physical columns 0..3; column 3 is logical column 0
        const bool aligned = (((uintptr_t)c | (uintptr_t)p) & 3) == 0;
        int carry = 0;
        constexpr int kPre = 8;  // a row in chunks of 8 steps of 256 pixels: the chunk's loads are requested together
        for (int xc = 0; xc < W; xc += 256 * kPre) {
            uint32_t cvv[kPre], pvv[kPre];
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const int x = xc + 256 * u + lane * 4;
                const bool fast = aligned && x + 3 < W;
                const int xs = fast ? x : 0;  // an address that is always there; the slow lanes fill in below
                uint32_t cv = aligned ? *reinterpret_cast<const uint32_t *>(c + xs) : 0u;
                uint32_t pv = aligned && !diff_given ? *reinterpret_cast<const uint32_t *>(p + xs) : 0u;
                if (!fast) {
                    cv = 0;
                    pv = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) {
                            cv |= (uint32_t)c[x + k] << (8 * k);
                            pv |= (uint32_t)p[x + k] << (8 * k);
                        }
                }
                cvv[u] = cv;
                pvv[u] = diff_given ? 0u : pv;
            }
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const int x = xc + 256 * u + lane * 4;
                if (xc + 256 * u >= W) break;  // wave-uniform
                const uint32_t cv = cvv[u], pv = pvv[u];
                const int p0 = absdiff_u8(cv, pv, 0);
                const int p1 = p0 + absdiff_u8(cv, pv, 1);
                const int p2 = p1 + absdiff_u8(cv, pv, 2);
                const int p3 = p2 + absdiff_u8(cv, pv, 3);
                const int incl = wave_inclusive_scan(p3);
                const int base = carry + incl - p3;
                if (x < W) *reinterpret_cast<int4 *>(out + x + 4) = make_int4(base + p0, base + p1, base + p2, base + p3);
                carry += __shfl(incl, 63);
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int u = 0; u < COLS; ++u) {
        const int c4 = (threadIdx.x + 256 * u) * 4;
        if (c4 >= pitch) continue;
        for (int r = 0; r < rows; ++r) {
            const int4 v = *reinterpret_cast<const int4 *>(rp + r * pitch + c4);
            acc[u].x += v.x;
            acc[u].y += v.y;
            acc[u].z += v.z;
            acc[u].w += v.w;
            if (FINAL) *reinterpret_cast<int4 *>(integ + (int64_t)(y0 + r + 1) * pitch + c4) = acc[u];
        }
    }
    __syncthreads();  // the next slab (or item) reuses the row-prefix buffer
    }
    if (!FINAL) {
#pragma unroll
        for (int u = 0; u < COLS; ++u) {
            const int c4 = (threadIdx.x + 256 * u) * 4;
            if (c4 < pitch) *reinterpret_cast<int4 *>(a.band_totals + bt + c4) = acc[u];
        }
    }
    }
}

__global__ __launch_bounds__(256) void band_scan_kernel(int32_t *band_totals, int pitch, int n_bands, const int32_t *gate)
{
    if (gate != nullptr && *gate == 0) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= pitch) return;
    int32_t *t = band_totals + (int64_t)blockIdx.y * n_bands * pitch + c;
    int acc = 0;
    constexpr int kPre = 16;  // bands whose totals are requested together
    for (int b0 = 0; b0 < n_bands; b0 += kPre) {
        int v[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) v[u] = b0 + u < n_bands ? t[(int64_t)(b0 + u) * pitch] : 0;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            if (b0 + u < n_bands) t[(int64_t)(b0 + u) * pitch] = acc;
            acc += v[u];
        }
    }
}

__global__ __launch_bounds__(256) void unpack_integral_kernel(const int32_t *src, int pitch, int W, int H,
                                                              int32_t *dst)
{
    const int64_t n = (int64_t)(W + 1) * (H + 1);
    const int pair = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / (W + 1)), x = (int)(i - (int64_t)y * (W + 1));
        dst[(int64_t)pair * n + i] = src[((int64_t)pair * (H + 1) + y) * pitch + x + kIntegralColOffset];
    }
}

// ------------------------------------------------------------------------------------------------
// cv::cvtColor(CV_BGR2GRAY) on 8UC3 (MoFREAKUtilities.cpp:395, :410): Y = (1868 B + 9617 G + 4899 R + 8192) >> 14.
// A streaming kernel (3 B in, 1 B out per pixel): a workgroup takes 4096 pixels of one row, pulls its 12 KB in with
// fully coalesced 16-byte loads into LDS, and every thread then converts 16 pixels and stores 16 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int kGrayPxPerBlock = 4096;

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t *bgr, int W, int H, int64_t row_stride, int64_t frame_stride,
                                                       uint8_t *gray, int chunks_per_row)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[kGrayPxPerBlock * 3];
    const int chunk = blockIdx.x % chunks_per_row, y = blockIdx.x / chunks_per_row, frame = blockIdx.y;
    const int x0 = chunk * kGrayPxPerBlock;
    const int npx = min(kGrayPxPerBlock, W - x0);
    const uint8_t *src = bgr + (int64_t)frame * frame_stride + (int64_t)y * row_stride + (int64_t)x0 * 3;
    uint8_t *dst = gray + ((int64_t)frame * H + y) * W + x0;
    const int nbytes = npx * 3;
    if (((uintptr_t)src & 15) == 0) {
        for (int i = threadIdx.x * 16; i < nbytes; i += 256 * 16) {
            if (i + 16 <= nbytes)
                *reinterpret_cast<uint4 *>(buf + i) = *reinterpret_cast<const uint4 *>(src + i);
            else
                for (int k = i; k < nbytes; ++k) buf[k] = src[k];
        }
    } else {
        for (int i = threadIdx.x; i < nbytes; i += 256) buf[i] = src[i];
    }
    __syncthreads();
    const int p0 = threadIdx.x * 16;
    if (p0 >= npx) return;
    uint32_t out[4] = {0, 0, 0, 0};
    if (p0 + 16 <= npx) {
        const uint4 a = *reinterpret_cast<const uint4 *>(buf + p0 * 3), b = *reinterpret_cast<const uint4 *>(buf + p0 * 3 + 16),
                    c = *reinterpret_cast<const uint4 *>(buf + p0 * 3 + 32);
        const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = 3 * k;
            const uint32_t B = (w[o >> 2] >> (8 * (o & 3))) & 0xff, G = (w[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 0xff,
                           R = (w[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 0xff;
            out[k >> 2] |= ((B * 1868u + G * 9617u + R * 4899u + 8192u) >> 14) << (8 * (k & 3));
        }
        if (((uintptr_t)(dst + p0) & 15) == 0) {
            *reinterpret_cast<uint4 *>(dst + p0) = make_uint4(out[0], out[1], out[2], out[3]);
            return;
        }
        for (int k = 0; k < 16; ++k) dst[p0 + k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
        return;
    }
    for (int k = 0; p0 + k < npx; ++k) {
        const uint8_t *q = buf + (p0 + k) * 3;
        dst[p0 + k] = (uint8_t)((q[0] * 1868u + q[1] * 9617u + q[2] * 4899u + 8192u) >> 14);
    }
}

struct WaveScratch {
    uint2 tx[20];  // per output column: .x = ofs | ofs1 << 16, .y = c0 | c1 << 16 (the weights as v_dot2_u32_u16 takes them)
    uint4 ty[20];  // per output row: .x = ofs | ofs1 << 16, .y = c0 << 12, .z = c1 << 12 (resize_y's pre-shifted weights)
    uint32_t p19[2][kP19Pad / 4];
};

// The pair a keypoint belongs to: the last p in [0, n_pairs) with offsets[p] <= g (offsets ascend, offsets[0] <= g).  The
// wave looks at 64 offsets per step, all steps' loads independent -- a binary search would be a chain of dependent
// memory round trips in front of everything else the keypoint needs.
__device__ __forceinline__ int pair_of(const int64_t *offsets, int n_pairs, int64_t g, int lane)
{
    int below = 0;
    for (int b = 0; b < n_pairs; b += 64) {
        const int i = b + lane;
        const bool le = i < n_pairs && offsets[min(i, n_pairs - 1)] <= g;
        below += __popcll(__ballot(le));
    }
    return max(below - 1, 0);
}

// ------------------------------------------------------------------------------------------------
// describe_kernel: one wavefront per keypoint instance (grid-stride over the chunk's instances).
// DUMP: also the two whole 19x19 buffers -> a.out_roi19 (component tests); the product instantiation has none of it.
// ------------------------------------------------------------------------------------------------
template <bool DUMP>
__global__ __launch_bounds__(256, 5) void describe_kernel(DescribeArgs a)
{
    __shared__ __attribute__((aligned(16))) SmallTables st;
    __shared__ __attribute__((aligned(16))) WaveScratch scratch[4];
    __shared__ __attribute__((aligned(16))) ThetaBound s_theta[kThetaBounds];  // one memory round trip less per keypoint

    for (int i = threadIdx.x; i < (int)(sizeof(SmallTables) / 4); i += 256)
        reinterpret_cast<int32_t *>(&st)[i] = reinterpret_cast<const int32_t *>(a.small)[i];
    for (int i = threadIdx.x; i < kThetaBounds; i += 256) s_theta[i] = a.theta[i];
    __syncthreads();

    const int lane = lane_id();
    // the same in every lane: in a scalar register, and with it the item index, the keypoint's id and record, its pair,
    // its ROI -- everything below that is per keypoint rather than per lane
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveScratch &ws = scratch[wave_in_block];
    const int W = a.f.W, H = a.f.H;
    // Per-lane constants of the MIP part.  Only the positions motionInterchangePattern reads are resampled (225 of the
    // previous buffer, 51 of the current one: st.mip_prev / st.mip_cur, ascending): four passes over the first list, one
    // over the second; a lane's position in each: column | row << 8, or -1.
    constexpr int kPrevPasses = 4, kMipPasses = kPrevPasses + 1;  // st.mip_n_prev <= 256, st.mip_n_cur <= 64 (SmallTables)
    int mip_at[kMipPasses];
#pragma unroll
    for (int u = 0; u < kMipPasses; ++u) {
        const int j = lane + 64 * u, n = u < kPrevPasses ? st.mip_n_prev : st.mip_n_cur;
        const int o = u < kPrevPasses ? (int)st.mip_prev[min(j, 255)] : (int)st.mip_cur[lane];
        const int dy = (o * 27) >> 9;  // o / 19 for o < 361
        mip_at[u] = (u < kPrevPasses ? j : lane) < n ? (o - dy * kPatch) | dy << 8 : -1;
    }
    const MipStripLane strips = mip_strip_lane();
    const int stride = (int)a.f.row_stride;  // (the context takes frames of < 2^31 bytes and rows of < 2^23: 32-bit offsets, 24-bit factors)

    // gather path behind the tile kernel: the instances are the binning pass's slow list (device-resident count)
    int64_t n_slow = a.slow_list ? (int64_t)*a.slow_count : 0, slow_lo = 0;
    if (a.slow_list && a.kp_offsets) {  // the list is in (pair, band) order: this chunk's pairs own one piece of it
        const int32_t *b = a.band_start;
        slow_lo = b[a.first_pair * a.bands_per_pair] - b[0];
        n_slow = b[(a.first_pair + a.n_pairs) * a.bands_per_pair] - b[0] - slow_lo;
    }
    const int64_t n_items = a.slow_list ? (a.kp_offsets ? n_slow : n_slow * a.n_pairs) : a.n_items;

    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own: every XCD takes one contiguous eighth
    // of the items, so that the wavefronts in flight behind one L2 work on neighbouring keypoints -- neighbouring rows of
    // one pair's integral, a band that fits that L2 -- instead of on all pairs at once (launch_describe rounds the grid
    // up to a multiple of 8).
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_items + 7) / 8, xcd_end = min(n_items, (xcd + 1) * per_xcd);
    const int64_t xcd_waves = (int64_t)(gridDim.x >> 3) * 4;
    // Which keypoint of which pair an item is takes two dependent memory round trips (list entry -> keypoint record and
    // pair offsets) before the keypoint's own work can start: they are taken ahead of time -- the list entry two items
    // ahead, the record one item ahead -- so that a wavefront's chain per keypoint is only the keypoint's own.
    struct ItemId {
        int64_t g, out_idx;
        int pair_local;  // -1: the pair is found from kp_offsets (CSR)
        bool live;
    };
    struct ItemKp {
        mofreak_keypoint kp;
        int pair_local;
        bool live;
    };
    auto item_id = [&](int64_t item) -> ItemId {
        ItemId r{0, 0, -1, item < xcd_end};
        if (!r.live) return r;
        if (a.slow_list != nullptr) {
            if (a.kp_offsets == nullptr) {
                r.pair_local = (int)(item / n_slow);
                r.g = a.slow_list[item - (int64_t)r.pair_local * n_slow];
                r.out_idx = (a.first_pair + r.pair_local) * a.n_kp + r.g;
            } else {
                r.g = a.slow_list[slow_lo + item];
                r.out_idx = r.g;
            }
        } else if (a.kp_offsets == nullptr) {
            r.pair_local = (int)((uint64_t)item / (uint64_t)a.n_kp);
            r.g = item - (int64_t)r.pair_local * a.n_kp;
            r.out_idx = a.item_base + item;
        } else {
            r.g = a.item_base + item;
            r.out_idx = r.g;
        }
        return r;
    };
    auto item_kp = [&](const ItemId &id) -> ItemKp {
        ItemKp r{mofreak_keypoint{0.f, 0.f, 0.f}, id.pair_local, id.live};
        if (!id.live) return r;
        r.kp = a.kps[id.g];
        if (id.pair_local < 0) {
            if (a.slow_list != nullptr) {
                const int lo = pair_of(a.kp_offsets, (int)a.n_pairs_total, id.g, lane);  // pair within the whole call
                r.live = lo >= a.first_pair && lo < a.first_pair + a.n_pairs;              // keypoints of other chunks: skipped
                r.pair_local = lo - (int)a.first_pair;
            } else {
                r.pair_local = pair_of(a.kp_offsets + a.first_pair, a.n_pairs, id.g, lane);  // kp_offsets[first_pair+lo] <= g < kp_offsets[first_pair+lo+1]
            }
        }
        return r;
    };
    const int64_t item0 = xcd * per_xcd + (int64_t)(blockIdx.x >> 3) * 4 + wave_in_block;
    ItemId id_next = item_id(item0), id_next2 = item_id(item0 + xcd_waves);
    ItemKp kp_next = item_kp(id_next);
    for (int64_t item = item0; item < xcd_end; item += xcd_waves) {
        const ItemId id = id_next;
        const ItemKp ik = kp_next;
        id_next = id_next2;
        id_next2 = item_id(item + 2 * xcd_waves);
        kp_next = item_kp(id_next);
        if (!ik.live) continue;
        const int64_t out_idx = id.out_idx;
        const int pair_local = ik.pair_local;
        const mofreak_keypoint kp = ik.kp;
        const float kx = kp.x, ky = kp.y, size = kp.size;

        // ---- DescriptorExtractor::compute + FREAK::computeImpl keypoint filter
        bool ok = (size >= FLT_EPSILON) && (size <= FLT_MAX) && (fabsf(kx) <= FLT_MAX) && (fabsf(ky) <= FLT_MAX);
        int idx;
        if (st.scale_normalized) {
            const bool ge = (lane < kNbScales - 1) && (size >= st.scale_thresholds[lane]);
            idx = __popcll(__ballot(ge));
        } else {
            idx = st.fixed_scale_index;
        }
        const int ps = st.pattern_sizes[idx];
        if (kx <= ps || ky <= ps || kx >= W - ps || ky >= H - ps) ok = false;

        uint64_t app = 0, mot = 0;
        int theta = -1, direction0 = 0, direction1 = 0;
        const uint8_t *cur = a.f.cur + (int64_t)pair_local * a.f.pair_stride;
        const uint8_t *prev = a.f.prev + (int64_t)pair_local * a.f.pair_stride;
        const int x_i = (int)kx, y_i = (int)ky;                  // :460 float -> int parameters
        const int tl_x = x_i - ((int)size) / 2, tl_y = y_i - ((int)size) / 2;  // :293-294
        const int L = (int)ceilf(size);                                          // :295
        if (ok) {
            if (L > kMaxRoiSide) {
                if (lane == 0) atomicOr(a.status, 2);
                ok = false;
            } else if (tl_x < 0 || tl_y < 0 || tl_x + L > W || tl_y + L > H) {
                if (lane == 0) atomicOr(a.status, 1);  // the reference throws from cv::Mat::operator()(Rect)
                ok = false;
            }
        }

        if (ok) {
            // the 19 + 19 resize taps of this ROI side: asked for now, parked in LDS when the MIP part gets to them
            ResizeTap my_tap{0, 0, 0, 0};
            if (lane < 2 * kPatch) my_tap = a.resize[L * 2 * kPatch + lane];
            // ================= FREAK on the difference image (through its integral)
            const int32_t *integ = a.integral + (int64_t)pair_local * (H + 1) * a.pitch;
            const PatternPoint *lut_scale = a.lut + (int64_t)idx * kNbOrientation * kNbPoints;
            theta = 0;
            if (st.orientation_normalized) {
                int v0 = 0;
                if ((st.need_orient >> lane) & 1) v0 = mean_intensity(integ, a.pitch, kx, ky, lut_scale[lane]);  // (bits 43.. are clear)
                int t0 = 0, t1 = 0;
                {
                    const OrientPair op = st.orient[lane < kNbOrientPairs ? lane : 0];
                    const int delta = __shfl(v0, op.i) - __shfl(v0, op.j);
                    if (lane < kNbOrientPairs) {
                        t0 = delta * op.weight_dx / 2048;  // C division: truncates toward zero, per term
                        t1 = delta * op.weight_dy / 2048;
                    }
                }
                direction0 = wave_sum(t0);
                direction1 = wave_sum(t1);
                theta = theta_index(s_theta, direction0, direction1);
            }
            int v = 0;
            if ((st.need_bits >> lane) & 1) v = mean_intensity(integ, a.pitch, kx, ky, lut_scale[theta * kNbPoints + lane]);
            {
                const int va = __shfl(v, (int)st.bit_pair_i[lane]);
                const int vb = __shfl(v, (int)st.bit_pair_j[lane]);
                bool bit;
                if (st.bit_mode == MOFREAK_BITS_SSE)
                    bit = va >= vb;
                else if (st.bit_mode == MOFREAK_BITS_NATURAL)
                    bit = va > vb;
                else
                    bit = (int)(int8_t)va > (int)(int8_t)vb;
                app = __ballot(bit);
            }

            // ================= MIP on (current, previous) gray frames
            // cv::resize(ROI -> 19x19, INTER_LINEAR) of both frames straight from memory: an output pixel's two taps of a
            // row are neighbouring bytes (or the same byte at the right edge), so one unaligned 4-byte load per row and
            // frame brings them (a load that would run past the end of its frame row starts up to 3 bytes early instead);
            // v_perm picks the two bytes, the horizontal step is a packed dot product, the vertical one resize_y -- the
            // tile kernel's arithmetic.  All of a lane's loads (two per position, five positions) are in flight together.
            {
                const uint32_t ofsw = (uint32_t)(uint16_t)my_tap.ofs | (uint32_t)(uint16_t)my_tap.ofs1 << 16;
                if (lane < kPatch)
                    ws.tx[lane] = make_uint2(ofsw, (uint32_t)(uint16_t)my_tap.c0 | (uint32_t)(uint16_t)my_tap.c1 << 16);
                else if (lane < 2 * kPatch)
                    ws.ty[lane - kPatch] = make_uint4(ofsw, (uint32_t)my_tap.c0 << 12, (uint32_t)my_tap.c1 << 12, 0u);
            }
            wave_lds_sync();
            struct Sample {
                uint32_t q0, q1, sel;  // the two rows' dwords; v_perm selector: byte of the left tap | byte of the right tap << 16
            };
            auto ask = [&](const uint8_t *frame, int at) -> Sample {
                const int dxy = max(at, 0);
                const uint2 tx = ws.tx[dxy & 0xff];
                const uint32_t ty = ws.ty[dxy >> 8].x;
                const int ofs = (int)(tx.x & 0xffff), col = tl_x + ofs, xs = min(col, W - 4);
                const uint32_t left = (uint32_t)(col - xs), right = left + (tx.x >> 16) - (uint32_t)ofs;
                Sample r;
                r.sel = left | right << 16 | 0x0c000c00u;
                __builtin_memcpy(&r.q0, frame + (uint32_t)(__mul24(tl_y + (int)(ty & 0xffff), stride) + xs), 4);
                __builtin_memcpy(&r.q1, frame + (uint32_t)(__mul24(tl_y + (int)(ty >> 16), stride) + xs), 4);
                return r;
            };
            auto put = [&](const Sample &r, int at, uint32_t *dst) {  // the weights again from LDS: cheaper than keeping them while the loads fly
                const int dxy = max(at, 0);
                const u16x2 wx = __builtin_bit_cast(u16x2, ws.tx[dxy & 0xff].y);
                const uint4 ty = ws.ty[dxy >> 8];
                const uint32_t t0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, r.q0, r.sel)), wx, 0u, false);
                const uint32_t t1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, r.q1, r.sel)), wx, 0u, false);
                if (at >= 0) reinterpret_cast<uint8_t *>(dst)[(dxy >> 8) * kPatch + (dxy & 0xff)] = (uint8_t)resize_y(t0, t1, ty.y, ty.z);
            };
            if (!DUMP) {
                Sample sm[kMipPasses];
#pragma unroll
                for (int u = 0; u < kMipPasses; ++u) sm[u] = ask(u < kPrevPasses ? prev : cur, mip_at[u]);
#pragma unroll
                for (int u = 0; u < kMipPasses; ++u) put(sm[u], mip_at[u], ws.p19[u < kPrevPasses ? 1 : 0]);
            } else {  // all 361 positions of both buffers
                for (int j = lane; j < kPatch * kPatch; j += 64) {
                    const int dy = (j * 27) >> 9, at = (j - dy * kPatch) | dy << 8;
                    const Sample s1 = ask(prev, at), s0 = ask(cur, at);
                    put(s1, at, ws.p19[1]);
                    put(s0, at, ws.p19[0]);
                }
            }
            wave_lds_sync();
            mot = mip_bits_strips(ws.p19[0], ws.p19[1], strips, st.mip_theta);
            if (DUMP) {
                uint8_t *dst = a.out_roi19 + out_idx * (2 * kPatch * kPatch);
                for (int o = lane; o < 2 * kPatch * kPatch; o += 64)
                    dst[o] = o < kPatch * kPatch ? reinterpret_cast<const uint8_t *>(ws.p19[0])[o] : reinterpret_cast<const uint8_t *>(ws.p19[1])[o - kPatch * kPatch];
            }
            wave_lds_sync();  // the next keypoint of this wave overwrites the scratch
        } else if (DUMP) {
            uint8_t *dst = a.out_roi19 + out_idx * (2 * kPatch * kPatch);
            for (int o = lane; o < 2 * kPatch * kPatch; o += 64) dst[o] = 0;
        }

        if (lane == 0) {
            uint4 d;
            d.x = (uint32_t)app;
            d.y = (uint32_t)(app >> 32);
            d.z = (uint32_t)mot;
            d.w = (uint32_t)(mot >> 32);
            *reinterpret_cast<uint4 *>(a.out_desc + out_idx * 16) = d;
            a.out_valid[out_idx] = ok ? 1 : 0;
            if (a.out_info != nullptr)
                *reinterpret_cast<int4 *>(a.out_info + out_idx * 4) = make_int4(idx, ok ? theta : -1, direction0, direction1);
        }
    }
}

__global__ __launch_bounds__(256) void mip19_kernel(const uint8_t *cur19, const uint8_t *prev19, int64_t n,
                                                    int mip_theta, uint8_t *out)
{
    __shared__ uint8_t buf[4][2][kP19Pad];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int64_t item = (int64_t)blockIdx.x * 4 + w; item < n; item += (int64_t)gridDim.x * 4) {
        for (int o = lane; o < kPatch * kPatch; o += 64) {
            buf[w][0][o] = cur19[item * (kPatch * kPatch) + o];
            buf[w][1][o] = prev19[item * (kPatch * kPatch) + o];
        }
        wave_lds_sync();
        const uint64_t m = mip_bits(buf[w][0], buf[w][1], mip_theta);
        if (lane < 8) out[item * 8 + lane] = (uint8_t)(m >> (8 * lane));
        wave_lds_sync();
    }
}

__global__ __launch_bounds__(256) void theta_kernel(const ThetaBound *tb, const int32_t *dirs, int64_t n, int32_t *out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = theta_index(tb, dirs[2 * i], dirs[2 * i + 1]);
}

// ------------------------------------------------------------------------------------------------
// Stable compaction of valid descriptors into 32-byte rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan_256(int v, int *total)
{
    __shared__ int wave_tot[4];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int incl = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wave_tot[i];
    *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return base + incl - v;
}

// Is item `item` a row?  With pair labels (many clips in one call) a pair that straddles two clips has a negative label and
// none of its keypoints is one.
__device__ __forceinline__ bool compact_keeps(const CompactArgs &a, int64_t item, int *pair_out)
{
    if (item >= a.n_items || a.valid[item] == 0) return false;
    if (a.pair_label == nullptr && a.pair_rows == nullptr) return true;
    int pair;
    if (a.kp_offsets == nullptr) {
        pair = (int)(item / a.n_kp);  // shared keypoint list: n_kp items per pair
    } else {  // per-pair lists (a detector's keypoints): the pair whose segment holds the item
        int lo = 0, hi = a.n_pairs;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.kp_offsets[mid] <= item)
                lo = mid;
            else
                hi = mid;
        }
        pair = lo;
    }
    *pair_out = pair;
    return a.pair_label == nullptr || a.pair_label[pair] >= 0;
}

__global__ __launch_bounds__(256) void compact_count_kernel(CompactArgs a)
{
    const int64_t base = (int64_t)blockIdx.x * kCompactItemsPerBlock + threadIdx.x * 4;
    int cnt = 0;
    int first_pair = -1, last_pair = -1, first_cnt = 0, last_cnt = 0;  // a thread's four items lie in at most two pairs (n_kp >= 4) ...
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int pair = -1;
        const bool keep = compact_keeps(a, base + k, &pair);
        cnt += keep;
        if (a.pair_rows != nullptr && keep) {
            if (first_pair < 0 || pair == first_pair) {
                first_pair = pair;
                ++first_cnt;
            } else if (last_pair < 0 || pair == last_pair) {
                last_pair = pair;
                ++last_cnt;
            } else {  // ... unless the list is shorter than that: count the item on its own
                atomicAdd(a.pair_rows + pair, 1);
            }
        }
    }
    if (a.pair_rows != nullptr) {
        // Rows per pair: a wave's 256 items span few pairs, so one atomic per distinct pair and wave, not per thread
        // (atomics on one address are served one after the other).
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int my_pair = side ? last_pair : first_pair, my_cnt = side ? last_cnt : first_cnt;
            for (unsigned long long todo = __ballot(my_pair >= 0); todo;) {
                const int leader = __ffsll((long long)todo) - 1;
                const int pr = __shfl(my_pair, leader);
                const bool mine = my_pair == pr;
                const int sum = wave_sum(mine ? my_cnt : 0);
                if (lane_id() == leader) atomicAdd(a.pair_rows + pr, sum);
                todo &= ~__ballot(mine);
            }
        }
    }
    int total;
    block_exclusive_scan_256(cnt, &total);
    if (threadIdx.x == 0) a.block_offsets[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void compact_scan_kernel(int64_t *block_offsets, int n_blocks, int64_t *row_base)
{
    // single workgroup: exclusive scan of the per-block counts; block_offsets[n_blocks] = total.  With row_base the
    // positions start behind the rows of earlier launches (*row_base), which this launch's total is added to.
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = row_base ? *row_base : 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_blocks; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < n_blocks ? (int)block_offsets[b] : 0;
        int total;
        const int excl = block_exclusive_scan_256(v, &total);
        const int64_t carry = carry_s;
        if (b < n_blocks) block_offsets[b] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        block_offsets[n_blocks] = carry_s;
        if (row_base) *row_base = carry_s;
    }
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(CompactArgs a)
{
    const int64_t base = (int64_t)blockIdx.x * kCompactItemsPerBlock + threadIdx.x * 4;
    int flags[4], cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int unused;
        flags[k] = compact_keeps(a, base + k, &unused);
        cnt += flags[k];
    }
    int total;
    int64_t pos = a.block_offsets[blockIdx.x] + block_exclusive_scan_256(cnt, &total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!flags[k]) continue;
        const int64_t item = base + k;
        int64_t g;
        int pair;
        if (a.kp_offsets == nullptr) {
            pair = (int)(item / a.n_kp);
            g = item - (int64_t)pair * a.n_kp;
        } else {
            g = item;
            int lo = 0, hi = a.n_pairs;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (a.kp_offsets[mid] <= g)
                    lo = mid;
                else
                    hi = mid;
            }
            pair = lo;
        }
        if (pos < a.capacity) {
            const mofreak_keypoint kp = a.kps[g];
            const uint4 d = *reinterpret_cast<const uint4 *>(a.desc + item * 16);
            uint4 lo4, hi4;
            lo4.x = __float_as_uint(kp.x);
            lo4.y = __float_as_uint(kp.y);
            lo4.z = (uint32_t)(a.pair_label != nullptr ? a.pair_label[pair] : a.first_frame_number + pair);
            lo4.w = __float_as_uint(kp.size);
            hi4 = d;
            uint4 *dst = reinterpret_cast<uint4 *>(a.rows + pos);
            dst[0] = lo4;
            dst[1] = hi4;
        }
        ++pos;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int launch_integral(const IntegralArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)std::min<int64_t>((int64_t)a.n_bands * a.n_pairs, 4096));
    const size_t lds = (size_t)kBandRows * a.pitch * sizeof(int32_t);
    const auto first = a.pitch <= 2048 ? &band_kernel<false, 2> : a.pitch <= 4096 ? &band_kernel<false, 4> : &band_kernel<false, kBandColItersMax>;
    const auto final = a.pitch <= 2048 ? &band_kernel<true, 2> : a.pitch <= 4096 ? &band_kernel<true, 4> : &band_kernel<true, kBandColItersMax>;
    if (lds > 64 * 1024) {  // more than the default dynamic-LDS limit: opt in (a CU has 160 KiB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(first), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(final), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(first, grid, dim3(256), lds, s, a);
    hipLaunchKernelGGL(band_scan_kernel, dim3((a.pitch + 255) / 256, a.n_pairs), dim3(256), 0, s, a.band_totals,
                       a.pitch, a.n_bands, a.gate);
    hipLaunchKernelGGL(final, grid, dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

int launch_describe(const DescribeArgs &a, int n_blocks, void *stream)
{
    if (a.out_roi19 != nullptr)
        hipLaunchKernelGGL(describe_kernel<true>, dim3((n_blocks + 7) & ~7), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(describe_kernel<false>, dim3((n_blocks + 7) & ~7), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int launch_mip19(const uint8_t *cur19, const uint8_t *prev19, int64_t n, int mip_theta, uint8_t *out, void *stream)
{
    const int blocks = (int)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048);
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(mip19_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cur19, prev19, n,
                       mip_theta, out);
    return (int)hipGetLastError();
}

int launch_theta(const ThetaBound *tb, const int32_t *dirs, int64_t n, int32_t *out, void *stream)
{
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(theta_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tb, dirs, n, out);
    return (int)hipGetLastError();
}

int launch_compact(const CompactArgs &a, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.n_blocks > 0) {
        hipLaunchKernelGGL(compact_count_kernel, dim3(a.n_blocks), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, s, a.block_offsets, a.n_blocks, a.row_base);
    if (a.n_blocks > 0) {
        hipLaunchKernelGGL(compact_scatter_kernel, dim3(a.n_blocks), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

int launch_bgr2gray(const uint8_t *bgr, int W, int H, int64_t row_stride, int64_t frame_stride, int n_frames, uint8_t *gray,
                    void *stream)
{
    const int chunks = (W + kGrayPxPerBlock - 1) / kGrayPxPerBlock;
    for (int f0 = 0; f0 < n_frames; f0 += 32768) {
        const int nf = n_frames - f0 < 32768 ? n_frames - f0 : 32768;
        hipLaunchKernelGGL(bgr2gray_kernel, dim3((unsigned)chunks * H, nf), dim3(256), 0, static_cast<hipStream_t>(stream),
                           bgr + (int64_t)f0 * frame_stride, W, H, row_stride, frame_stride, gray + (int64_t)f0 * W * H, chunks);
    }
    return (int)hipGetLastError();
}

int launch_unpack_integral(const int32_t *src, int pitch, int W, int H, int n_pairs, int32_t *dst, void *stream)
{
    hipLaunchKernelGGL(unpack_integral_kernel, dim3(512, n_pairs), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       pitch, W, H, dst);
    return (int)hipGetLastError();
}

}  // namespace mofreak
