// .mofreak text on the device: MoFREAKUtilities::writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719) for rows that are
// already in HBM, byte for byte what mofreak_format_rows (format.cpp) and the reference's `ofstream <<` give.
//
// One thread per row, three passes: (1) the row's length (the digits are counted, not made), summed per block of 256 rows;
// (2) an exclusive scan of the block sums; (3) the text itself -- a block formats its 256 rows into LDS at their offsets,
// staged so that LDS byte k and the output byte it goes to are 16-byte congruent, and copies the staging area out with
// 16-byte stores (byte stores only for the block's first and last partial chunk).  A row is ~75 characters.
//
// `ostream << float` with default flags is printf("%g"): 6 significant digits, correctly rounded (half to even on the exact
// binary value), trailing zeros dropped.  Done here in integers: a float is m * 2^e, the digits are round(m * 5^d * 2^(e + d))
// for the d decimals %g wants (m * 5^d < 2^48: exact in 64 bits).  The fixed-notation range [1e-4, 1e6) and zero are
// formatted; a row with a float outside it (%g's exponent notation), negative or not finite is reported (the call returns
// MOFREAK_ERR_UNSUPPORTED and the caller formats on the host): coordinates and sizes of keypoints are never such values.
#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kFmtBlock = 256;     // rows per block
constexpr int kFmtMaxRow = 120;    // longest row: 3 floats x 12 + frame 12 + "0 0 " + 16 x 4 + newline = 117
constexpr int kFmtStage = 16 + kFmtBlock * kFmtMaxRow;

struct FmtCount {  // counts characters
    int n = 0;
    __device__ __forceinline__ void put(char) { ++n; }
};
struct FmtStore {  // writes them to LDS
    uint8_t *p;
    int n = 0;
    __device__ __forceinline__ void put(char c) { p[n++] = (uint8_t)c; }
};

template <class E>
__device__ __forceinline__ void fmt_uint(E &e, uint32_t u)
{
    // (decimal digits of a 32-bit value, most significant first; no buffer: the count of digits first)
    int nd = 1;
    for (uint32_t t = u; t >= 10; t /= 10) ++nd;
    uint32_t pw = 1;
    for (int i = 1; i < nd; ++i) pw *= 10;
    for (int i = 0; i < nd; ++i) {
        const uint32_t d = u / pw;
        e.put((char)('0' + d));
        u -= d * pw;
        pw /= 10;
    }
}

template <class E>
__device__ __forceinline__ void fmt_byte(E &e, uint32_t v)  // 0..255
{
    const uint32_t h = (v * 41u) >> 12;  // v / 100 for v < 256
    const uint32_t r = v - 100u * h;
    const uint32_t t = (r * 205u) >> 11;  // r / 10 for r < 100
    if (h) e.put((char)('0' + h));
    if (h | t) e.put((char)('0' + t));
    e.put((char)('0' + (r - 10u * t)));
}

// printf("%g", v); returns false for a value this formatter leaves to the host
template <class E>
__device__ __forceinline__ bool fmt_float(E &e, float v)
{
    const uint32_t bits = __builtin_bit_cast(uint32_t, v);
    if (bits == 0u) {
        e.put('0');
        return true;
    }
    const uint32_t ex = (bits >> 23) & 0xffu;
    if ((bits >> 31) || ex == 255u || ex == 0u) return false;  // negative (-0 prints "-0"), inf / nan, denormal (exponent notation)
    if (v < 100000.0f && v == (float)(int)v) {  // small integral values: coordinates, sizes like 12 (format.cpp's shortcut)
        fmt_uint(e, (uint32_t)(int)v);
        return true;
    }
    const double dv = (double)v;
    // (a float never lies strictly between a power of ten and the double nearest to it: these comparisons decide as the
    // real numbers do)
    if (dv < 9.9e-5 || dv >= 1e6) return false;
    // (just below 1e-4 the six digits may still round up to 1.00000e-04, which %g prints as 0.0001: decided after the rounding)
    int X = -5 + (dv >= 1e-4) + (dv >= 1e-3) + (dv >= 1e-2) + (dv >= 1e-1) + (dv >= 1e0) + (dv >= 1e1) + (dv >= 1e2) + (dv >= 1e3) + (dv >= 1e4) + (dv >= 1e5);
    const int d = 5 - X;  // decimals of the 6-significant-digit value: 0..10
    constexpr uint64_t kPow5[11] = {1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull, 1953125ull, 9765625ull};
    uint64_t p5 = 1;
#pragma unroll
    for (int i = 0; i < 11; ++i) p5 = i == d ? kPow5[i] : p5;
    const uint64_t N = (uint64_t)((bits & 0x7fffffu) | 0x800000u) * p5;  // m * 5^d, below 2^48
    const int sh = (int)ex - 150 + d;                                     // v * 10^d = N * 2^sh
    uint64_t q;
    if (sh >= 0) {
        q = N << sh;
    } else {
        const int s = -sh;  // <= 37 for v >= 1e-4
        q = N >> s;
        const uint64_t rem = N & ((1ull << s) - 1ull), half = 1ull << (s - 1);
        if (rem > half || (rem == half && (q & 1ull))) ++q;  // round half to even on the exact value
    }
    if (q >= 1000000ull) {  // rounded up into the next decade
        q = 100000ull;
        if (++X >= 6) return false;
    }
    if (X < -4) return false;  // exponent notation
    uint32_t digit[6];
    uint32_t t = (uint32_t)q;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        const uint32_t u = t / 10u;
        digit[i] = t - 10u * u;
        t = u;
    }
    int nd = 6;  // significant digits without the trailing zeros
#pragma unroll
    for (int i = 5; i >= 1; --i)
        if (nd == i + 1 && digit[i] == 0u) nd = i;
    if (X >= 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i <= X) e.put((char)('0' + digit[i]));  // (digits beyond nd are zeros: printed as such)
        if (nd > X + 1) {
            e.put('.');
#pragma unroll
            for (int i = 1; i < 6; ++i)
                if (i > X && i < nd) e.put((char)('0' + digit[i]));
        }
    } else {
        e.put('0');
        e.put('.');
        for (int i = 0; i < -X - 1; ++i) e.put('0');
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < nd) e.put((char)('0' + digit[i]));
    }
    return true;
}

// One row: "x y frame scale 0 0 a0..a7 m0..m7 \n", every value followed by one space (format.cpp format_one).
template <class E>
__device__ __forceinline__ bool fmt_row(E &e, const mofreak_row &r)
{
    bool ok = fmt_float(e, r.x);
    e.put(' ');
    ok = fmt_float(e, r.y) && ok;
    e.put(' ');
    if (r.frame_number < 0) {
        e.put('-');
        fmt_uint(e, 0u - (uint32_t)r.frame_number);
    } else {
        fmt_uint(e, (uint32_t)r.frame_number);
    }
    e.put(' ');
    ok = fmt_float(e, r.scale) && ok;
    e.put(' ');
    e.put('0');  // motion_x, motion_y: always 0 (MoFREAKUtilities.cpp:476-477)
    e.put(' ');
    e.put('0');
    e.put(' ');
#pragma unroll
    for (int i = 0; i < MOFREAK_APPEARANCE_BYTES; ++i) {
        fmt_byte(e, r.appearance[i]);
        e.put(' ');
    }
#pragma unroll
    for (int i = 0; i < MOFREAK_MOTION_BYTES; ++i) {
        fmt_byte(e, r.motion[i]);
        e.put(' ');
    }
    e.put('\n');
    return ok;
}

__device__ __forceinline__ mofreak_row load_row(const mofreak_row *rows, int64_t i)
{
    const uint4 a = reinterpret_cast<const uint4 *>(rows + i)[0], b = reinterpret_cast<const uint4 *>(rows + i)[1];
    mofreak_row r;
    static_assert(sizeof(mofreak_row) == 32, "two 16-byte loads per row");
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    __builtin_memcpy(&r, w, 32);
    return r;
}

// pass 1: row lengths (one byte each) and their sum per block; bit 0 of *status_bits if a row cannot be formatted here
__global__ __launch_bounds__(kFmtBlock) void fmt_len_kernel(const mofreak_row *rows, int64_t n, uint8_t *len8, uint32_t *block_sum, int32_t *status_bits)
{
    const int64_t i = (int64_t)blockIdx.x * kFmtBlock + threadIdx.x;
    int len = 0;
    bool ok = true;
    if (i < n) {
        FmtCount c;
        ok = fmt_row(c, load_row(rows, i));
        len = c.n;
        len8[i] = (uint8_t)len;
    }
    if (!ok) atomicOr(status_bits, 1);
    __shared__ int wave_tot[kFmtBlock / 64];
    const int s = wave_sum(len);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kFmtBlock / 64; ++w) t += (uint32_t)wave_tot[w];
        block_sum[blockIdx.x] = t;
    }
}

// pass 2: exclusive scan of the block sums (one workgroup); block_off[n_blocks] = the text's length
__global__ __launch_bounds__(1024) void fmt_scan_kernel(const uint32_t *block_sum, int64_t n_blocks, uint64_t *block_off)
{
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int64_t b = b0 + threadIdx.x;
        const uint64_t v = b < n_blocks ? block_sum[b] : 0u;
        uint64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += t;
        }
        if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t base = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
        if (b < n_blocks) block_off[b] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_off[n_blocks] = carry;
}

// pass 3: the text
__global__ __launch_bounds__(kFmtBlock) void fmt_write_kernel(const mofreak_row *rows, int64_t n, const uint8_t *len8, const uint64_t *block_off, char *out,
                                                               uint64_t cap)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFmtStage];
    __shared__ int wave_tot[kFmtBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kFmtBlock + threadIdx.x;
    const uint64_t g0 = block_off[blockIdx.x], g1 = block_off[blockIdx.x + 1];
    if (g1 > cap) return;  // (the caller's buffer is too small: nothing is written beyond it; the host reports it)
    const int total = (int)(g1 - g0);
    const int shift = (int)(((uintptr_t)out + g0) & 15u);  // LDS byte shift + k <-> output byte g0 + k: the same position in a 16-byte chunk
    const int len = i < n ? (int)len8[i] : 0;
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = shift;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
    if (i < n) {
        FmtStore st{stage + base + incl - len};
        fmt_row(st, load_row(rows, i));
    }
    __syncthreads();
    // copy out: chunk c = LDS bytes [16 c, 16 c + 16) = output bytes starting at g0 - shift + 16 c
    const int n_chunks = (shift + total + 15) >> 4;
    char *obase = out + g0 - shift;
    for (int c = threadIdx.x; c < n_chunks; c += kFmtBlock) {
        const int lo = 16 * c, hi = lo + 16;
        if (lo >= shift && hi <= shift + total) {
            *reinterpret_cast<uint4 *>(obase + lo) = *reinterpret_cast<const uint4 *>(stage + lo);
        } else {  // the block's first or last chunk: its other bytes belong to the neighbouring blocks
            for (int k = lo > shift ? lo : shift; k < hi && k < shift + total; ++k) obase[k] = (char)stage[k];
        }
    }
}

// byte offset of the text of row row_starts[s] (ascending row indices; n: the text's end)
__global__ void fmt_segment_kernel(const int64_t *row_starts, int n_segments, int64_t n, const uint8_t *len8, const uint64_t *block_off, uint64_t *seg_off)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segments) return;
    const int64_t r = row_starts[s] < n ? (row_starts[s] < 0 ? 0 : row_starts[s]) : n;
    const int64_t b = r / kFmtBlock;
    uint64_t off = block_off[b];
    for (int64_t k = b * kFmtBlock; k < r; ++k) off += len8[k];
    seg_off[s] = off;
}

}  // namespace

// workspace: len8 [n] + block_sum [n_blocks] (u32) + block_off [n_blocks + 1] (u64) + seg_off [n_segments] (u64), 16-byte aligned pieces
size_t format_workspace_bytes(int64_t n_rows, int n_segments)
{
    const int64_t nb = (n_rows + kFmtBlock - 1) / kFmtBlock;
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    return up((size_t)n_rows) + up((size_t)nb * 4) + up((size_t)(nb + 1) * 8) + up((size_t)(n_segments > 0 ? n_segments : 1) * 8) + 64;
}

// pass 1 + 2 (+ segment offsets): after it total_out[0] (device, in the workspace) holds the text's length
int launch_format_measure(const mofreak_row *d_rows, int64_t n_rows, void *ws, const int64_t *d_row_starts, int n_segments, int32_t *d_status_bits,
                          uint64_t **d_total_out, uint64_t **d_seg_off_out, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nb = (n_rows + kFmtBlock - 1) / kFmtBlock;
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    uint8_t *p = static_cast<uint8_t *>(ws);
    uint8_t *len8 = p;
    p += up((size_t)n_rows);
    uint32_t *block_sum = reinterpret_cast<uint32_t *>(p);
    p += up((size_t)nb * 4);
    uint64_t *block_off = reinterpret_cast<uint64_t *>(p);
    p += up((size_t)(nb + 1) * 8);
    uint64_t *seg_off = reinterpret_cast<uint64_t *>(p);
    if (nb > 0) hipLaunchKernelGGL(fmt_len_kernel, dim3((unsigned)nb), dim3(kFmtBlock), 0, s, d_rows, n_rows, len8, block_sum, d_status_bits);
    hipLaunchKernelGGL(fmt_scan_kernel, dim3(1), dim3(1024), 0, s, block_sum, nb, block_off);
    if (n_segments > 0)
        hipLaunchKernelGGL(fmt_segment_kernel, dim3((unsigned)((n_segments + 255) / 256)), dim3(256), 0, s, d_row_starts, n_segments, n_rows, len8, block_off, seg_off);
    *d_total_out = block_off + nb;
    *d_seg_off_out = seg_off;
    return (int)hipGetLastError();
}

// pass 3, after launch_format_measure on the same workspace
int launch_format_write(const mofreak_row *d_rows, int64_t n_rows, void *ws, char *d_text, uint64_t cap, void *stream)
{
    const int64_t nb = (n_rows + kFmtBlock - 1) / kFmtBlock;
    if (nb == 0) return 0;
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    uint8_t *p = static_cast<uint8_t *>(ws);
    const uint8_t *len8 = p;
    p += up((size_t)n_rows) + up((size_t)nb * 4);
    const uint64_t *block_off = reinterpret_cast<const uint64_t *>(p);
    hipLaunchKernelGGL(fmt_write_kernel, dim3((unsigned)nb), dim3(kFmtBlock), 0, static_cast<hipStream_t>(stream), d_rows, n_rows, len8, block_off, d_text, cap);
    return (int)hipGetLastError();
}

}  // namespace mofreak
