// Bag-of-words codeword assignment for 16-byte MoFREAK descriptors on gfx950.
//
// Replaces BagOfWordsRepresentation::bruteForceMatch / hammingDistance (BagOfWordsRepresentation.cpp:22-72) -- per
// descriptor a scan of up to 10 100 codewords, 128 bits each, counted bit by bit on the CPU -- and the count /
// normalise part of buildHistogram (:74-138).
//
// This is the one GEMM-shaped piece of the path, and it runs on the matrix pipe: with every bit as +-64 in an int8,
// the dot product of a descriptor and a codeword is 4096 * (128 - 2 * Hamming distance), so the nearest codeword is
// the LARGEST dot product.  v_mfma_i32_32x32x32_i8 multiplies a 32-descriptor x 32-bit tile with a 32-bit x
// 32-codeword tile; four of them cover the 128 bits.  A wave keeps 128 descriptors (four M tiles, expanded once, in
// registers) and walks the whole codebook; a workgroup of 8 waves shares the codeword tiles through LDS, staged 256
// codewords at a time from a pre-expanded copy of the codebook (bow_expand_kernel: bits -> bytes, already in the
// matrix instruction's B-operand order).  The running best per (lane, accumulator) is one v_max on a packed key
//     key = dot + 2^19 + (4095 - codeword tile)     (dot is a multiple of 4096: the low 12 bits carry the tile)
// so that among equal distances the smallest tile wins; the lane is the codeword's column, and the final reduction
// across the 32 columns breaks ties towards the smallest column: together the FIRST minimum of the reference's strict
// `<` scan (:30).  The k <-> lane-byte assignment inside an MFMA operand does not matter as long as descriptors and
// codewords are expanded by the same function (expand16): a dot product is invariant under a permutation of k.
#include "device_helpers.h"

namespace mofreak {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBowThreads = 512;                 // 8 waves, two per SIMD (the kernel needs ~200 VGPRs)
constexpr int kBowWaves = kBowThreads / 64;
#ifndef BOW_MTILES
#define BOW_MTILES 3
#endif
constexpr int kBowMTiles = BOW_MTILES;           // 32-descriptor tiles per wave
constexpr int kBowDescPerWave = 32 * kBowMTiles;
constexpr int kBowDescPerBlock = kBowDescPerWave * kBowWaves;   // 1024 descriptors per workgroup pass
constexpr int kBowTileBytes = 32 * 128;          // one codeword tile, expanded: [4 k-blocks][2 lane groups][32 columns][16 bytes]
#ifndef BOW_SCHED_BARRIER
#define BOW_SCHED_BARRIER 1
#endif
#ifndef BOW_STAGE_TILES
#define BOW_STAGE_TILES 8
#endif
constexpr int kBowStageTiles = BOW_STAGE_TILES;  // codeword tiles per LDS stage (32 codewords each)
constexpr int kBowStageBytes = kBowStageTiles * kBowTileBytes;
constexpr int kBowKeyBias = 1 << 19;             // dot >= -128 * 4096
static_assert(kBowStageBytes % (kBowThreads * 16) == 0 && kBowStageTiles % 2 == 0, "whole 16-byte pieces per thread and stage; tiles in pairs");
constexpr int kBowCopyIters = kBowStageBytes / (kBowThreads * 16);

// 16 bits -> 16 bytes, bit set: +64, clear: -64.  (x * 0x00204081) & 0x01010101 spreads the four bits of a nibble
// over four bytes; shifted to the sign position and flipped with 0xC0 that is 0x40 / 0xC0.
__device__ __forceinline__ v4i expand16(uint32_t chunk)
{
    v4i r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t nib = (chunk >> (4 * i)) & 15u;
        r[i] = (int)((((nib * 0x00204081u) & 0x01010101u) << 7) ^ 0xC0C0C0C0u);
    }
    return r;
}

// The codebook expanded for the matrix instruction: tile t, k-block kb, lane group g, column c -> 16 bytes = bits
// [32 kb + 16 g, +16) of codeword 32 t + c.  Tiles past the codebook repeat its last codeword (equal distance, larger
// index: never the first minimum).
__global__ __launch_bounds__(256) void bow_expand_kernel(const uint8_t *codebook, int n_codewords, int n_tiles, v4i *expanded)
{
    const int i = blockIdx.x * 256 + threadIdx.x;  // (tile, kb, g, col)
    if (i >= n_tiles * 256) return;
    const int col = i & 31, g = (i >> 5) & 1, kb = (i >> 6) & 3, tile = i >> 8;
    const int cw = min(tile * 32 + col, n_codewords - 1);
    const uint32_t w = reinterpret_cast<const uint32_t *>(codebook)[cw * 4 + kb];
    expanded[i] = expand16((w >> (16 * g)) & 0xffffu);
}

__global__ __launch_bounds__(kBowThreads) void bow_assign_kernel(const uint8_t *desc, const uint8_t *valid, int64_t n,
                                                                 const v4i *expanded, int n_tiles, int32_t *out_index,
                                                                 unsigned int *counts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // two stages of codeword tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave index: a scalar
    const int col = lane & 31, g = lane >> 5;
    const int n_stages = n_tiles / kBowStageTiles;  // n_tiles is a multiple of kBowStageTiles (launcher)
    const uint4 *src = reinterpret_cast<const uint4 *>(expanded);

    for (int64_t base = (int64_t)blockIdx.x * kBowDescPerBlock; base < n; base += (int64_t)gridDim.x * kBowDescPerBlock) {
        // this wave's 128 descriptors, expanded into A operands: lane = (row, lane group)
        v4i A[kBowMTiles][4];
#pragma unroll
        for (int m = 0; m < kBowMTiles; ++m) {
            const int64_t d = min(base + wave * kBowDescPerWave + 32 * m + col, n - 1);
            const uint4 w = reinterpret_cast<const uint4 *>(desc)[d];
            const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) A[m][kb] = expand16((wd[kb] >> (16 * g)) & 0xffffu);
        }
        int best[kBowMTiles][16];
#pragma unroll
        for (int m = 0; m < kBowMTiles; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) best[m][i] = 0;
        v16i kinit_a, kinit_b;  // kBowKeyBias + 4095 - tile for the even / odd tile of a pair, stepped down before each pair
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            kinit_a[i] = kBowKeyBias + 4095 + 2;
            kinit_b[i] = kBowKeyBias + 4094 + 2;
        }

        // first stage of codeword tiles -> LDS
        __syncthreads();  // (the previous pass has finished with both buffers)
#pragma unroll
        for (int u = 0; u < kBowCopyIters; ++u) reinterpret_cast<uint4 *>(lds)[tid + u * kBowThreads] = src[tid + u * kBowThreads];
        __syncthreads();
        for (int s = 0; s < n_stages; ++s) {
            const uint8_t *buf = lds + (s & 1) * kBowStageBytes;
            // The next stage goes into the other buffer (which nobody has read since the barrier before this stage) by
            // LDS-DMA: global_load_lds_dwordx4 writes a wave's 64 x 16 bytes straight into LDS, no registers involved, and
            // runs under this stage's matrix work; the wave waits for its own pieces before the stage's closing barrier.
            if (s + 1 < n_stages) {
#pragma unroll
                for (int u = 0; u < kBowCopyIters; ++u) {
                    const int piece = u * kBowWaves + wave;  // 1 KiB pieces of the stage
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + (int64_t)(s + 1) * (kBowStageBytes / 16) + piece * 64 + lane),
                        (__attribute__((address_space(3))) void *)(lds + ((s + 1) & 1) * kBowStageBytes + piece * 1024), 16, 0, 0);
                }
            }
            const uint8_t *bsrc = buf + (g * 32 + col) * 16;
            // Two codeword tiles per step: the key's constant part goes in as the accumulators' initial value (one register
            // block per tile of the pair, shared by the M tiles, stepped once per pair), so that the epilogue is ONE
            // v_max3 per pair of accumulators.
#pragma unroll 1
            for (int tt = 0; tt < kBowStageTiles; tt += 2) {
                v4i Ba[4], Bb[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    Ba[kb] = *reinterpret_cast<const v4i *>(bsrc + tt * kBowTileBytes + kb * 1024);
                    Bb[kb] = *reinterpret_cast<const v4i *>(bsrc + (tt + 1) * kBowTileBytes + kb * 1024);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    kinit_a[i] -= 2;
                    kinit_b[i] -= 2;
                }
#pragma unroll
                for (int m = 0; m < kBowMTiles; ++m) {
#if BOW_SCHED_BARRIER
                    __builtin_amdgcn_sched_barrier(0);  // one M tile's chain pair and epilogue at a time: the registers do not hold more
#endif
                    v16i acc_a = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[m][0], Ba[0], kinit_a, 0, 0, 0);
                    v16i acc_b = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[m][0], Bb[0], kinit_b, 0, 0, 0);
#pragma unroll
                    for (int kb = 1; kb < 4; ++kb) {
                        acc_a = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[m][kb], Ba[kb], acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[m][kb], Bb[kb], acc_b, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) best[m][i] = max(max(best[m][i], acc_a[i]), acc_b[i]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed
            __syncthreads();
        }
        // accumulator i of lane (col, g) is row (i % 4) + 8 * (i / 4) + 4 * g of the 32 x 32 tile: for every row, the
        // best over the 32 columns; ties (equal distance) go to the smallest codeword index
#pragma unroll
        for (int m = 0; m < kBowMTiles; ++m) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = best[m][i];
                const int j = 32 * (4095 - (key & 4095)) + col;
                int k2 = ((key >> 12) << 14) | (16383 - j);
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) k2 = max(k2, __shfl_xor(k2, o));
                if (col == 0) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * g;
                    const int64_t d = base + wave * kBowDescPerWave + 32 * m + row;
                    if (d < n) {
                        const int idx = 16383 - (k2 & 16383);
                        const bool ok = valid == nullptr || valid[d] != 0;
                        if (out_index) out_index[d] = ok ? idx : -1;
                        if (counts && ok) atomicAdd(&counts[idx], 1u);
                    }
                }
            }
        }
    }
}

// histogram[c] = count[c] / sum(count) as buildHistogram's tail (:115, :125-136) computes it IN FLOAT: a bin is a float
// that was incremented by 1 per descriptor, so it stops growing at 2^24, and the sum adds the bins in index order with
// a rounding per step.  While the total stays below 2^24 all of that is exact integer arithmetic and the order does not
// matter; beyond it one lane replays the reference's loop.  Single workgroup.
__global__ __launch_bounds__(1024) void bow_normalize_kernel(const unsigned int *counts, int n_codewords, float *hist, int32_t *success)
{
    __shared__ unsigned long long wave_sum_s[16];
    __shared__ float sum_s;
    unsigned long long local = 0;
    for (int c = threadIdx.x; c < n_codewords; c += 1024) local += counts[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) wave_sum_s[threadIdx.x >> 6] = local;
    __syncthreads();
    unsigned long long total = 0;
    for (int i = 0; i < 16; ++i) total += wave_sum_s[i];
    if (threadIdx.x == 0) {
        float histogram_sum = (float)total;
        if (total > (1ull << 24)) {
            histogram_sum = 0;
            for (int c = 0; c < n_codewords; ++c) histogram_sum += (float)min(counts[c], 1u << 24);
        }
        sum_s = histogram_sum;
    }
    __syncthreads();
    const float histogram_sum = sum_s;
    for (int c = threadIdx.x; c < n_codewords; c += 1024) hist[c] = total ? (float)min(counts[c], 1u << 24) / histogram_sum : 0.0f;
    if (threadIdx.x == 0 && success) *success = total ? 1 : 0;
}

}  // namespace

size_t bow_expanded_bytes(int n_codewords)
{
    const int tiles = (n_codewords + 31) / 32;
    return (size_t)((tiles + kBowStageTiles - 1) / kBowStageTiles * kBowStageTiles) * kBowTileBytes;
}

int launch_bow_assign(const uint8_t *desc, const uint8_t *valid, int64_t n, const uint8_t *codebook, int n_codewords,
                      int32_t *out_index, unsigned int *counts, int n_cus, void *expanded_ws, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_tiles = (int)(bow_expanded_bytes(n_codewords) / kBowTileBytes);
    hipLaunchKernelGGL(bow_expand_kernel, dim3(n_tiles), dim3(256), 0, s, codebook, n_codewords, n_tiles, static_cast<v4i *>(expanded_ws));
    const size_t lds = 2 * (size_t)kBowStageBytes;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bow_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int64_t want = (n + kBowDescPerBlock - 1) / kBowDescPerBlock;
    const int blocks = (int)(want < (int64_t)n_cus ? want : (int64_t)n_cus);
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(bow_assign_kernel, dim3(blocks), dim3(kBowThreads), lds, s, desc, valid, n, static_cast<const v4i *>(expanded_ws), n_tiles,
                       out_index, counts);
    return (int)hipGetLastError();
}

int launch_bow_normalize(const unsigned int *counts, int n_codewords, float *hist, int32_t *success, void *stream)
{
    hipLaunchKernelGGL(bow_normalize_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), counts, n_codewords, hist, success);
    return (int)hipGetLastError();
}

}  // namespace mofreak
