// Bag-of-words codeword assignment for 16-byte MoFREAK descriptors on gfx950.
//
// Replaces BagOfWordsRepresentation::bruteForceMatch / hammingDistance (BagOfWordsRepresentation.cpp:22-72) -- per
// descriptor a scan of up to 10 100 codewords, 128 bits each, counted bit by bit on the CPU -- and the count /
// normalise part of buildHistogram (:74-138).
//
// The codebook (n_codewords x 16 B, at most 160 KiB) lives in LDS; every lane owns two descriptors in registers;
// all lanes read the same codeword at a time (a broadcast ds_read_b128), XOR, four v_bcnt_u32 and a strict
// less-than keep the FIRST minimum (the reference's tie-break).  Integer work: no MFMA.
#include "device_helpers.h"

namespace mofreak {
namespace {

constexpr int kBowThreads = 1024;
constexpr int kBowPerThread = 2;

__device__ __forceinline__ int hamming128(const uint4 a, const uint4 b)
{
    return __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
}

__global__ __launch_bounds__(kBowThreads) void bow_assign_kernel(const uint8_t *desc, const uint8_t *valid, int64_t n,
                                                                 const uint8_t *codebook, int n_codewords, int32_t *out_index,
                                                                 unsigned int *counts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint4 *cb = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < n_codewords; i += kBowThreads) cb[i] = reinterpret_cast<const uint4 *>(codebook)[i];
    __syncthreads();

    const int64_t stride = (int64_t)gridDim.x * kBowThreads * kBowPerThread;
    for (int64_t base = ((int64_t)blockIdx.x * kBowThreads + threadIdx.x) * kBowPerThread; base < n; base += stride) {
        uint4 d[kBowPerThread];
        int best[kBowPerThread], idx[kBowPerThread];
#pragma unroll
        for (int u = 0; u < kBowPerThread; ++u) {
            const int64_t k = base + u < n ? base + u : n - 1;
            d[u] = reinterpret_cast<const uint4 *>(desc)[k];
            best[u] = 0x7fffffff;
            idx[u] = -1;
        }
#pragma unroll 4
        for (int c = 0; c < n_codewords; ++c) {
            const uint4 w = cb[c];
#pragma unroll
            for (int u = 0; u < kBowPerThread; ++u) {
                const int dist = hamming128(d[u], w);
                if (dist < best[u]) {  // strict: the first minimum wins (BagOfWordsRepresentation.cpp:30)
                    best[u] = dist;
                    idx[u] = c;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kBowPerThread; ++u) {
            const int64_t k = base + u;
            if (k < n) {
                const bool ok = valid == nullptr || valid[k] != 0;
                if (out_index) out_index[k] = ok ? idx[u] : -1;
                if (counts && ok && idx[u] >= 0) atomicAdd(&counts[idx[u]], 1u);
            }
        }
    }
}

// histogram[c] = count[c] / sum(count) in float, as buildHistogram's tail (:125-136); single workgroup.
__global__ __launch_bounds__(1024) void bow_normalize_kernel(const unsigned int *counts, int n_codewords, float *hist, int32_t *success)
{
    __shared__ unsigned long long wave_sum_s[16];
    unsigned long long local = 0;
    for (int c = threadIdx.x; c < n_codewords; c += 1024) local += counts[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) wave_sum_s[threadIdx.x >> 6] = local;
    __syncthreads();
    unsigned long long total = 0;
    for (int i = 0; i < 16; ++i) total += wave_sum_s[i];
    // the reference accumulates in float: exact while the counts stay below 2^24
    const float histogram_sum = (float)total;
    for (int c = threadIdx.x; c < n_codewords; c += 1024) hist[c] = total ? (float)counts[c] / histogram_sum : 0.0f;
    if (threadIdx.x == 0 && success) *success = total ? 1 : 0;
}

}  // namespace

int launch_bow_assign(const uint8_t *desc, const uint8_t *valid, int64_t n, const uint8_t *codebook, int n_codewords,
                      int32_t *out_index, unsigned int *counts, int n_cus, void *stream)
{
    const size_t lds = (size_t)n_codewords * 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bow_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;  // workgroups that fit a CU's LDS (1024 threads each)
    const int64_t want = (n + kBowThreads * kBowPerThread - 1) / (kBowThreads * kBowPerThread);
    const int blocks = (int)(want < (int64_t)n_cus * per_cu ? want : (int64_t)n_cus * per_cu);
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(bow_assign_kernel, dim3(blocks), dim3(kBowThreads), lds, static_cast<hipStream_t>(stream), desc, valid, n,
                       codebook, n_codewords, out_index, counts);
    return (int)hipGetLastError();
}

int launch_bow_normalize(const unsigned int *counts, int n_codewords, float *hist, int32_t *success, void *stream)
{
    hipLaunchKernelGGL(bow_normalize_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), counts, n_codewords, hist, success);
    return (int)hipGetLastError();
}

}  // namespace mofreak
