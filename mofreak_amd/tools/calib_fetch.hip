// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes this repository's kernels use
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel reads a
// known number of DISTINCT bytes once from a buffer far larger than the caches.  Build and run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch mofreak_amd/tools/calib_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./calib_fetch        (then again with WRITE_SIZE)
// and divide FETCH_SIZE * 1024 of each kernel by the byte count it prints.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// 16 bytes per lane, consecutive lanes consecutive: the tile kernel's staging loads, the band kernels
__global__ void stream16(const uint4 *src, uint32_t *sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 4 bytes per lane, coalesced: the detector's score / pyramid kernels
__global__ void stream4(const uint32_t *src, uint32_t *sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// one dword per lane, every lane in a 64-byte line of its own, lines in a scrambled order, each line touched once: the
// gather path's box corners in the 32-bit integral
__global__ void scatter4(const uint32_t *src, uint32_t *sink, size_t n_lines)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * 2654435761ull + 12345) % n_lines;  // n_lines is a power of two: a permutation (odd multiplier)
        acc ^= src[line * 16 + (i & 15)];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 16 bytes per lane streaming store: descriptors out
__global__ void store16(uint4 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// 1 byte per lane store, coalesced: the detector's score / bookkeeping planes
__global__ void store1(uint8_t *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)i;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;  // 2 GiB: eight times the Infinity Cache
    void *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4 *)buf, (uint32_t *)sink, bytes / 16);
        hipLaunchKernelGGL(stream4, grid, block, 0, 0, (const uint32_t *)buf, (uint32_t *)sink, bytes / 4);
        hipLaunchKernelGGL(scatter4, grid, block, 0, 0, (const uint32_t *)buf, (uint32_t *)sink, bytes / 64);
        hipLaunchKernelGGL(store16, grid, block, 0, 0, (uint4 *)buf, bytes / 16);
        hipLaunchKernelGGL(store1, grid, block, 0, 0, (uint8_t *)buf, bytes / 4);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("expected bytes per launch: stream16 %zu stream4 %zu scatter4 %zu (64-byte lines touched once) store16 %zu store1 %zu\n", bytes, bytes,
                bytes, bytes, bytes / 4);
    return 0;
}
