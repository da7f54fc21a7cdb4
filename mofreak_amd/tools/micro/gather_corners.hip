// Microbenchmark behind profiles/r05_frame_loop_experiments.txt: box sums from a u32 integral (1080p, 16 pairs, 6969 spatially sorted
// keypoints each, a FREAK-like pattern of 43 boxes of radius R), one wave per keypoint: (A) lane = box, four loads -- the gather
// path's form; (B) four lanes per box, a corner each, three rounds -- neighbouring lanes on the same image row.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mofreak_amd/tools/micro/gather_corners.hip -o gather_corners && ./gather_corners 50
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
struct Box { short dx, dy, w, h; };
__global__ __launch_bounds__(256) void kernA(const uint32_t *integ, int pitch, const int2 *kps, int n, const Box *boxes, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const Box b = boxes[lane < 43 ? lane : 0];
    for (int k = wave; k < n; k += nw) {
        const int2 kp = kps[k];
        uint32_t s = 0;
        if (lane < 43) {
            const uint32_t *p = integ + (kp.y + b.dy) * pitch + kp.x + b.dx;
            s = p[b.h * pitch + b.w] - p[b.h * pitch] - p[b.w] + p[0];
        }
        if (lane < 43) out[(size_t)k * 64 + lane] = s;
    }
}
__global__ __launch_bounds__(256) void kernB(const uint32_t *integ, int pitch, const int2 *kps, int n, const Box *boxes, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    Box b[3];
    for (int r = 0; r < 3; ++r) b[r] = boxes[min(16 * r + (lane >> 2), 42)];
    const int cx = lane & 1, cy = (lane >> 1) & 1;
    for (int k = wave; k < n; k += nw) {
        const int2 kp = kps[k];
        uint32_t v[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) v[r] = integ[(kp.y + b[r].dy + cy * b[r].h) * pitch + kp.x + b[r].dx + cx * b[r].w];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint32_t s = (cx ^ cy) ? 0u - v[r] : v[r];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if ((lane & 3) == 0 && 16 * r + (lane >> 2) < 43) out[(size_t)k * 64 + 16 * r + (lane >> 2)] = s;
        }
    }
}
// (C) as (A) on an integral stored in 4 x 4-entry blocks of 64 bytes (a 2-D blocked layout): bx = x >> 2, by = y >> 2
__device__ __forceinline__ uint32_t blk(const uint32_t *integ, int pitch_b, int x, int y)
{
    return integ[(((size_t)(y >> 2) * pitch_b + (x >> 2)) << 4) + ((y & 3) << 2) + (x & 3)];
}
__global__ __launch_bounds__(256) void kernC(const uint32_t *integ, int pitch_b, const int2 *kps, int n, const Box *boxes, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const Box b = boxes[lane < 43 ? lane : 0];
    for (int k = wave; k < n; k += nw) {
        const int2 kp = kps[k];
        if (lane < 43) {
            const int x = kp.x + b.dx, y = kp.y + b.dy;
            out[(size_t)k * 64 + lane] = blk(integ, pitch_b, x + b.w, y + b.h) - blk(integ, pitch_b, x, y + b.h) - blk(integ, pitch_b, x + b.w, y) + blk(integ, pitch_b, x, y);
        }
    }
}
int main(int argc, char **argv)
{
    const int W = 1920, H = 1080, pitch = W + 1, pairs = 16;
    const int R = argc > 1 ? atoi(argv[1]) : 50;       // pattern radius
    const int kp_per_pair = 6969;
    std::vector<uint32_t> integ((size_t)pairs * (H + 1) * pitch);
    for (size_t i = 0; i < integ.size(); ++i) integ[i] = (uint32_t)(i * 2654435761u);
    std::vector<Box> boxes(43);
    srand(1);
    // FREAK-like: rings of 6 at radii R, .78R, .6R, .45R, .32R, .2R, .1R(6), centre; sigma ~ radius/2 shrinking
    const double rad[8] = {1.0, 0.78, 0.6, 0.45, 0.32, 0.2, 0.1, 0}, sg[8] = {0.33, 0.26, 0.2, 0.15, 0.11, 0.075, 0.05, 0.04};
    for (int i = 0; i < 43; ++i) {
        const int ring = i / 6;
        const double a = (i % 6) * M_PI / 3 + (ring & 1) * M_PI / 6, r = rad[ring] * R, s = std::max(1.0, sg[ring] * R);
        boxes[i] = Box{(short)lrint(r * cos(a) - s), (short)lrint(r * sin(a) - s), (short)(2 * lrint(s) + 1), (short)(2 * lrint(s) + 1)};
    }
    std::vector<int2> kps;
    for (int p = 0; p < pairs; ++p) {  // spatially sorted: bands of 64 rows, x ascending
        std::vector<int2> v;
        for (int i = 0; i < kp_per_pair; ++i) v.push_back(int2{2 * R + rand() % (W - 4 * R), 2 * R + rand() % (H - 4 * R)});
        std::sort(v.begin(), v.end(), [](int2 a, int2 b) { return (a.y / 64) != (b.y / 64) ? a.y / 64 < b.y / 64 : a.x < b.x; });
        for (auto k : v) kps.push_back(int2{k.x, k.y + p * (H + 1)});
    }
    const int n = (int)kps.size();
    uint32_t *d_integ, *d_outA, *d_outB; int2 *d_kps; Box *d_boxes;
    hipMalloc(&d_integ, integ.size() * 4); hipMalloc(&d_outA, (size_t)n * 256); hipMalloc(&d_outB, (size_t)n * 256);
    hipMalloc(&d_kps, n * sizeof(int2)); hipMalloc(&d_boxes, 43 * sizeof(Box));
    hipMemcpy(d_integ, integ.data(), integ.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_kps, kps.data(), n * sizeof(int2), hipMemcpyHostToDevice);
    hipMemcpy(d_boxes, boxes.data(), 43 * sizeof(Box), hipMemcpyHostToDevice);
    hipMemset(d_outA, 0, (size_t)n * 256); hipMemset(d_outB, 0, (size_t)n * 256);
    // the same values in the blocked layout (all pairs as one tall image of pairs * (H + 1) rows)
    const int rows_all = pairs * (H + 1), pitch_b = (pitch + 3) / 4, rows_b = (rows_all + 3) / 4;
    std::vector<uint32_t> integ_b((size_t)rows_b * pitch_b * 16, 0u);
    for (int y = 0; y < rows_all; ++y)
        for (int x = 0; x < pitch; ++x) integ_b[(((size_t)(y >> 2) * pitch_b + (x >> 2)) << 4) + ((y & 3) << 2) + (x & 3)] = integ[(size_t)y * pitch + x];
    uint32_t *d_integ_b, *d_outC;
    hipMalloc(&d_integ_b, integ_b.size() * 4); hipMalloc(&d_outC, (size_t)n * 256);
    hipMemcpy(d_integ_b, integ_b.data(), integ_b.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_outC, 0, (size_t)n * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (which == 0) hipLaunchKernelGGL(kernA, dim3(256 * 5), dim3(256), 0, 0, d_integ, pitch, d_kps, n, d_boxes, d_outA);
                else if (which == 1) hipLaunchKernelGGL(kernB, dim3(256 * 5), dim3(256), 0, 0, d_integ, pitch, d_kps, n, d_boxes, d_outB);
                else hipLaunchKernelGGL(kernC, dim3(256 * 5), dim3(256), 0, 0, d_integ_b, pitch_b, d_kps, n, d_boxes, d_outC);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("R=%d kernel %c: %.3f ms per launch of %d keypoints (%.2f ns per keypoint, %.1f G corner reads/s)\n", R, 'A' + which, ms / 10, n, ms / 10 * 1e6 / n, 172.0 * n / (ms / 10) / 1e6);
        }
    }
    std::vector<uint32_t> a((size_t)n * 64), b((size_t)n * 64);
    hipMemcpy(a.data(), d_outA, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_outB, b.size() * 4, hipMemcpyDeviceToHost);
    std::vector<uint32_t> c((size_t)n * 64);
    hipMemcpy(c.data(), d_outC, c.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]) + (a[i] != c[i]);
    printf("mismatches %zu\n", bad);
    return 0;
}
