// Does the cost of level 1's store pattern depend on WHERE the driver put the buffer?  (Round 5: the same bench process measured level 1 at
// 137 or at 176 ms on one box, stable within a process.)  Allocates the buffer several times in one process -- exact size / whole GiB,
// with spacers kept in between -- and times the pattern in today's layout ([bucket][workgroup] segments) and in the transposed one
// ([workgroup][bucket]: a workgroup's 512 cursors inside ~116 MB instead of spread over the whole buffer).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_placement.hip -o tools/probe_placement.bin && tools/probe_placement.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
constexpr int BUCKETS = 512;

__global__ void __launch_bounds__(512) k_l1(uint8_t* __restrict__ out, uint64_t wg_stride, uint64_t bucket_stride, uint32_t tiles) {
    const uint32_t tid = threadIdx.x;
    uint8_t* seg0 = out + (uint64_t)blockIdx.x * wg_stride;
    uint32_t cur = 0;
    for (uint32_t t = 0; t < tiles; ++t) {
        const u32x4 v = {tid, t, 3u, 4u};
        const uint32_t ng = (t % 5 == 1 || t % 5 == 3) ? 5 : 4;
        for (uint32_t gi = tid; gi < ng * BUCKETS; gi += 512) {
            const uint32_t b = gi / ng, q = gi - b * ng;
            uint8_t* p = seg0 + (uint64_t)b * bucket_stride + cur + q * 24;
            *reinterpret_cast<u32x4_a4*>(p) = v;
            *reinterpret_cast<u32x2_a4*>(p + 16) = u32x2{tid, t};
        }
        cur += ng * 24;
    }
}
__global__ void __launch_bounds__(1024) k_fill(u32x4* __restrict__ out, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = u32x4{1u, 2u, 3u, 4u};
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t wgs = (uint32_t)cus * 3;
    const double round_items = 12.4e9;
    const uint32_t tiles = (uint32_t)(round_items / wgs / 8160.0);
    const uint64_t seg_bytes = ((uint64_t)tiles * 106 + 255) / 256 * 256;        // 4.4 groups x 24 bytes per tile and bucket, and a little
    const uint64_t need = seg_bytes * wgs * BUCKETS;
    std::vector<void*> spacers;
    auto time_it = [&](auto launch) { float best = 1e9f; for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
    for (int cycle = 0; cycle < 6; ++cycle) {
        size_t free_b = 0, total_b = 0; hipMemGetInfo(&free_b, &total_b);
        uint64_t bytes = need;
        const char* how = "exact";
        if (cycle % 3 == 1) { bytes = (need + ((uint64_t)1 << 30) - 1) >> 30 << 30; how = "whole GiB"; }
        if (cycle % 3 == 2) { bytes = (uint64_t)(0.85 * (double)free_b); how = "0.85 of free"; }
        uint8_t* out = nullptr;
        hipEvent_t a0, a1; (void)a0; (void)a1;
        if (hipMalloc(&out, bytes) != hipSuccess) { printf("cycle %d: alloc of %.1f GB failed\n", cycle, bytes / 1e9); return 1; }
        const float t_fill = time_it([&] { hipLaunchKernelGGL(k_fill, dim3(cus * 2), dim3(1024), 0, 0, (u32x4*)out, need / 16); });
        const float t_a = time_it([&] { hipLaunchKernelGGL(k_l1, dim3(wgs), dim3(512), 0, 0, out, seg_bytes, seg_bytes * wgs, tiles); });
        const float t_t = time_it([&] { hipLaunchKernelGGL(k_l1, dim3(wgs), dim3(512), 0, 0, out, seg_bytes * BUCKETS, seg_bytes, tiles); });
        printf("cycle %d  %-13s %.1f GB at %p (free before %.1f GB)   fill %.1f GB: %6.2f ms   [bucket][wg] %6.2f ms   [wg][bucket] %6.2f ms\n",
               cycle, how, bytes / 1e9, (void*)out, free_b / 1e9, need / 1e9, t_fill, t_a, t_t);
        fflush(stdout);
        hipFree(out);
        void* sp = nullptr;                                                        // perturb where the next one lands
        if (hipMalloc(&sp, (size_t)(3 + cycle) << 30) == hipSuccess) spacers.push_back(sp);
    }
    for (void* s : spacers) hipFree(s);
    return 0;
}
