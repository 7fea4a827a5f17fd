// What level 1's STORE PATTERN costs by itself (round 5): every workgroup appends, tile after tile, a piece of ~16 six-byte items to its own
// segment of each of 512 buckets.
//   A  today's:  groups of four items = 16 + 8 bytes at a 24-byte stride, a piece = 4 or 5 groups (padded), dword-aligned only
//   C  blocks:   ten items in 64 bytes (60 + 4 unused), 64-byte aligned, a piece = 1 or 2 whole blocks (what a carry of up to nine items per
//                bucket in LDS would buy; costs the LDS of the third workgroup per CU)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l1_layout.hip -o tools/ubench_l1_layout.bin && tools/ubench_l1_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
constexpr int BUCKETS = 512;

template <int MODE>
__global__ void __launch_bounds__(512) k_l1(uint8_t* __restrict__ out, uint64_t bucket_stride, uint64_t seg_bytes, uint32_t tiles) {
    const uint32_t tid = threadIdx.x;
    uint8_t* seg0 = out + (uint64_t)blockIdx.x * seg_bytes;          // this workgroup's segment of bucket 0; bucket b: + b * bucket_stride
    uint32_t cur = 0;                                                // bytes written to each of my segments so far
    for (uint32_t t = 0; t < tiles; ++t) {
        const u32x4 v = {tid, t, 3u, 4u};
        if (MODE == 0) {
            const uint32_t ng = (t % 5 == 1 || t % 5 == 3) ? 5 : 4;      // 4, 5, 4, 5, 4 groups per bucket: 4.4 on average = 16.3 k-mers + padding
            for (uint32_t gi = tid; gi < ng * BUCKETS; gi += 512) {
                const uint32_t b = gi / ng, q = gi - b * ng;
                uint8_t* p = seg0 + (uint64_t)b * bucket_stride + cur + q * 24;
                *reinterpret_cast<u32x4_a4*>(p) = v;
                *reinterpret_cast<u32x2_a4*>(p + 16) = u32x2{tid, t};
            }
            cur += ng * 24;
        } else {
            const uint32_t nb = (t % 5 == 1 || t % 5 == 4) ? 1 : 2;      // 2, 1, 2, 2, 1 blocks of ten: 16 items on average
            for (uint32_t gi = tid; gi < nb * BUCKETS; gi += 512) {
                const uint32_t b = gi / nb, q = gi - b * nb;
                uint8_t* p = seg0 + (uint64_t)b * bucket_stride + cur + q * 64;
#pragma unroll
                for (int w = 0; w < 4; ++w) *reinterpret_cast<u32x4*>(p + 16 * w) = v;
            }
            cur += nb * 64;
        }
    }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double round_items = 12.4e9;
    uint8_t* out = nullptr;
    const uint64_t cap = (uint64_t)96 << 30;
    if (hipMalloc(&out, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int per_cu : {3, 2}) {
        const uint32_t wgs = (uint32_t)(cus * per_cu);
        const uint32_t tiles = (uint32_t)(round_items / wgs / 8160.0);           // 8 K-base tiles, 8160 window starts each
        const uint64_t seg_bytes = ((uint64_t)tiles * 116 + 4095) / 4096 * 4096;  // room for either layout (<= 5 x 24 or 2 x 64 bytes per tile)
        const uint64_t bucket_stride = seg_bytes * wgs;
        if (bucket_stride * BUCKETS > cap) { printf("buffer too small\n"); return 1; }
        const double items = (double)wgs * tiles * BUCKETS * 16.0;
        auto time = [&](const char* name, double bytes, auto launch) {
            launch(); hipDeviceSynchronize();
            float best = 1e9f;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("%d WG/CU  %-44s best %7.2f ms  %6.1f G items/s  written %5.2f TB/s\n", per_cu, name, best, items / best / 1e6, bytes / best / 1e9);
        };
        time("A: 24 B groups, pieces of 4-5 groups", items / 16 * 4.4 * 24, [&] { hipLaunchKernelGGL(k_l1<0>, dim3(wgs), dim3(512), 0, 0, out, bucket_stride, seg_bytes, tiles); });
        time("C: 64 B blocks of ten, 1-2 per piece", items / 16 * 1.6 * 64, [&] { hipLaunchKernelGGL(k_l1<1>, dim3(wgs), dim3(512), 0, 0, out, bucket_stride, seg_bytes, tiles); });
    }
    return 0;
}
