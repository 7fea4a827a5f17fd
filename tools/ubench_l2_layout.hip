// What the level-2 kernel's MEMORY PATTERN costs by itself, for the run layouts on the table (round 5):
//   A  today's:   groups of four 5-byte items = 16 + 4 bytes at a 20-byte stride (a run's piece of a tile: 4 groups = 80 contiguous bytes,
//                 dword-aligned only: every piece ends in partial 32-byte sectors)
//   B  blocks:    six 5-byte items in 32 bytes (24 B of low words + 6 high bytes + 2 B unused), 32-byte aligned: a piece = 2 or 3 whole sectors
//   C  blocks64:  twelve items in 64 bytes, 64-byte aligned (a piece = 1 or 2 whole 64-byte lines; what a 12-item carry would buy)
// One workgroup of 1024 threads per CU streams a bucket in (16 + 8-byte loads of 24-byte groups, as k_p2_fast does) and appends 16 items
// per tile to each of 1024 runs -- no LDS sort, no hash: the loads and stores only.  Each also runs with loads only / stores only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_layout.hip -o /tmp/ubench_l2 && /tmp/ubench_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));

constexpr int TILE = 16384, RUNS = 1024;

template <int MODE /* 0: A, 1: B, 2: C, 3: D = A's groups in a chunk-interleaved layout */, bool LOADS, bool STORES, int GPC = 64 /* D: groups per chunk */>
__global__ void __launch_bounds__(1024) k_l2(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t items_per_wg, uint64_t run_cap_bytes, uint32_t* sink) {
    const uint32_t tid = threadIdx.x;
    const uint8_t* bucket = in + (uint64_t)blockIdx.x * (items_per_wg / 4) * 24;
    uint8_t* runs = out + (uint64_t)blockIdx.x * RUNS * run_cap_bytes;
    uint32_t acc = 0;
    uint32_t cur = 0;                                   // bytes written to each run so far (the same for every run: 16 items per tile)
    uint32_t t = 0;
    for (uint64_t tbeg = 0; tbeg + TILE <= items_per_wg; tbeg += TILE, ++t) {
        u32x4 lo[4]; u32x2 hi[4];
        if (LOADS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint8_t* p = bucket + ((tbeg >> 2) + (uint64_t)u * 1024 + tid) * 24;
                lo[u] = *reinterpret_cast<const u32x4_a4*>(p);
                hi[u] = *reinterpret_cast<const u32x2_a4*>(p + 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= lo[u].x ^ lo[u].w ^ hi[u].x ^ hi[u].y;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { lo[u] = u32x4{tid, (uint32_t)tbeg, 3u, (uint32_t)u}; hi[u] = u32x2{tid, 7u}; }
        }
        if (STORES) {
            if (MODE == 0) {
                // 4096 groups: group gi of the tile -> run gi / 4, its group gi % 4 of this tile
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t gi = u * 1024 + tid, b = gi >> 2, q = gi & 3;
                    uint8_t* p = runs + (uint64_t)b * run_cap_bytes + cur + q * 20;
                    *reinterpret_cast<u32x4_a4*>(p) = lo[u];
                    *reinterpret_cast<uint32_t*>(p + 16) = hi[u].x;
                }
            } else if (MODE == 3) {
                // D: run b's groups [GPC c, GPC c + GPC) are chunk c of the run, and chunk c of ALL 1024 runs lie side by side: a tile's 4096
                // stores land inside a window of 1024 x GPC x 20 bytes instead of 1024 places 144 KB apart (pages, DRAM rows)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t gi = u * 1024 + tid, b = gi >> 2, g = t * 4 + (gi & 3);
                    uint8_t* p = runs + ((uint64_t)(g / GPC) * RUNS + b) * (GPC * 20) + (g % GPC) * 20;
                    *reinterpret_cast<u32x4_a4*>(p) = lo[u];
                    *reinterpret_cast<uint32_t*>(p + 16) = hi[u].x;
                }
            } else if (MODE == 1) {
                // 16 items per run and tile = 2.67 blocks of six: 3, 3, 2 blocks over three tiles
                const uint32_t nb = (t % 3 == 2) ? 2 : 3;
                for (uint32_t u = 0; u < nb; ++u) {
                    const uint32_t gi = u * 1024 + tid, b = gi / nb, q = gi - b * nb;
                    uint8_t* p = runs + (uint64_t)b * run_cap_bytes + cur + q * 32;
                    *reinterpret_cast<u32x4*>(p) = lo[u];
                    *reinterpret_cast<u32x4*>(p + 16) = u32x4{lo[u].y, lo[u].z, hi[u].x, hi[u].y};
                }
            } else if (MODE == 4) {
                // Q (round 6): C's blocks, but a QUAD of lanes writes a block in ONE instruction (16 bytes each: a whole 64-byte line per request)
                const uint32_t nb = (t % 3 == 2) ? 2 : 1;
                for (uint32_t u = 0; u < nb; ++u) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const uint32_t qi = (u * 4 + h) * 1024 + tid, gi = qi >> 2, w = qi & 3;
                        const uint32_t b = gi / nb, q = gi - b * nb;
                        uint8_t* p = runs + (uint64_t)b * run_cap_bytes + cur + q * 64;
                        *reinterpret_cast<u32x4*>(p + 16 * w) = lo[(u + h) & 3];
                    }
                }
            } else {
                // 16 items per run and tile = 1.33 blocks of twelve: 1, 1, 2 blocks over three tiles
                const uint32_t nb = (t % 3 == 2) ? 2 : 1;
                for (uint32_t u = 0; u < nb; ++u) {
                    const uint32_t gi = u * 1024 + tid, b = gi / nb, q = gi - b * nb;
                    uint8_t* p = runs + (uint64_t)b * run_cap_bytes + cur + q * 64;
#pragma unroll
                    for (int w = 0; w < 4; ++w) *reinterpret_cast<u32x4*>(p + 16 * w) = lo[(u + w) & 3];
                }
            }
        }
        cur += MODE == 0 || MODE == 3 ? 80u : MODE == 1 ? ((t % 3 == 2) ? 64u : 96u) : ((t % 3 == 2) ? 128u : 64u);      // (modes 2 and 4: blocks of twelve)
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const uint64_t items_per_wg = (uint64_t)1479 * TILE;            // a bucket of config 4's rounds: 24.2 M items
    const uint64_t in_bytes = (uint64_t)cus * (items_per_wg / 4) * 24 + 4096;
    const uint64_t tiles = items_per_wg / TILE;
    const uint64_t run_cap = ((tiles * 96 + 4095) / 4096) * 4096 + 4096;      // bytes per run: room for every layout
    const uint64_t out_bytes = (uint64_t)cus * RUNS * run_cap;
    uint8_t *in, *out; uint32_t* sink;
    if (hipMalloc(&in, in_bytes) != hipSuccess || hipMalloc(&out, out_bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed (%.1f + %.1f GB)\n", in_bytes / 1e9, out_bytes / 1e9); return 1; }
    hipMemset(in, 1, in_bytes); hipMemset(out, 0, out_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double items = (double)cus * items_per_wg;
    printf("%d CUs, %.1f M items per workgroup, in %.1f GB, out %.1f GB (run capacity %llu B)\n", cus, items_per_wg / 1e6, in_bytes / 1e9, out_bytes / 1e9, (unsigned long long)run_cap);
    auto time = [&](const char* name, double in_b, double out_b, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-46s best %7.2f ms (mean %7.2f)  %6.2f G items/s  payload %5.2f TB/s\n", name, best, sum / 3, items / best / 1e6, (in_b + out_b) / best / 1e9);
    };
    const double in_b = items * 6.0;
#define RUN(MODE, L, S, NAME, OUTB) time(NAME, (L) ? in_b : 0.0, (S) ? (OUTB) : 0.0, [&] { hipLaunchKernelGGL((k_l2<MODE, L, S>), dim3(cus), dim3(1024), 0, 0, in, out, items_per_wg, run_cap, sink); })
#define RUND(GPC, L, S, NAME) time(NAME, (L) ? in_b : 0.0, (S) ? items * 5.0 : 0.0, [&] { hipLaunchKernelGGL((k_l2<3, L, S, GPC>), dim3(cus), dim3(1024), 0, 0, in, out, items_per_wg, run_cap, sink); })
    RUN(0, true, false, "loads only (24-byte groups, 16 + 8 B)", 0.0);
    RUN(0, false, true, "A stores only: 20 B groups, 80 B pieces", items * 5.0);
    RUN(1, false, true, "B stores only: 32 B blocks of six", items * 32.0 / 6.0);
    RUN(2, false, true, "C stores only: 64 B blocks of twelve", items * 64.0 / 12.0);
    RUND(64, false, true, "D stores only: A's groups, chunks of 64 groups");
    RUND(16, false, true, "D stores only: chunks of 16 groups");
    RUND(256, false, true, "D stores only: chunks of 256 groups");
    RUN(0, true, true, "A loads + stores", items * 5.0);
    RUND(64, true, true, "D loads + stores: chunks of 64 groups");
    RUND(16, true, true, "D loads + stores: chunks of 16 groups");
    RUND(256, true, true, "D loads + stores: chunks of 256 groups");
    RUN(1, true, true, "B loads + stores", items * 32.0 / 6.0);
    RUN(2, true, true, "C loads + stores", items * 64.0 / 12.0);
    RUN(4, false, true, "Q stores only: 64 B blocks, a quad each", items * 64.0 / 12.0);
    RUN(4, true, true, "Q loads + stores", items * 64.0 / 12.0);
    return 0;
}
