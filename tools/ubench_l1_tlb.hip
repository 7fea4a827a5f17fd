// Is level 1's store pattern bound by ADDRESS TRANSLATION (round 6)?  Every workgroup appends, tile after tile, a piece of ~16 six-byte
// items to its own segment of each of 512 buckets (tools/ubench_l1_layout.hip: that pattern alone costs what the real kernel does).  With the
// level-1 buffer laid out [bucket][workgroup] a workgroup's 512 open streams lie one bucket stride (~160 MB) apart: 512 translations that
// no per-CU TLB holds.  The same stores with the buffer laid out [workgroup][bucket] (a workgroup's segments side by side: ~100 MB, some
// fifty 2 MB fragments) or in WINDOWS (the pieces of W consecutive tiles of all 512 buckets side by side: one or two fragments at a time).
//   layout 0  [bucket][wg][segment]            today's
//   layout 1  [wg][bucket][segment]
//   layout 2  [wg][window][bucket][W tiles]    W = 16 tiles
//   pattern A  groups of four items = 16 + 8 bytes at a 24-byte stride, a piece = 4 or 5 groups, dword-aligned only (today's)
//   pattern C  64-byte blocks of ten, one lane per block (four 16-byte stores);  pattern Q: the same blocks, four lanes per block
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l1_tlb.hip -o tools/ubench_l1_tlb.bin && tools/ubench_l1_tlb.bin [only-this-case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
constexpr int BUCKETS = 512;
constexpr uint32_t WIN = 16;            // tiles per window (layout 2)
constexpr uint32_t WIN_PIECE = 2048;    // bytes a bucket has in a window: 16 tiles x <= 120 bytes, rounded up

struct Geo { uint64_t bucket_stride, wg_stride, win_stride; int layout; };

__device__ __forceinline__ uint8_t* piece_base(uint8_t* out, const Geo& g, uint32_t b, uint32_t t, uint32_t cur, uint32_t cur_win) {
    if (g.layout == 2) return out + (uint64_t)blockIdx.x * g.wg_stride + (uint64_t)(t / WIN) * g.win_stride + (uint64_t)b * WIN_PIECE + cur_win;
    return out + (uint64_t)blockIdx.x * g.wg_stride + (uint64_t)b * g.bucket_stride + cur;
}

template <int PAT>
__global__ void __launch_bounds__(512) k_l1(uint8_t* __restrict__ out, Geo g, uint32_t tiles) {
    const uint32_t tid = threadIdx.x;
    uint32_t cur = 0, cur_win = 0;
    for (uint32_t t = 0; t < tiles; ++t) {
        if (t % WIN == 0) cur_win = 0;
        const u32x4 v = {tid, t, 3u, 4u};
        if (PAT == 0) {
            const uint32_t ng = (t % 5 == 1 || t % 5 == 3) ? 5 : 4;
            for (uint32_t gi = tid; gi < ng * BUCKETS; gi += 512) {
                const uint32_t b = gi / ng, q = gi - b * ng;
                uint8_t* p = piece_base(out, g, b, t, cur, cur_win) + q * 24;
                *reinterpret_cast<u32x4_a4*>(p) = v;
                *reinterpret_cast<u32x2_a4*>(p + 16) = u32x2{tid, t};
            }
            cur += ng * 24; cur_win += ng * 24;
        } else if (PAT == 1) {
            const uint32_t nb = (t % 5 == 1 || t % 5 == 4) ? 1 : 2;
            for (uint32_t gi = tid; gi < nb * BUCKETS; gi += 512) {
                const uint32_t b = gi / nb, q = gi - b * nb;
                uint8_t* p = piece_base(out, g, b, t, cur, cur_win) + q * 64;
#pragma unroll
                for (int w = 0; w < 4; ++w) *reinterpret_cast<u32x4*>(p + 16 * w) = v;
            }
            cur += nb * 64; cur_win += nb * 64;
        } else {
            const uint32_t nb = (t % 5 == 1 || t % 5 == 4) ? 1 : 2;
            for (uint32_t qi = tid; qi < 4 * nb * BUCKETS; qi += 512) {
                const uint32_t gi = qi >> 2, w = qi & 3;
                const uint32_t b = gi / nb, q = gi - b * nb;
                uint8_t* p = piece_base(out, g, b, t, cur, cur_win) + q * 64;
                *reinterpret_cast<u32x4*>(p + 16 * w) = v;
            }
            cur += nb * 64; cur_win += nb * 64;
        }
    }
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double round_items = 12.4e9;
    uint8_t* out = nullptr;
    const uint64_t cap = (uint64_t)100 << 30;
    if (hipMalloc(&out, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
    int case_no = 0;
    for (int per_cu : {3, 2}) {
        const uint32_t wgs = (uint32_t)(cus * per_cu);
        const uint32_t tiles = (uint32_t)(round_items / wgs / 8160.0) / WIN * WIN;
        const uint64_t seg_bytes = ((uint64_t)tiles * 116 + 4095) / 4096 * 4096;
        const double items = (double)wgs * tiles * BUCKETS * 16.0;
        for (int layout = 0; layout < 3; ++layout) {
            Geo g{};
            g.layout = layout;
            if (layout == 0) { g.bucket_stride = seg_bytes * wgs; g.wg_stride = seg_bytes; }
            else if (layout == 1) { g.bucket_stride = seg_bytes; g.wg_stride = seg_bytes * BUCKETS; }
            else { g.win_stride = (uint64_t)WIN_PIECE * BUCKETS; g.wg_stride = g.win_stride * (tiles / WIN); }
            const uint64_t need = layout == 2 ? g.wg_stride * wgs : seg_bytes * wgs * BUCKETS;
            if (need > cap) { printf("buffer too small for layout %d (%.1f GB)\n", layout, need / 1e9); continue; }
            for (int pat = 0; pat < 3; ++pat, ++case_no) {
                if (only >= 0 && only != case_no) continue;
                const double bytes = pat == 0 ? items / 16 * 4.4 * 24 : items / 16 * 1.6 * 64;
                auto launch = [&] {
                    if (pat == 0) hipLaunchKernelGGL(k_l1<0>, dim3(wgs), dim3(512), 0, 0, out, g, tiles);
                    else if (pat == 1) hipLaunchKernelGGL(k_l1<1>, dim3(wgs), dim3(512), 0, 0, out, g, tiles);
                    else hipLaunchKernelGGL(k_l1<2>, dim3(wgs), dim3(512), 0, 0, out, g, tiles);
                };
                launch(); hipDeviceSynchronize();
                float best = 1e9f;
                for (int r = 0; r < 3; ++r) {
                    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                static const char* lname[] = {"[bucket][wg]", "[wg][bucket]", "[wg][window][bucket]"};
                static const char* pname[] = {"A 24 B groups", "C 64 B blocks, a lane each", "Q 64 B blocks, a quad each"};
                printf("case %2d  %d WG/CU  %-22s %-28s best %7.2f ms  %6.1f G items/s  written %5.2f TB/s\n", case_no, per_cu, lname[layout], pname[pat], best,
                       items / best / 1e6, bytes / best / 1e9);
                fflush(stdout);
            }
        }
    }
    return 0;
}
