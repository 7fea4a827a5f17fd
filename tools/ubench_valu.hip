// tools/ubench_valu.hip -- what the integer instructions the stage kernels are made of cost on gfx950, measured: a dependent chain of
// each per lane, enough waves to fill every SIMD (8 per SIMD), cycles per wave-instruction from the wall clock and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Prints lane-operations per second for the chip and the rate relative to v_add_u32, which is what tools/isa_mix.py's weights mean.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ITER = 4096, UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(512) k(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = seed | 1u;
    uint64_t w = ((uint64_t)a << 32) | b;
    __shared__ uint32_t lds[2048];
    if (OP >= 8) { for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i; __syncthreads(); }
    for (int i = 0; i < ITER / UNROLL; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) a = a + b;                                                      // v_add_u32
            else if (OP == 1) a = a * c;                                                 // v_mul_lo_u32
            else if (OP == 2) a = __umulhi(a, c);                                        // v_mul_hi_u32
            else if (OP == 3) a = __umul24(a & 0xFFFFFF, c & 0xFFFF) + 1;                // v_mul_u32_u24 (+ and, add)
            else if (OP == 4) w = (w << 3) ^ w;                                          // v_lshlrev_b64 + 2 xor
            else if (OP == 5) a = __builtin_amdgcn_alignbit(a, b, 7);                    // v_alignbit_b32
            else if (OP == 6) w = w * 0xff51afd7ed558ccdULL;                             // 64-bit multiply (mul_lo x3 + mul_hi / mad_u64)
            else if (OP == 7) a = (w > ((uint64_t)b << 32 | a)) ? a + 1 : a ^ b;          // v_cmp_u64 + cndmask
            else if (OP == 8) a = lds[a & 2047];                                         // dependent ds_read_b32
            else if (OP == 9) a = __shfl_up(a, 1, 64);                                   // ds_bpermute
            else if (OP == 10) a = atomicAdd(&lds[(a * 2654435761u) >> 21], 1u) + a;     // returning LDS atomic, random bank
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ (uint32_t)w;
}

template <int OP>
double run(const char* name, uint32_t* d, double base) {
    const int blocks = 256 * 4, threads = 512;       // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * threads * ITER / (ms * 1e-3);
    printf("%-34s %8.3f ms  %8.2f T lane-ops/s  %5.2f x v_add_u32\n", name, ms, ops / 1e12, base > 0 ? base / ops : 1.0);
    return ops;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 4 * 512 * 4);
    const double base = run<0>("v_add_u32", d, 0);
    run<1>("v_mul_lo_u32", d, base);
    run<2>("v_mul_hi_u32", d, base);
    run<3>("v_mul_u32_u24 + and + add", d, base);
    run<4>("v_lshlrev_b64 + xor x2", d, base);
    run<5>("v_alignbit_b32", d, base);
    run<6>("64-bit multiply", d, base);
    run<7>("v_cmp_u64 + cndmask + add/xor", d, base);
    run<8>("dependent ds_read_b32", d, base);
    run<9>("ds_bpermute (shfl_up)", d, base);
    run<10>("returning LDS atomic add", d, base);
    return 0;
}
