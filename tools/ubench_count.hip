// ubench_count.hip -- where does k_count's time go?  Variants of the insert step over the same extraction front end.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I kat_amd/csrc tools/ubench_count.hip -o /tmp/ubench_count
#include "kg_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace kg;

enum Mode { FULL = 0, EXTRACT_ONLY = 1, LOAD_ONLY = 2, ADD_NORET = 3, ADD_WG_SCOPE = 4, PLAIN_RMW = 5, ADD_ONLY_RET = 6, LOAD_NT = 7, LOAD_ADD_NORET = 8, PIPELINED = 9 };

template <int MODE>
__global__ void __launch_bounds__(COUNT_BLOCK)
u_count(DevTable t, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_chunks, unsigned long long* sink) {
    __shared__ uint32_t s_code[COUNT_BLOCK + 2];
    __shared__ uint32_t s_bad[COUNT_BLOCK + 2];
    const uint32_t tid = threadIdx.x, k = t.k;
    uint32_t new_distinct = 0;
    uint64_t acc = 0;
    if (tid < 2) { s_code[COUNT_BLOCK + tid] = 0; s_bad[COUNT_BLOCK + tid] = 0xFFFF; }
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t off = chunk * CHUNK_STARTS + (uint64_t)tid * BASES_PER_LANE;
        uint32_t w[4] = {'N' * 0x01010101u, 'N' * 0x01010101u, 'N' * 0x01010101u, 'N' * 0x01010101u};
        if (off + BASES_PER_LANE <= n) { const uint4 v = *reinterpret_cast<const uint4*>(bases + off); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        uint32_t code, bad;
        encode16(w, code, bad);
        s_code[tid] = code; s_bad[tid] = bad;
        __syncthreads();
        if (tid < LANES_WITH_STARTS) {
            uint64_t hi = ((uint64_t)s_code[tid] << 32) | s_code[tid + 1];
            uint64_t lo = (uint64_t)s_code[tid + 2] << 32;
            uint64_t m = ((uint64_t)s_bad[tid] << 48) | ((uint64_t)s_bad[tid + 1] << 32) | ((uint64_t)s_bad[tid + 2] << 16);
            const uint32_t kshift = 64 - 2 * k, mshift = 64 - k;
            if (MODE == PIPELINED) {
                uint64_t keyv[BASES_PER_LANE], posv[BASES_PER_LANE], cur[BASES_PER_LANE];
                bool okv[BASES_PER_LANE];
#pragma unroll
                for (int j = 0; j < BASES_PER_LANE; ++j) {
                    okv[j] = (m >> mshift) == 0;
                    uint64_t fwd = hi >> kshift, rc = kmer_revcomp(fwd, k);
                    keyv[j] = rc < fwd ? rc : fwd;
                    posv[j] = probe_start(keyv[j], t).pos();
                    hi = (hi << 2) | (lo >> 62); lo <<= 2; m <<= 1;
                }
#pragma unroll
                for (int j = 0; j < BASES_PER_LANE; ++j) cur[j] = okv[j] ? t.keys[posv[j]] : 0;
#pragma unroll
                for (int j = 0; j < BASES_PER_LANE; ++j) {
                    if (!okv[j]) continue;
                    if (cur[j] == keyv[j]) atomicAdd(&t.counts[posv[j]], 1u);
                    else table_add(t, keyv[j], 1, new_distinct);
                }
            } else
#pragma unroll 4
            for (int j = 0; j < BASES_PER_LANE; ++j) {
                if ((m >> mshift) == 0) {
                    uint64_t fwd = hi >> kshift;
                    uint64_t rc = kmer_revcomp(fwd, k);
                    uint64_t key = rc < fwd ? rc : fwd;
                    if (MODE == FULL) table_add(t, key, 1, new_distinct);
                    else {
                        uint64_t pos = probe_start(key, t).pos();
                        if (MODE == EXTRACT_ONLY) acc ^= pos;
                        if (MODE == LOAD_ONLY) acc ^= t.keys[pos];
                        if (MODE == LOAD_NT) acc ^= __builtin_nontemporal_load(&t.keys[pos]);
                        if (MODE == ADD_NORET) atomicAdd(&t.counts[pos], 1u);
                        if (MODE == ADD_ONLY_RET) acc ^= atomicAdd(&t.counts[pos], 1u);
                        if (MODE == ADD_WG_SCOPE) __hip_atomic_fetch_add(&t.counts[pos], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (MODE == PLAIN_RMW) t.counts[pos] = t.counts[pos] + 1;
                        if (MODE == LOAD_ADD_NORET) { if (t.keys[pos] == key) atomicAdd(&t.counts[pos], 1u); else acc ^= pos; }
                    }
                }
                hi = (hi << 2) | (lo >> 62); lo <<= 2; m <<= 1;
            }
        }
        __syncthreads();
    }
    if (MODE == FULL) flush_distinct(t, new_distinct);
    if (acc == 0x1234567) atomicAdd(sink, 1ULL);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
float run(DevTable t, const uint8_t* bases, uint64_t n, int grid, unsigned long long* sink, bool reset) {
    if (reset) { CK(hipMemset(t.keys, 0xFF, t.cap * 8)); CK(hipMemset(t.counts, 0, t.cap * 4)); CK(hipMemset(t.ctrs, 0, CTR_WORDS * 8)); }
    const uint64_t n_chunks = (n + CHUNK_STARTS - 1) / CHUNK_STARTS;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(u_count<MODE>, dim3(grid), dim3(COUNT_BLOCK), 0, 0, t, bases, n, n_chunks, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000000ULL;        // bases
    const uint64_t genome = argc > 2 ? strtoull(argv[2], 0, 10) : 100000000ULL;    // distinct-ish
    const int k = 27;
    uint8_t *g, *bases;
    CK(hipMalloc(&g, genome)); CK(hipMalloc(&bases, n));
    hipLaunchKernelGGL(k_synth_genome, dim3(2048), dim3(256), 0, 0, g, genome, 1ULL, 0ULL);
    const uint64_t n_reads = n / 151;
    hipLaunchKernelGGL(k_synth_reads, dim3(2048), dim3(256), 0, 0, g, genome, bases, 0ULL, n_reads, 150u, 350u, 0u, 1ULL);
    CK(hipDeviceSynchronize());
    DevTable t{};
    t.region_slots = 8192; t.n_regions = (uint32_t)((uint64_t)(genome / 0.6) / 8192 + 1); t.p1 = 1; t.p2 = t.n_regions; t.cap = (uint64_t)t.n_regions * t.region_slots; t.k = k; t.canonical = 1;
    CK(hipMalloc(&t.keys, t.cap * 8)); CK(hipMalloc(&t.counts, t.cap * 4)); CK(hipMalloc(&t.ovf_keys, OVF_CAP * 8)); CK(hipMalloc(&t.ovf_hi, OVF_CAP * 8)); CK(hipMalloc(&t.ctrs, CTR_WORDS * 8));
    CK(hipMemset(t.ovf_keys, 0xFF, OVF_CAP * 8)); CK(hipMemset(t.ovf_hi, 0, OVF_CAP * 8));
    unsigned long long* sink; CK(hipMalloc(&sink, 8)); CK(hipMemset(sink, 0, 8));
    const uint64_t nb = n_reads * 151, inst = n_reads * 124;
    printf("bases %llu, k-mer instances %llu, table %llu slots (%.1f MB)\n", (unsigned long long)nb, (unsigned long long)inst, (unsigned long long)t.cap, t.cap * 12 / 1e6);
    for (int grid : {2048, 1536}) {
        printf("grid %d\n", grid);
        for (int rep = 0; rep < 2; ++rep) {
            float f = run<FULL>(t, bases, nb, grid, sink, rep == 0);
            printf("  FULL%s          %8.2f ms  %6.2f G k-mers/s\n", rep ? " (2nd pass)" : " (1st pass)", f, inst / f / 1e6);
        }
        const char* names[] = {"", "EXTRACT_ONLY", "LOAD_ONLY", "ADD_NORET", "ADD_WG_SCOPE", "PLAIN_RMW", "ADD_ONLY_RET", "LOAD_NT", "LOAD_ADD_NORET", "PIPELINED"};
        float r[10];
        r[8] = run<LOAD_ADD_NORET>(t, bases, nb, grid, sink, false);
        r[9] = run<PIPELINED>(t, bases, nb, grid, sink, false);
        r[1] = run<EXTRACT_ONLY>(t, bases, nb, grid, sink, false);
        r[2] = run<LOAD_ONLY>(t, bases, nb, grid, sink, false);
        r[3] = run<ADD_NORET>(t, bases, nb, grid, sink, false);
        r[4] = run<ADD_WG_SCOPE>(t, bases, nb, grid, sink, false);
        r[5] = run<PLAIN_RMW>(t, bases, nb, grid, sink, false);
        r[6] = run<ADD_ONLY_RET>(t, bases, nb, grid, sink, false);
        r[7] = run<LOAD_NT>(t, bases, nb, grid, sink, false);
        for (int i = 1; i < 10; ++i) printf("  %-14s %8.2f ms  %6.2f G k-mers/s\n", names[i], r[i], inst / r[i] / 1e6);
    }
    return 0;
}
