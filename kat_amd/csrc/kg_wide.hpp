// kg_wide.hpp -- the kernels that exist only for wide tables (33 <= k <= 63; kg_device.hpp "wide keys"): counting, regrow, record
// merge, export, lookup, per-position profile.  The reducers (k_hist, k_total, k_gcp<true>, k_comp<PASS, true>, k_comp3_pass1<true>, k_comp3_pass3)
// are the narrow ones with the second key word read where the k-mer itself matters.
//
// Replaces the same reference code as the narrow kernels -- mer_iterator + multi-word mer_dna (mer_iterator.hpp:59-89,
// mer_dna.hpp:235-258,330-378) and hash_counter::add -- for k-mers that take two machine words.  k_count_w is the direct counter
// (global atomics, ~ the first-round kernel of the narrow path): small inputs and whatever the partitioned counter leaves; inputs of
// size go through kg_partition_wide.hpp (16-byte items through two radix levels, regions applied in LDS).
#pragma once
#include "kg_kernels.hpp"

namespace kg {

constexpr int WIDE_OVERLAP = 64;                                   // >= k-1 for k <= 63, keeps chunk starts 16-byte aligned
constexpr int WIDE_CHUNK_STARTS = CHUNK_BYTES - WIDE_OVERLAP;      // 4032 window starts per staged chunk
constexpr int WIDE_LANES_WITH_STARTS = WIDE_CHUNK_STARTS / BASES_PER_LANE;   // 252

// K1 for wide k-mers.  Same staging as k_count (one 16-byte load per lane, 2-bit codes + validity bits through LDS); lane t
// then owns the 16 window starts [16t, 16t+16) and needs bases 16t .. 16t+77: five code words = a 160-bit register window
// (hi, lo, nx) slid 16 times.  The k-mer is the top 2k bits of (hi, lo); its reverse complement is recomputed per window.
template <bool ALIGNED>
__global__ void __launch_bounds__(COUNT_BLOCK)
k_count_w(DevTable t, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_chunks) {
    __shared__ uint32_t s_code[COUNT_BLOCK + 4];
    __shared__ uint32_t s_bad[COUNT_BLOCK + 4];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = t.k;
    const bool canonical = t.canonical != 0;
    uint32_t new_distinct = 0;
    if (tid < 4) { s_code[COUNT_BLOCK + tid] = 0; s_bad[COUNT_BLOCK + tid] = 0xFFFF; }

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t off = chunk * WIDE_CHUNK_STARTS + (uint64_t)tid * BASES_PER_LANE;
        uint32_t w[4];
        if (ALIGNED && off + BASES_PER_LANE <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(bases + off);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint64_t i = off + q * 4 + b;
                    const uint32_t c = i < n ? bases[i] : (uint32_t)'N';     // past the end == separator
                    x |= c << (8 * b);
                }
                w[q] = x;
            }
        }
        uint32_t code, bad;
        encode16(w, code, bad);
        s_code[tid] = code;
        s_bad[tid] = bad;
        __syncthreads();

        if (tid < WIDE_LANES_WITH_STARTS) {
            uint64_t hi = ((uint64_t)s_code[tid] << 32) | s_code[tid + 1];       // bases 16t    .. 16t+31
            uint64_t lo = ((uint64_t)s_code[tid + 2] << 32) | s_code[tid + 3];   // bases 16t+32 .. 16t+63
            uint64_t nx = (uint64_t)s_code[tid + 4] << 32;                       // bases 16t+64 .. 16t+79
            uint64_t m = ((uint64_t)s_bad[tid] << 48) | ((uint64_t)s_bad[tid + 1] << 32) | ((uint64_t)s_bad[tid + 2] << 16) | s_bad[tid + 3];
            uint64_t mn = (uint64_t)s_bad[tid + 4] << 48;
            const uint32_t s = 128 - 2 * k, mshift = 64 - k;                     // s: 2..62, mshift: 1..31
#pragma unroll 2
            for (int j = 0; j < BASES_PER_LANE; ++j) {
                if ((m >> mshift) == 0) {                                        // k valid bases from this start (k <= 63 < 64 flags)
                    const uint64_t fhi = hi >> s, flo = (lo >> s) | (hi << (64 - s));
                    KeyW key = keyw_from_words(fhi, flo);
                    if (canonical) {
                        uint64_t rhi, rlo;
                        revcomp_words(fhi, flo, k, rhi, rlo);
                        if (rhi < fhi || (rhi == fhi && rlo < flo)) key = keyw_from_words(rhi, rlo);
                    }
                    table_add_w(t, key, 1, new_distinct);
                }
                hi = (hi << 2) | (lo >> 62);
                lo = (lo << 2) | (nx >> 62);
                nx <<= 2;
                m = (m << 1) | (mn >> 63);
                mn <<= 1;
            }
        }
        __syncthreads();
    }
    flush_distinct(t, new_distinct);
}

// K2 for wide tables: hash_counter::double_size (hash_counter.hpp:204-244)
static __global__ void __launch_bounds__(256)
k_regrow_w(DevTable dst, DevTable src, uint32_t src_n_ovf) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < src.cap; i += stride) {
        const uint64_t a = src.keys[i];
        if (a != EMPTY) table_add_w(dst, KeyW{a, src.keys_b[i]}, slot_count(src, i, i, src_n_ovf), new_distinct);
    }
    flush_distinct(dst, new_distinct);
}

// K7 for wide tables: (hi, lo, count) records into a table
static __global__ void __launch_bounds__(256)
k_merge_w(DevTable dst, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, const uint64_t* __restrict__ counts, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i]) table_add_w(dst, keyw_from_words(hi[i], lo[i]), counts[i], new_distinct);
    flush_distinct(dst, new_distinct);
}

// every (k-mer, count) of the table, in slot order: a wave compacts its occupied lanes with one ballot and one cursor add
static __global__ void __launch_bounds__(256)
k_export_w(DevTable t, uint32_t n_ovf, uint64_t* __restrict__ hi, uint64_t* __restrict__ lo, uint64_t* __restrict__ counts, unsigned long long* cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rounds = (t.cap + stride - 1) / stride;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = first + r * stride;
        const uint64_t a = i < t.cap ? t.keys[i] : EMPTY;
        const bool occ = a != EMPTY;
        const unsigned long long live = __ballot(occ);
        if (!live) continue;
        unsigned long long base = 0;
        if (lane == (uint32_t)(__ffsll((long long)live) - 1)) base = atomicAdd(cursor, (unsigned long long)__popcll(live));
        base = __shfl(base, __ffsll((long long)live) - 1, 64);
        if (occ) {
            const uint64_t at = base + __popcll(live & ((1ULL << lane) - 1));
            const KeyW kw{a, t.keys_b[i]};
            hi[at] = keyw_hi(kw); lo[at] = keyw_lo(kw); counts[at] = slot_count(t, i, i, n_ovf);
        }
    }
}

// owner-partitioned export for the multi-GPU merge (k_partition for wide tables).  MODE 0: records per part; MODE 1: scatter
// (hi, lo, count) records to cursors[part]++.
template <int MODE>
__global__ void __launch_bounds__(256)
k_partition_w(DevTable t, uint32_t n_ovf, uint32_t n_parts, unsigned long long* __restrict__ sizes_or_cursors,
              uint64_t* __restrict__ out_hi, uint64_t* __restrict__ out_lo, uint64_t* __restrict__ out_counts) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.cap; i += stride) {
        const uint64_t a = t.keys[i];
        if (a == EMPTY) continue;
        const KeyW kw{a, t.keys_b[i]};
        const uint32_t part = n_parts > 1 ? owner_of_w(kw, t.k, n_parts) : 0;
        const unsigned long long at = atomicAdd(&sizes_or_cursors[part], 1ULL);
        if (MODE == 1) { out_hi[at] = keyw_hi(kw); out_lo[at] = keyw_lo(kw); out_counts[at] = slot_count(t, i, i, n_ovf); }
    }
}

// batch lookup (JellyfishHelper::getCount, lib/src/jellyfish_helper.cc:189-194)
static __global__ void __launch_bounds__(256)
k_get_w(DevTable t, uint32_t n_ovf, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n, int canonicalise, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KeyW kw = keyw_from_words(hi[i], lo[i]);
    if (canonicalise) kw = keyw_canonical(kw, t.k);
    out[i] = table_get_w(t, kw, n_ovf);
}

// K8 for wide tables: per-position coverage (kat sect / kat cold; src/sect.cc:516-535).  k_profile's shape with k_count_w's
// 160-bit window: 16 counts per lane = one 128-byte line of `out`.
template <bool ALIGNED>
__global__ void __launch_bounds__(COUNT_BLOCK)
k_profile_w(DevTable t, uint32_t n_ovf, int canonicalise, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_chunks,
            uint64_t* __restrict__ out) {
    __shared__ uint32_t s_code[COUNT_BLOCK + 4];
    __shared__ uint32_t s_bad[COUNT_BLOCK + 4];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = t.k;
    const uint64_t n_out = n - k + 1;
    if (tid < 4) { s_code[COUNT_BLOCK + tid] = 0; s_bad[COUNT_BLOCK + tid] = 0xFFFF; }

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t off = chunk * WIDE_CHUNK_STARTS + (uint64_t)tid * BASES_PER_LANE;
        uint32_t w[4];
        if (ALIGNED && off + BASES_PER_LANE <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(bases + off);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint64_t i = off + q * 4 + b;
                    const uint32_t c = i < n ? bases[i] : (uint32_t)'N';
                    x |= c << (8 * b);
                }
                w[q] = x;
            }
        }
        uint32_t code, bad;
        encode16(w, code, bad);
        s_code[tid] = code;
        s_bad[tid] = bad;
        __syncthreads();

        if (tid < WIDE_LANES_WITH_STARTS && off < n_out) {
            uint64_t hi = ((uint64_t)s_code[tid] << 32) | s_code[tid + 1];
            uint64_t lo = ((uint64_t)s_code[tid + 2] << 32) | s_code[tid + 3];
            uint64_t nx = (uint64_t)s_code[tid + 4] << 32;
            uint64_t m = ((uint64_t)s_bad[tid] << 48) | ((uint64_t)s_bad[tid + 1] << 32) | ((uint64_t)s_bad[tid + 2] << 16) | s_bad[tid + 3];
            uint64_t mn = (uint64_t)s_bad[tid + 4] << 48;
            const uint32_t s = 128 - 2 * k, mshift = 64 - k;
            uint64_t c[BASES_PER_LANE];
#pragma unroll
            for (int j = 0; j < BASES_PER_LANE; ++j) {
                c[j] = 0;
                if ((m >> mshift) == 0) {
                    KeyW key = keyw_from_words(hi >> s, (lo >> s) | (hi << (64 - s)));
                    if (canonicalise) key = keyw_canonical(key, k);
                    c[j] = table_get_w(t, key, n_ovf);
                }
                hi = (hi << 2) | (lo >> 62);
                lo = (lo << 2) | (nx >> 62);
                nx <<= 2;
                m = (m << 1) | (mn >> 63);
                mn <<= 1;
            }
            if (off + BASES_PER_LANE <= n_out) {
                ulonglong2* o = reinterpret_cast<ulonglong2*>(out + off);           // off is a multiple of 16: 128-byte aligned
#pragma unroll
                for (int j = 0; j < BASES_PER_LANE / 2; ++j) o[j] = make_ulonglong2(c[2 * j], c[2 * j + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < BASES_PER_LANE; ++j) if (off + j < n_out) out[off + j] = c[j];
            }
        }
        __syncthreads();
    }
}

}  // namespace kg
