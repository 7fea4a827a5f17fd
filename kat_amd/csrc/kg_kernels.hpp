// kg_kernels.hpp -- the HIP kernels of the kat hist/gcp/comp hot path, written for gfx950 (CDNA4, wave64).
//
//  K1 k_count          extract + canonicalise + 2-bit pack + hash + insert      (replaces mer_iterator + hash_counter::add)
//  K2 k_regrow         re-insert a table into a larger one                       (replaces hash_counter::double_size)
//  K3 k_hist           slot scan -> histogram                                    (replaces Histogram::binSlice)
//  K4 k_gcp            slot scan -> GC x coverage matrix                         (replaces Gcp::analyseSlice)
//  K5 k_comp_pass1/2   slot scan + probe of the other table -> matrix, counters  (replaces Comp::compareSlice)
//  K6 k_part_*         owner-partitioned export for the multi-GPU merge
//  K7 k_merge          add (key,count) records into a table
//
// All of this is integer / byte work bound by HBM (random 8-16 B slot accesses for K1/K5, streaming for K3/K4);
// none of it is a contraction, so no MFMA anywhere.
#pragma once
#include "kg_device.hpp"
#include <type_traits>

namespace kg {

constexpr int COUNT_BLOCK = 256;                 // 4 waves
constexpr int BASES_PER_LANE = 16;               // one 16-byte global load per lane
constexpr int CHUNK_BYTES = COUNT_BLOCK * BASES_PER_LANE;        // 4096 bytes staged per block iteration
constexpr int CHUNK_OVERLAP = 32;                // >= k-1 for k <= 32, keeps chunk starts 16-byte aligned
constexpr int CHUNK_STARTS = CHUNK_BYTES - CHUNK_OVERLAP;        // 4064 window start positions per chunk
constexpr int LANES_WITH_STARTS = CHUNK_STARTS / BASES_PER_LANE; // 254

// 16 ASCII bytes -> 16 two-bit codes (MSB-first in a u32) + 16 "not ACGTacgt" flags (MSB-first in the low 16 bits).
// code = x ^ (x >> 1) with x = (c >> 1) & 3 maps A,C,G,T (either case) to 0,1,2,3 (mer_dna.hpp:46-63).
// Four bytes at a time (a byte-by-byte form was 200 of level 1's 1750 vector instructions per wave and tile): x per byte; the letter
// that x stands for, looked up by v_perm_b32 with x as the selector; a byte that is not that letter (case folded) is flagged; the
// four 2-bit codes / four flags of a word are gathered by one multiply each (the partial products land on distinct bits: no carries).
__device__ __forceinline__ void encode16(const uint32_t w[4], uint32_t& code, uint32_t& bad) {
    code = 0; bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t v = w[q];
        const uint32_t x = (v >> 1) & 0x03030303u;                                   // A, C, T, G -> 0, 1, 2, 3
        const uint32_t c2 = x ^ ((x >> 1) & 0x01010101u);                            // A, C, G, T -> 0, 1, 2, 3
        const uint32_t letter = __builtin_amdgcn_perm(0u, 0x67746361u, x);           // 'a', 'c', 't', 'g' by x
        const uint32_t diff = (v | 0x20202020u) ^ letter;
        const uint32_t nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;    // 1 per byte that is no such letter
        code = (code << 8) | ((c2 * 0x40100401u) >> 24);                             // byte 0 (the first base) into the top pair
        bad = (bad << 4) | ((nz * 0x08040201u) >> 24);
    }
}

// K1.  One block iteration stages 4096 consecutive bytes of the base stream: each lane issues one coalesced 16-byte
// load (1 KiB per wave instruction), packs it to 32 code bits + 16 validity bits and parks both in LDS.  After the
// barrier lane t owns the 16 window starts [16t, 16t+16): it pulls three consecutive code words (48 bases) from
// LDS into a 96-bit register window and slides it 16 times -- no per-base loop, no re-reading of HBM.  The
// reverse complement is recomputed per window with v_bfrev (6 VALU ops) rather than rolled.
template <bool ALIGNED>
__global__ void __launch_bounds__(COUNT_BLOCK)
k_count(DevTable t, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_chunks) {
    __shared__ uint32_t s_code[COUNT_BLOCK + 2];
    __shared__ uint32_t s_bad[COUNT_BLOCK + 2];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = t.k;
    const bool canonical = t.canonical != 0;
    uint32_t new_distinct = 0;
    if (tid < 2) { s_code[COUNT_BLOCK + tid] = 0; s_bad[COUNT_BLOCK + tid] = 0xFFFF; }

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t off = chunk * CHUNK_STARTS + (uint64_t)tid * BASES_PER_LANE;
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
                    uint64_t i = off + q * 4 + b;
                    uint32_t c = i < n ? bases[i] : (uint32_t)'N';       // past the end == separator
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

        if (tid < LANES_WITH_STARTS) {
            uint64_t hi = ((uint64_t)s_code[tid] << 32) | s_code[tid + 1];   // bases 16t .. 16t+31
            uint64_t lo = (uint64_t)s_code[tid + 2] << 32;                   // bases 16t+32 .. 16t+47
            uint64_t m = ((uint64_t)s_bad[tid] << 48) | ((uint64_t)s_bad[tid + 1] << 32) | ((uint64_t)s_bad[tid + 2] << 16);
            const uint32_t kshift = 64 - 2 * k, mshift = 64 - k;
#pragma unroll 4
            for (int j = 0; j < BASES_PER_LANE; ++j) {
                if ((m >> mshift) == 0) {                                    // k valid bases from this start
                    uint64_t fwd = hi >> kshift;
                    uint64_t key = fwd;
                    if (canonical) { uint64_t rc = kmer_revcomp(fwd, k); key = rc < fwd ? rc : fwd; }
                    table_inc(t, key, new_distinct);
                }
                hi = (hi << 2) | (lo >> 62);
                lo <<= 2;
                m <<= 1;
            }
        }
        __syncthreads();
    }
    flush_distinct(t, new_distinct);
}

// K2 / K7: add (key,count) records into a table.  src == another table's slots (regrow) or a record list (merge).
static __global__ void __launch_bounds__(256)
k_regrow(DevTable dst, DevTable src, uint32_t src_n_ovf) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < src.cap; i += stride) {
        const SlotView v = slot_view(src, i);
        if (v.occ) table_add(dst, v.key, slot_total(src, i, v.key, v.cnt, src_n_ovf), new_distinct);
    }
    flush_distinct(dst, new_distinct);
}

static __global__ void __launch_bounds__(256)
k_merge(DevTable dst, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ counts, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i]) table_add(dst, keys[i], counts[i], new_distinct);
    flush_distinct(dst, new_distinct);
}

// Overflow guard for table_inc's unchecked 32-bit adds (KV12 tables; packed tables add checked): every counter >= thr gives thr to
// the side table.  Reports the largest counter left behind (scratch[0], atomicMax) so the host knows how many more unchecked adds are safe.
static __global__ void __launch_bounds__(256)
k_sweep(DevTable t, uint32_t thr, unsigned long long* scratch) {
    uint32_t mx = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.cap; i += stride) {
        uint32_t c = t.counts[i];
        if (c >= thr) {
            uint64_t key = t.keys[i];
            uint32_t take = (c / thr) * thr;
            c -= take;
            t.counts[i] = c;
            ovf_add(t, key, take);
        }
        mx = c > mx ? c : mx;
    }
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_down(mx, off, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) atomicMax(scratch, (unsigned long long)mx);
}

// Σ counts (table_stats "total")
static __global__ void __launch_bounds__(256)
k_total(DevTable t, uint32_t n_ovf, uint64_t* out) {
    uint64_t s = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.cap; i += stride) {
        const uint64_t w = t.keys[i];
        if (t.cbits ? w != 0 : w != EMPTY) s += slot_total(t, i, w, t.cbits ? pk_count(w, t.cbits) : (uint64_t)t.counts[i], n_ovf);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}

// occupied slots of [pos_lo, pos_hi): what a share of the table holds, when the table's load as a whole does not say (merge_regions_impl)
static __global__ void __launch_bounds__(256)
k_occupied(DevTable t, uint64_t pos_lo, uint64_t pos_hi, unsigned long long* out) {
    uint32_t s = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = pos_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pos_hi; i += stride) {
        const uint64_t w = t.keys[i];
        s += (t.cbits ? w != 0 : w != EMPTY) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, (unsigned long long)s);
}

// LDS-privatised increment with wave-level aggregation: lanes that hit the same bin as the wave's first active
// lane are folded into one LDS atomic (ballot + popcount); the rest fall through to their own atomic.  K-mer
// spectra are dominated by a handful of bins (singletons!), so this removes most same-address serialisation.
__device__ __forceinline__ void lds_inc_aggregated(uint32_t* bins, uint32_t idx, bool active) {
    unsigned long long live = __ballot(active);
    if (!live) return;
    int leader = __ffsll((long long)live) - 1;
    uint32_t lead_idx = lane_value(idx, leader);
    bool same = active && idx == lead_idx;
    unsigned long long peers = __ballot(same);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&bins[lead_idx], (uint32_t)__popcll(peers));
    if (active && !same) atomicAdd(&bins[idx], 1u);
}

// K3.  Histogram::binSlice (src/histogram.cc:183-199).  The first lds_bins buckets live in LDS (u32, flushed once
// per block); anything above goes straight to the global u64 array.
// Slot scan of the reducers: four consecutive slots per lane and step, as 16-byte loads (two for the keys, one for the counts;
// cap is a multiple of 4 and both arrays are 256-byte aligned), 1024-thread workgroups so that the one or two workgroups a CU can
// hold next to their LDS-privatised result still put 16-32 waves on it.  (The first edition read 8 + 4 bytes per lane from 256-thread
// workgroups, one or two per CU: 0.8-1.5 TB/s.)
constexpr int SCAN_BLOCK = 1024;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
struct Slots4 { uint64_t w[4]; uint64_t cnt[4]; bool occ[4]; };     // w: the k-mer (KV12, wide: its first word) or the packed word
// (the loads are unconditional, from a clamped address, and the lanes beyond the range are masked afterwards: loads issued inside
// a branch make hipcc wait for them at the end of the branch, which turns every prefetch into a stall)
struct Raw4 { u64x2 a, b; u32x4s c; };                               // four slots as they come from HBM (c: KV12 counts)
template <bool PK>
__device__ __forceinline__ Raw4 load_raw4(const DevTable& t, uint64_t i4 /* first slot, multiple of 4 */, bool in_range) {
    Raw4 r;
    const uint64_t at = in_range ? i4 : 0;
    r.a = *reinterpret_cast<const u64x2*>(t.keys + at); r.b = *reinterpret_cast<const u64x2*>(t.keys + at + 2);
    if constexpr (!PK) r.c = *reinterpret_cast<const u32x4s*>(t.counts + at); else r.c = u32x4s{0u, 0u, 0u, 0u};
    return r;
}
template <bool PK>
__device__ __forceinline__ Slots4 decode_slots4(const DevTable& t, const Raw4& raw, bool in_range) {
    Slots4 r;
    r.w[0] = raw.a.x; r.w[1] = raw.a.y; r.w[2] = raw.b.x; r.w[3] = raw.b.y;
    if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { r.occ[j] = in_range && r.w[j] != 0; r.cnt[j] = pk_count(r.w[j], t.cbits); }
    } else {
        r.cnt[0] = raw.c.x; r.cnt[1] = raw.c.y; r.cnt[2] = raw.c.z; r.cnt[3] = raw.c.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.occ[j] = in_range && r.w[j] != EMPTY;
    }
    return r;
}
template <bool PK>
__device__ __forceinline__ Slots4 load_slots4(const DevTable& t, uint64_t i4, bool in_range) { return decode_slots4<PK>(t, load_raw4<PK>(t, i4, in_range), in_range); }

template <bool PK>
__global__ void __launch_bounds__(SCAN_BLOCK)
k_hist(DevTable t, uint32_t n_ovf, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t nb,
       unsigned long long* __restrict__ out, uint32_t lds_bins) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bins[];
    for (uint32_t i = threadIdx.x; i < lds_bins; i += blockDim.x) s_bins[i] = 0;
    __syncthreads();
    const uint64_t quads = t.cap / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rounds = (quads + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {              // uniform trip count: the aggregation uses whole-wave ballots
        const uint64_t q = first + r * stride;
        const Slots4 s4 = load_slots4<PK>(t, q * 4, q < quads);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool occ = s4.occ[j];
            uint64_t idx = 0;
            if (occ) {
                const uint64_t v = slot_total(t, q * 4 + j, s4.w[j], s4.cnt[j], n_ovf);
                idx = v < base ? 0 : (v > ceil_ ? nb - 1 : (inc == 1 ? v - base : (v - base) / inc));
            }
            const bool in_lds = occ && idx < lds_bins;
            lds_inc_aggregated(s_bins, (uint32_t)idx, in_lds);
            if (occ && !in_lds) atomicAdd(&out[idx], 1ULL);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // the all-ones key lives outside the slots
        uint64_t v = t.ctrs[CTR_ONES];
        if (v) { uint64_t idx = v < base ? 0 : (v > ceil_ ? nb - 1 : (v - base) / inc); atomicAdd(&out[idx], 1ULL); }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lds_bins; i += blockDim.x)
        if (s_bins[i]) atomicAdd(&out[i], (unsigned long long)s_bins[i]);
}

// ceil((double)count * scale) exactly as the host does it (src/gcp.cc:190, src/comp.hpp:303-306): one IEEE
// double multiply, one ceil, no contraction possible.
__device__ __forceinline__ uint64_t scale_count(uint64_t c, double scale) {
    if (scale == 1.0) return c;            // (the default; uniform.  Every caller clamps to its bins, so counts beyond 2^53 end up the same)
    return c == 0 ? 0 : (uint64_t)ceil((double)c * scale);
}
// the same on a count known to fit 32 bits (a packed slot's, with no side-table entries): saturates, and every caller clamps to its bins
__device__ __forceinline__ uint32_t scale_count(uint32_t c, double scale) {
    if (scale == 1.0) return c;
    if (c == 0) return 0;
    const double v = ceil((double)c * scale);
    return v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
}

// K4.  Gcp::analyseSlice (src/gcp.cc:179-197): row = popcount-based GC count, column = min(ceil(count*scale), bins).
// The whole k x (bins+1) matrix is privatised in LDS as u32 when it fits (27 x 1001 x 4 B = 108 KB of the 160 KB).
template <bool W, bool PK>                                // W: wide table (k > 32, kg_device.hpp "wide keys"); PK: packed slots
__global__ void __launch_bounds__(SCAN_BLOCK)
k_gcp(DevTable t, uint32_t n_ovf, double scale, uint32_t bins, unsigned long long* __restrict__ out, uint32_t use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bins[];
    const uint32_t cols = bins + 1, k = t.k, cells = k * cols;
    if (use_lds) { for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) s_bins[i] = 0; __syncthreads(); }
    const uint64_t quads = t.cap / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rounds = (quads + stride - 1) / stride;
    // The next round's slots are asked for before this round's are worked on: with the matrix in LDS a CU holds ONE workgroup, 16 waves,
    // and a load per wave in flight is 32 KB per CU -- 8 MB on the chip, what 3.5 TB/s keep in flight (k_hist, two workgroups: 5.7 TB/s).
    Raw4 ahead = load_raw4<PK>(t, first * 4, first < quads);
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t q = first + r * stride;
        const bool in_range = q < quads;
        const Raw4 raw = ahead;
        ahead = load_raw4<PK>(t, (q + stride) * 4, q + stride < quads);
        const Slots4 s4 = decode_slots4<PK>(t, raw, in_range);
        uint64_t kb[4] = {0, 0, 0, 0};
        RegionPlace rp{};
        if constexpr (PK) {                                // the quad lies in one region (regions are whole quads): its digits once
            const uint32_t region = region_of_pos(t, in_range ? q * 4 : 0);
            rp.pl = place_make(t.k, t.p1, t.n1, t.l2);
            rp.d1 = region >> t.l2;
            rp.d2 = region & (t.p2 - 1);
        }
        if constexpr (W) {
            const uint64_t at = in_range ? q * 4 : 0;
            const u64x2 a = *reinterpret_cast<const u64x2*>(t.keys_b + at), b = *reinterpret_cast<const u64x2*>(t.keys_b + at + 2);
            kb[0] = a.x; kb[1] = a.y; kb[2] = b.x; kb[3] = b.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool occ = s4.occ[j];
            uint32_t cell = 0;
            if (occ) {
                uint32_t g;
                if constexpr (W) g = keyw_gc(KeyW{s4.w[j], kb[j]}, k);
                else if constexpr (PK) g = kmer_gc(key_in(pk_rem(s4.w[j], t.cbits), rp), k);
                else g = kmer_gc(s4.w[j], k);
                uint64_t pos = scale_count(slot_total(t, q * 4 + j, s4.w[j], s4.cnt[j], n_ovf), scale);
                if (pos > bins) pos = bins;
                occ = g < k;                                   // the reference's matrix has k rows: GC == k never printed
                cell = g * cols + (uint32_t)pos;
            }
            if (use_lds == 2) { if (occ) (void)__hip_atomic_fetch_add(&s_bins[cell], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (one atomic per lane: kg_reduce.hip, KATGPU_COMP_PLAIN_INC)
            else if (use_lds) lds_inc_aggregated(s_bins, cell, occ);
            else if (occ) atomicAdd(&out[cell], 1ULL);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t v = t.ctrs[CTR_ONES];                     // all-T 32-mer: GC count 0
        if (v) { uint64_t pos = scale_count(v, scale); if (pos > bins) pos = bins; atomicAdd(&out[pos], 1ULL); }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x)
            if (s_bins[i]) atomicAdd(&out[i], (unsigned long long)s_bins[i]);
    }
}

// K4 for packed tables: a WAVE takes a region at a time (quads of four slots, 64 lanes wide, the next load in flight behind the work on
// this one).  The k-mer of a packed slot is its remainder joined with the region's two digits (key_in): per region those are scalars,
// where the flat scan above found them per quad through region_of_pos (a double multiply and two 64-bit checks), and the count is the
// slot's low 32 bits unless the table has side-table entries (n_ovf, uniform).  3.25 -> 3.07 ms from the load ahead alone at config 3.
static __global__ void __launch_bounds__(SCAN_BLOCK)
k_gcp_pk(DevTable t, uint32_t n_ovf, double scale, uint32_t bins, unsigned long long* __restrict__ out, uint32_t use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bins[];
    const uint32_t cols = bins + 1, k = t.k, cells = k * cols;
    if (use_lds) { for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) s_bins[i] = 0; __syncthreads(); }
    const uint32_t lane = threadIdx.x & 63, S = t.region_slots, R = t.n_regions, qpr = S / 4;      // (S is a multiple of 4)
    const uint32_t n_waves = gridDim.x * (SCAN_BLOCK / 64);
    uint32_t reg = __builtin_amdgcn_readfirstlane(blockIdx.x * (SCAN_BLOCK / 64) + (threadIdx.x >> 6)), q0 = 0;
    RegionPlace rp{};
    rp.pl = place_make(t.k, t.p1, t.n1, t.l2);
    const uint32_t cmask32 = (uint32_t)pk_cmask(t.cbits);                                      // cbits <= 32
    auto load_at = [&](uint32_t rg, uint32_t qq) { const uint32_t q = qq + lane; return load_raw4<true>(t, (uint64_t)rg * S + (uint64_t)(q < qpr ? q : 0) * 4, true); };
    Raw4 ahead{};
    if (reg < R) ahead = load_at(reg, 0);
    while (reg < R) {
        uint32_t nreg = reg, nq0 = q0 + 64;
        if (nq0 >= qpr) { nq0 = 0; nreg = reg + n_waves; }
        const Raw4 raw = ahead;
        ahead = load_at(nreg < R ? nreg : reg, nq0);
        rp.d1 = reg >> t.l2; rp.d2 = reg & (t.p2 - 1);
        const uint32_t q = q0 + lane;
        const bool in_range = q < qpr;
        const uint64_t w4[4] = {raw.a.x, raw.a.y, raw.b.x, raw.b.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool occ = in_range && w4[j] != 0;
            uint32_t cell = 0;
            if (occ) {
                const uint32_t g = kmer_gc(key_in(pk_rem(w4[j], t.cbits), rp), k);
                uint32_t pos;
                if (n_ovf == 0) { pos = scale_count((uint32_t)w4[j] & cmask32, scale); pos = pos > bins ? bins : pos; }
                else {
                    uint64_t p64 = scale_count(slot_total(t, (uint64_t)reg * S + (uint64_t)q * 4 + j, w4[j], pk_count(w4[j], t.cbits), n_ovf), scale);
                    pos = p64 > bins ? bins : (uint32_t)p64;
                }
                occ = g < k;                                   // the reference's matrix has k rows: GC == k never printed
                cell = g * cols + pos;
            }
            if (use_lds == 2) { if (occ) (void)__hip_atomic_fetch_add(&s_bins[cell], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (use_lds) lds_inc_aggregated(s_bins, cell, occ);
            else if (occ) atomicAdd(&out[cell], 1ULL);
        }
        reg = nreg; q0 = nq0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {             // (a packed table keeps every k-mer in a slot; the counter is read as the flat scan reads it)
        uint64_t v = t.ctrs[CTR_ONES];
        if (v) { uint64_t pos = scale_count(v, scale); if (pos > bins) pos = bins; atomicAdd(&out[pos], 1ULL); }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x)
            if (s_bins[i]) atomicAdd(&out[i], (unsigned long long)s_bins[i]);
    }
}

// ---- K5: comp ----
constexpr uint32_t COMP_TILE = 64;     // main-matrix cells [0,64) x [0,64) are privatised in LDS (u32)
constexpr int JOIN_BLOCK = 1024;       // the join form: two workgroups per CU beside their LDS (result tile + spectra + the probed region)
enum { CC_H1_TOTAL, CC_H2_TOTAL, CC_H3_TOTAL, CC_H1_DISTINCT, CC_H2_DISTINCT, CC_H3_DISTINCT, CC_H1_ONLY_TOTAL,
       CC_H2_ONLY_TOTAL, CC_H1_ONLY_DISTINCT, CC_H2_ONLY_DISTINCT, CC_SH_H1_TOTAL, CC_SH_H2_TOTAL, CC_SH_DISTINCT };

struct CompArgs {
    double d1_scale, d2_scale;
    uint32_t d1_bins, d2_bins, spec_size;
    uint32_t canon_probe;            // pass 1: canonicalise the probe key iff input 2 is canonical (src/comp.cc:401)
    unsigned long long* main_mx;     // d1_bins x d2_bins
    unsigned long long* counters;    // 13
    unsigned long long* spectra;     // 4 x spec_size
    uint32_t* seen;                  // join pass 1 -> pass 2: one bit per slot of hash 2, set when hash 1 holds the slot's key (null: off);
    uint32_t seen_wpr;               // words per region of that bitmap
    uint32_t fold;                   // pass 1: spectra of the k-mers that land in the LDS tile are taken from the tile at the end (below)
    uint32_t plain_inc;              // LDS increments as one no-return atomic per lane (the default) instead of ballot-aggregated per wave (kg_reduce.hip: KATGPU_COMP_PLAIN_INC)
};

__device__ __forceinline__ void comp_inc(const CompArgs& a, uint32_t* bins, uint32_t idx, bool active) {
    if (a.plain_inc) { if (active) (void)__hip_atomic_fetch_add(&bins[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    else lds_inc_aggregated(bins, idx, active);
}
__device__ __forceinline__ uint32_t spectrum_bin(uint64_t c, uint32_t size) { return c >= size ? size - 1 : (uint32_t)c; }  // comp_counters.cc:130-140
__device__ __forceinline__ uint32_t spectrum_bin(uint32_t c, uint32_t size) { return c >= size ? size - 1 : c; }

__device__ __forceinline__ void block_sum_u64(unsigned long long* s_acc, int slot, uint64_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_acc[slot], (unsigned long long)v);
}

// Per-lane running sums of one comp pass
// (the tallies of k-mers are per lane: a lane sees fewer than 2^32 slots of any table that fits the device; the sums of counts stay 64-bit)
struct CompAcc { uint64_t a_total = 0, a_only_total = 0, sh_a = 0, sh_b = 0; uint32_t a_distinct = 0, a_only_distinct = 0, sh_n = 0; };

// The two increments nearly every lane of a wave wants at once: "seen once here, absent there" and "count 1" -- the sequencing-error
// k-mers of a read set, the unique k-mers of an assembly.  As LDS atomics they are 64 lanes on ONE address, served one lane a cycle
// (24 % of k_comp_fused's cycles at config 4 were LDS bank-conflict cycles); a CompHot keeps them as a per-lane tally instead, added to
// the tile / the spectrum once at the end of the kernel (comp_hot_flush).  Which cell and bin those are follows from the scales, so the
// kernel asks comp_cell for them rather than assuming (1, 0).
struct CompHot { uint32_t tile_cell = 0xFFFFFFFFu, spec_bin = 0xFFFFFFFFu, n_tile = 0, n_spec = 0; };

// What one k-mer of the scanned table contributes once its count in the other table (cb) is known.  All lanes of the
// wave call this together (the LDS increments are wave-aggregated with ballots); occ says whether the lane holds a k-mer.
template <int PASS, bool HOT = false, typename T = uint64_t /* the counts' type: uint32_t where the caller knows they fit (k_comp_fused without side-table entries) */>
__device__ __forceinline__ void comp_account(bool occ, T ca, T cb, const CompArgs& a, uint32_t* s_tile, uint32_t* s_spec, CompAcc& acc, CompHot* hot = nullptr) {
    uint32_t cell = 0;
    bool in_tile = false, in_mx = false;
    if (occ) {
        acc.a_total += ca; acc.a_distinct += 1;
        if (!cb) { acc.a_only_total += ca; acc.a_only_distinct += 1; }
        if (PASS == 1) {
            if (ca && cb) { acc.sh_a += ca; acc.sh_b += cb; acc.sh_n += 1; }
            T s1 = scale_count(ca, a.d1_scale), s2 = scale_count(cb, a.d2_scale);
            if (s1 >= a.d1_bins) s1 = a.d1_bins - 1;
            if (s2 >= a.d2_bins) s2 = a.d2_bins - 1;
            in_mx = true;
            in_tile = s1 < COMP_TILE && s2 < COMP_TILE;
            cell = in_tile ? (uint32_t)s1 * COMP_TILE + (uint32_t)s2 : (uint32_t)s1 * a.d2_bins + (uint32_t)s2;
        } else if (!cb) {                                                  // only k-mers absent from hash 1 (src/comp.cc:453-462)
            T s2 = scale_count(ca, a.d2_scale);
            if (s2 >= a.d2_bins) s2 = a.d2_bins - 1;
            in_mx = true;
            in_tile = s2 < COMP_TILE;
            cell = (uint32_t)s2;                                            // row 0 in both the tile and the matrix
        }
    }
    bool to_tile = in_mx && in_tile;
    if constexpr (HOT) { const bool h = to_tile && cell == hot->tile_cell; hot->n_tile += h ? 1u : 0u; to_tile = to_tile && !h; }
    comp_inc(a, s_tile, cell, to_tile);
    if (in_mx && !in_tile) atomicAdd(&a.main_mx[cell], 1ULL);
    // CompArgs::fold (unscaled bins, more than COMP_TILE of them): a k-mer of pass 1 that lands in the tile has s1 == ca < 64 and
    // s2 == cb < 64, so spectrum1, shared_spectrum1 and shared_spectrum2 are marginals of the tile -- comp_flush adds them; one LDS
    // atomic per k-mer instead of four (the spectra's hot bins are the same few addresses for every lane of the chip)
    if (PASS == 1 && a.fold) occ = occ && !in_tile;
    const uint32_t sbin = spectrum_bin(ca, a.spec_size);
    bool to_spec = occ;
    if constexpr (HOT) { const bool h = to_spec && sbin == hot->spec_bin; hot->n_spec += h ? 1u : 0u; to_spec = to_spec && !h; }
    comp_inc(a, s_spec, sbin, to_spec);                                                 // spectrum1 / spectrum2
    if (PASS == 1) {
        bool shared = occ && ca && cb;
        comp_inc(a, s_spec + a.spec_size, spectrum_bin(ca, a.spec_size), shared);         // shared_spectrum1
        comp_inc(a, s_spec + 2 * a.spec_size, spectrum_bin(cb, a.spec_size), shared);     // shared_spectrum2
    }
}

// the tile cell comp_account<PASS> gives a k-mer of these counts (0xFFFFFFFF: none, or outside the tile): where a CompHot listens
template <int PASS>
__device__ __forceinline__ uint32_t comp_tile_cell(uint64_t ca, uint64_t cb, const CompArgs& a) {
    if (PASS == 1) {
        uint64_t s1 = scale_count(ca, a.d1_scale), s2 = scale_count(cb, a.d2_scale);
        if (s1 >= a.d1_bins) s1 = a.d1_bins - 1;
        if (s2 >= a.d2_bins) s2 = a.d2_bins - 1;
        return s1 < COMP_TILE && s2 < COMP_TILE ? (uint32_t)(s1 * COMP_TILE + s2) : 0xFFFFFFFFu;
    }
    if (cb) return 0xFFFFFFFFu;
    uint64_t s2 = scale_count(ca, a.d2_scale);
    if (s2 >= a.d2_bins) s2 = a.d2_bins - 1;
    return s2 < COMP_TILE ? (uint32_t)s2 : 0xFFFFFFFFu;
}
// a CompHot's tallies into the tile and the spectrum they were kept out of: one LDS atomic per wave and target (before comp_flush)
__device__ __forceinline__ void comp_hot_flush(const CompHot& h, uint32_t* s_tile, uint32_t* s_spec) {
    uint32_t nt = h.n_tile, ns = h.n_spec;
    for (int off = 32; off > 0; off >>= 1) { nt += __shfl_down(nt, off, 64); ns += __shfl_down(ns, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (nt) atomicAdd(&s_tile[h.tile_cell], nt);
        if (ns) atomicAdd(&s_spec[h.spec_bin], ns);
    }
}

// block-level flush of the accumulators, the matrix tile and the spectra (end of a comp kernel)
template <int PASS, bool TILE = true /* false: the tile is another pass's to flush (k_comp_fused) */>
__device__ __forceinline__ void comp_flush(const CompArgs& a, unsigned long long* s_acc, uint32_t* s_tile, uint32_t* s_spec, const CompAcc& acc) {
    const uint32_t n_spec = (PASS == 1 ? 3u : 1u) * a.spec_size;
    block_sum_u64(s_acc, 0, acc.a_total); block_sum_u64(s_acc, 1, acc.a_distinct);
    block_sum_u64(s_acc, 2, acc.a_only_total); block_sum_u64(s_acc, 3, acc.a_only_distinct);
    if (PASS == 1) { block_sum_u64(s_acc, 4, acc.sh_a); block_sum_u64(s_acc, 5, acc.sh_b); block_sum_u64(s_acc, 6, acc.sh_n); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (PASS == 1) {
            atomicAdd(&a.counters[CC_H1_TOTAL], s_acc[0]); atomicAdd(&a.counters[CC_H1_DISTINCT], s_acc[1]);
            atomicAdd(&a.counters[CC_H1_ONLY_TOTAL], s_acc[2]); atomicAdd(&a.counters[CC_H1_ONLY_DISTINCT], s_acc[3]);
            atomicAdd(&a.counters[CC_SH_H1_TOTAL], s_acc[4]); atomicAdd(&a.counters[CC_SH_H2_TOTAL], s_acc[5]);
            atomicAdd(&a.counters[CC_SH_DISTINCT], s_acc[6]);
        } else {
            atomicAdd(&a.counters[CC_H2_TOTAL], s_acc[0]); atomicAdd(&a.counters[CC_H2_DISTINCT], s_acc[1]);
            atomicAdd(&a.counters[CC_H2_ONLY_TOTAL], s_acc[2]); atomicAdd(&a.counters[CC_H2_ONLY_DISTINCT], s_acc[3]);
        }
    }
    for (uint32_t i = threadIdx.x; TILE && i < COMP_TILE * COMP_TILE; i += blockDim.x) {
        uint32_t v = s_tile[i];
        if (!v) continue;
        uint32_t r = i / COMP_TILE, c = i % COMP_TILE;
        if (r < a.d1_bins && c < a.d2_bins) atomicAdd(&a.main_mx[(uint64_t)r * a.d2_bins + c], (unsigned long long)v);
        // the tile's k-mers of pass 1 (row >= 1: a k-mer of hash 1 counts at least 1; row 0 is where pass 2 puts what only hash 2
        // has): spectrum1[ca], and when shared, shared_spectrum1[ca] / shared_spectrum2[cb]
        if (PASS == 1 && a.fold && r) {
            atomicAdd(&s_spec[r], v);
            if (c) { atomicAdd(&s_spec[a.spec_size + r], v); atomicAdd(&s_spec[2 * a.spec_size + c], v); }
        }
    }
    if (PASS == 1 && a.fold) __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_spec; i += blockDim.x) {
        uint32_t v = s_spec[i];
        if (!v) continue;
        uint32_t which = i / a.spec_size, bin = i % a.spec_size;
        uint32_t dst = PASS == 1 ? (which == 0 ? 0u : which + 1) : 1u;       // spectra order: s1, s2, shared1, shared2
        atomicAdd(&a.spectra[(uint64_t)dst * a.spec_size + bin], (unsigned long long)v);
    }
}

// LDS carve of both comp kernels: [0,16) u64 scalar accumulators | tile 64x64 u32 | spectra (3 or 1) u32 | (join: region)
__device__ __forceinline__ void comp_lds_init(const CompArgs& a, int pass, unsigned long long* s_acc, uint32_t* s_tile, uint32_t* s_spec) {
    const uint32_t n_spec = (pass == 1 ? 3u : 1u) * a.spec_size;
    for (uint32_t i = threadIdx.x; i < 16; i += blockDim.x) s_acc[i] = 0;
    for (uint32_t i = threadIdx.x; i < COMP_TILE * COMP_TILE; i += blockDim.x) s_tile[i] = 0;
    for (uint32_t i = threadIdx.x; i < n_spec; i += blockDim.x) s_spec[i] = 0;
    __syncthreads();
}

// K5 (probe form): scan one table, probe the other in HBM.  Used when the two tables' region grids differ or when the
// probe key is not the stored key (mixed canonical flags).
template <int PASS, bool W>
__global__ void __launch_bounds__(256)
k_comp(DevTable ta, uint32_t na_ovf, DevTable tb, uint32_t nb_ovf, CompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_raw);
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(s_raw + 16 * sizeof(unsigned long long));
    uint32_t* s_spec = s_tile + COMP_TILE * COMP_TILE;
    comp_lds_init(a, PASS, s_acc, s_tile, s_spec);
    const uint32_t k = ta.k;
    CompAcc acc;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_slots = ta.cap + 1;                   // virtual slot cap == the all-ones key
    const uint64_t rounds = (n_slots + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        uint64_t i = first + r * stride;
        uint64_t key = EMPTY, ca = 0, cb = 0;
        bool occ = false;
        if (i < ta.cap) {
            if constexpr (W) { key = ta.keys[i]; occ = key != EMPTY; if (occ) ca = slot_count(ta, i, key, na_ovf); }
            else { const SlotView v = slot_view(ta, i); key = v.key; occ = v.occ; if (occ) ca = slot_total(ta, i, key, v.cnt, na_ovf); }
        } else if (i == ta.cap) {
            ca = ta.ctrs[CTR_ONES];
            occ = ca != 0;
        }
        if (occ) {
            // pass 1: hash-1 key probed in hash 2, canonicalised iff input 2 is canonical (src/comp.cc:401)
            // pass 2: hash-2 key probed in hash 1, ALWAYS canonicalised (src/comp.cc:447 passes a pointer as the bool)
            if constexpr (W) {
                KeyW kw{key, ta.keys_b[i]};
                if (PASS == 2 || a.canon_probe) kw = keyw_canonical(kw, k);
                cb = table_get_w(tb, kw, nb_ovf);
            } else {
                uint64_t probe = (PASS == 2 || a.canon_probe) ? kmer_canonical(key, k) : key;
                cb = table_get(tb, probe, nb_ovf);
            }
        }
        comp_account<PASS>(occ, ca, cb, a, s_tile, s_spec, acc);
    }
    comp_flush<PASS>(a, s_acc, s_tile, s_spec, acc);
}

// K5 (join form): when both tables share the region grid (p1 x p2) and the probe key IS the stored key, region r of
// one table can only match region r of the other -- a partitioned hash join.  A persistent workgroup walks regions:
// the probed table's region goes into LDS (coalesced), the scanned table's region streams past it, every probe is an
// LDS probe.  HBM sees each table exactly once per pass, as a stream.
// PK: both tables packed (the same grid gives the same remainder bits, hence the same layout): 8 bytes per slot on either side,
// and what is compared is the slot's remainder -- tables of one grid give one k-mer one remainder -- so no k-mer is ever decoded.
template <int PASS, bool PK>
__global__ void __launch_bounds__(JOIN_BLOCK)
k_comp_join(DevTable ta, uint32_t na_ovf, DevTable tb, uint32_t nb_ovf, CompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_raw);
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(s_raw + 16 * sizeof(unsigned long long));
    uint32_t* s_spec = s_tile + COMP_TILE * COMP_TILE;
    const uint32_t n_spec = (PASS == 1 ? 3u : 1u) * a.spec_size;
    const uint32_t Sa = ta.region_slots, Sb = tb.region_slots;
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(s_raw + ((16 * 8 + COMP_TILE * COMP_TILE * 4 + n_spec * 4 + 15) & ~15u));
    uint32_t* rc = reinterpret_cast<uint32_t*>(rk + Sb);                     // (KV12 only)
    uint32_t* s_seen = PK ? reinterpret_cast<uint32_t*>(rk + Sb) : rc + Sb;  // (a.seen) which slots of this region of hash 2 were found
    const bool mark = PASS == 1 && a.seen != nullptr;
    const uint32_t cb_bits = tb.cbits;                                       // == ta.cbits when PK
    comp_lds_init(a, PASS, s_acc, s_tile, s_spec);
    CompAcc acc;
    const uint32_t R = ta.n_regions;
    for (uint32_t r = blockIdx.x; r < R; r += gridDim.x) {
        __syncthreads();
        const uint64_t bbase = (uint64_t)r * Sb, abase = (uint64_t)r * Sa;
        const RegionPlace rpb = region_place(tb, r);
        if constexpr (PK) {                                                  // 16 bytes per lane and load (Sb is a multiple of 4)
            for (uint32_t i = threadIdx.x * 2; i < Sb; i += blockDim.x * 2)
                *reinterpret_cast<u64x2*>(rk + i) = *reinterpret_cast<const u64x2*>(tb.keys + bbase + i);
        } else {
            for (uint32_t i = threadIdx.x; i < Sb; i += blockDim.x) { rk[i] = tb.keys[bbase + i]; rc[i] = tb.counts[bbase + i]; }
        }
        if (mark) for (uint32_t i = threadIdx.x; i < a.seen_wpr; i += blockDim.x) s_seen[i] = 0;
        __syncthreads();
        constexpr int JB = 4;                                               // slots per lane in flight
        for (uint32_t i0 = 0; i0 < Sa; i0 += JB * blockDim.x) {             // uniform trip count: ballots inside comp_account
            uint64_t keys[JB]; uint32_t cnts[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const uint32_t i = i0 + u * blockDim.x + threadIdx.x;
                const uint32_t ic = i < Sa ? i : Sa - 1;                    // clamped: the loads stay in one basic block
                keys[u] = ta.keys[abase + ic];
                if constexpr (!PK) cnts[u] = ta.counts[abase + ic]; else cnts[u] = 0;
                if (i >= Sa) keys[u] = PK ? 0ULL : EMPTY;
            }
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const uint64_t key = keys[u];
                const bool occ = PK ? key != 0 : key != EMPTY;
                uint64_t ca = 0, cb = 0;
                if (occ) {
                    const uint32_t i = i0 + u * blockDim.x + threadIdx.x;
                    if constexpr (PK) {
                        ca = pk_count(key, cb_bits);
                        if (na_ovf) ca += ovf_get(ta, abase + i);
                        const uint64_t rem = pk_rem(key, cb_bits);
                        uint32_t s = place_offset(rem, rpb.pl, Sb);
                        for (uint32_t probe = 0; probe < Sb; ++probe) {
                            const unsigned long long cur = rk[s];
                            if (cur == 0) break;
                            if ((cur >> cb_bits) == rem) { cb = pk_count(cur, cb_bits); if (nb_ovf) cb += ovf_get(tb, bbase + s); if (mark) atomicOr(&s_seen[s >> 5], 1u << (s & 31)); break; }
                            s = s + 1 == Sb ? 0 : s + 1;
                        }
                    } else {
                        ca = cnts[u];
                        if (na_ovf) ca += ovf_get(ta, key);
                        uint32_t s = home_offset_in(key, rpb);
                        for (uint32_t probe = 0; probe < Sb; ++probe) {
                            const unsigned long long cur = rk[s];
                            if (cur == key) { cb = rc[s]; if (nb_ovf) cb += ovf_get(tb, key); if (mark) atomicOr(&s_seen[s >> 5], 1u << (s & 31)); break; }
                            if (cur == EMPTY) break;
                            s = s + 1 == Sb ? 0 : s + 1;
                        }
                    }
                }
                comp_account<PASS>(occ, ca, cb, a, s_tile, s_spec, acc);
            }
        }
        if (mark) {                                                          // pass 2 will read this instead of probing hash 1
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < a.seen_wpr; i += blockDim.x) a.seen[(uint64_t)r * a.seen_wpr + i] = s_seen[i];
        }
    }
    {   // the all-ones key lives outside the slots: one lane of block 0 takes it through the HBM path (KV12 tables of k = 32 only)
        uint64_t ca = 0, cb = 0;
        bool occ = false;
        if (!PK && blockIdx.x == 0 && threadIdx.x == 0) {
            ca = ta.ctrs[CTR_ONES];
            occ = ca != 0;
            if (occ) cb = table_get(tb, (PASS == 2 || a.canon_probe) ? kmer_canonical(EMPTY, ta.k) : EMPTY, nb_ovf);
        }
        comp_account<PASS>(occ, ca, cb, a, s_tile, s_spec, acc);
    }
    comp_flush<PASS>(a, s_acc, s_tile, s_spec, acc);
}

// Pass 2 after a join pass 1 that marked what it found (CompArgs::seen): the k-mers of hash 2 that hash 1 does not hold are the
// occupied slots whose bit is clear, so pass 2 is one scan of hash 2 and of a bit per slot -- no probe of hash 1 at all (the probe
// form spends 1.3 random sector reads per k-mer of hash 2: 25 ms at config 4, 49 ms when hash 2 is a second read library).  Both
// tables canonical (pass 2 probes the canonical form, src/comp.cc:447: only then is "found by pass 1" the same question), and the
// all-ones key is never canonical, so there is no slot-less key to look after.
template <bool PK>
__global__ void __launch_bounds__(512)
k_comp_seen(DevTable ta /* hash 2 */, uint32_t na_ovf, CompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_raw);
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(s_raw + 16 * sizeof(unsigned long long));
    uint32_t* s_spec = s_tile + COMP_TILE * COMP_TILE;
    comp_lds_init(a, 2, s_acc, s_tile, s_spec);
    CompAcc acc;
    const uint32_t S = ta.region_slots, R = ta.n_regions;
    constexpr int JB = 4;
    for (uint32_t r = blockIdx.x; r < R; r += gridDim.x) {
        const uint64_t base = (uint64_t)r * S;
        const uint32_t* seen = a.seen + (uint64_t)r * a.seen_wpr;
        for (uint32_t i0 = 0; i0 < S; i0 += JB * blockDim.x) {               // uniform trip count: ballots inside comp_account
            uint64_t keys[JB]; uint32_t cnts[JB], bits[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const uint32_t i = i0 + u * blockDim.x + threadIdx.x;
                const uint32_t ic = i < S ? i : S - 1;                      // clamped: the loads stay in one basic block
                keys[u] = ta.keys[base + ic]; bits[u] = seen[ic >> 5] >> (ic & 31);
                if constexpr (!PK) cnts[u] = ta.counts[base + ic]; else cnts[u] = 0;
                if (i >= S) keys[u] = PK ? 0ULL : EMPTY;
            }
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const bool occ = PK ? keys[u] != 0 : keys[u] != EMPTY;
                uint64_t ca = 0;
                if (occ) {
                    const uint32_t i = i0 + u * blockDim.x + threadIdx.x;
                    ca = PK ? pk_count(keys[u], ta.cbits) : (uint64_t)cnts[u];
                    if (na_ovf) ca += ovf_get(ta, PK ? base + i : keys[u]);
                }
                comp_account<2>(occ, ca, (uint64_t)(bits[u] & 1u), a, s_tile, s_spec, acc);
            }
        }
    }
    comp_flush<2>(a, s_acc, s_tile, s_spec, acc);
}

// K5 (fused join): both passes of Comp::compare in one sweep of the two tables, for packed tables of one grid whose k-mers are
// stored canonical (then "the probe key is the stored key" holds for pass 1 AND pass 2).  Region r of the RESIDENT table sits in
// LDS, region r of the STREAMED table flows past it and probes it; every streamed k-mer marks the resident slot it finds, and a
// sweep of the resident region in LDS afterwards accounts for the slots nobody marked.  The host streams the table with FEWER
// k-mers: the probes are then mostly successful ones (1.75 slots at load 0.6), where k_comp_join<1> -- hash 2 resident, hash 1
// streamed whatever the sizes -- spent 3.6 slots and a wave-long wait on each of the 2 G error k-mers of a read set that an
// assembly does not hold.
//   SWAP = false: hash 1 resident, hash 2 streamed.  A streamed k-mer gives pass 2 its item (count in 2, found or not) and, when
//                 found, pass 1 the pair (count in 1, count in 2); the sweep gives pass 1 the k-mers only hash 1 has.
//   SWAP = true:  hash 2 resident, hash 1 streamed.  A streamed k-mer is pass 1's item (count in 1, count in 2 or 0); the sweep is
//                 pass 2: every k-mer of hash 2, found or not.
// LDS: 16 u64 accumulators (0-6 pass 1, 8-11 pass 2) | tile 64 x 64 u32 | spectra 3 x ss (pass 1) | spectrum ss (pass 2) | the
// resident region's S words + FUSED_STEP | per wave, FUSED_QCAP queue entries (8 + 4 bytes) | S / 32 mark words.
constexpr int FUSED_BLOCK = 1024, FUSED_KP = 5;               // 5 x 1024 x 2 slots: regions of up to 10240 slots (resident and streamed)
constexpr int FUSED_STEP = 4;                                 // slots per step of the walk through the resident region (two ds_read2_b64)
constexpr int FUSED_QCAP = 192;                               // entries of a wave's queue: fewer than 64 left over + the 128 slots of a pair
typedef uint64_t u64x2a8 __attribute__((ext_vector_type(2), aligned(8)));
// JP: 16-byte pairs of streamed slots per lane -- the WHOLE streamed region sits in registers, loaded one region ahead: pair u of the
// next region is requested the moment pair u of this one has been taken out of its registers, so every HBM request has a region's
// worth of work (the other pairs, the sweep, the next store) to land behind.  (Round 4's form loaded 4 slots per lane, waited the full
// HBM latency and worked them off, two to three times per region: half of every wave's cycles were that wait.)  Wave-items past the
// region's end are skipped by the wave, not masked: a region of 6256 slots is 3.05 pairs per lane, not 4.
// OVF: some count of either table continues in its side table (katgpu_table::n_ovf); without, a count is the low cbits <= 32 bits of its
// slot and the whole account runs on 32-bit counts (the kernel is bound by VALU issue: ~230 instructions per streamed + swept slot).
template <bool SWAP, int JP, bool OVF>
__global__ void __launch_bounds__(FUSED_BLOCK)
k_comp_fused(DevTable t1, uint32_t n1_ovf, DevTable t2, uint32_t n2_ovf, CompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_raw);
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(s_raw + 16 * sizeof(unsigned long long));
    uint32_t* s_spec1 = s_tile + COMP_TILE * COMP_TILE;
    uint32_t* s_spec2 = s_spec1 + 3 * a.spec_size;
    const DevTable& tr = SWAP ? t2 : t1;                      // resident
    const DevTable& ts = SWAP ? t1 : t2;                      // streamed
    const uint32_t nr_ovf = OVF ? (SWAP ? n2_ovf : n1_ovf) : 0u, ns_ovf = OVF ? (SWAP ? n1_ovf : n2_ovf) : 0u;
    typedef typename std::conditional<OVF, uint64_t, uint32_t>::type CT;                // a count
    const uint32_t Sr = tr.region_slots, Ss = ts.region_slots, cb = tr.cbits;          // one grid: one remainder width, one cbits
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(s_raw + ((16 * 8 + COMP_TILE * COMP_TILE * 4 + 4 * a.spec_size * 4 + 15) & ~15u));
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    unsigned long long* q_w = rk + Sr + FUSED_STEP + (tid >> 6) * FUSED_QCAP;          // (rk[Sr ..]: the region's first slots again) this wave's queue: slot words
    uint32_t* q_i = reinterpret_cast<uint32_t*>(rk + Sr + FUSED_STEP + (FUSED_BLOCK / 64) * FUSED_QCAP) + (tid >> 6) * FUSED_QCAP;   // ... and where they were
    uint32_t* s_mark = reinterpret_cast<uint32_t*>(rk + Sr + FUSED_STEP + (FUSED_BLOCK / 64) * FUSED_QCAP) + (FUSED_BLOCK / 64) * FUSED_QCAP;
    const uint32_t mark_words = (Sr + 31) / 32;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(tid & ~63u);                 // this wave's first lane, as a scalar
    comp_lds_init(a, 1, s_acc, s_tile, s_spec1);
    for (uint32_t i = tid; i < a.spec_size; i += blockDim.x) s_spec2[i] = 0;
    CompAcc acc1, acc2;
    // what the streamed k-mers and the swept ones mostly are: count 1, absent from the other table (CompHot)
    CompHot hot_s, hot_r;
    hot_s.tile_cell = SWAP ? comp_tile_cell<1>(1, 0, a) : comp_tile_cell<2>(1, 0, a);
    hot_r.tile_cell = SWAP ? comp_tile_cell<2>(1, 0, a) : comp_tile_cell<1>(1, 0, a);
    hot_s.spec_bin = hot_r.spec_bin = spectrum_bin(1u, a.spec_size);
    const Place pl = place_make(tr.k, tr.p1, tr.n1, tr.l2);
    const uint32_t R = tr.n_regions;
    u32x4s kq[FUSED_KP], wq[JP];
    auto prefetch = [&](uint32_t r) {                          // the resident region, 16 bytes per lane and load, clamped and unconditional
        const uint64_t base = (uint64_t)r * Sr;
#pragma unroll
        for (int u = 0; u < FUSED_KP; ++u) { const uint32_t i = (u * FUSED_BLOCK + tid) * 2; kq[u] = *reinterpret_cast<const u32x4s*>(tr.keys + base + (i < Sr ? i : 0)); }
    };
    auto stream_pair = [&](uint32_t r, int u) {                // pair u of the streamed region (both region sizes are multiples of 4)
        const uint32_t i = (u * FUSED_BLOCK + tid) * 2;
        return *reinterpret_cast<const u32x4s*>(ts.keys + (uint64_t)r * Ss + (i < Ss ? i : 0));
    };
    if (blockIdx.x < R) {
        prefetch(blockIdx.x);
#pragma unroll
        for (int u = 0; u < JP; ++u) wq[u] = stream_pair(blockIdx.x, u);
    }
    for (uint32_t r = blockIdx.x; r < R; r += gridDim.x) {
        __syncthreads();                                       // the previous region's sweep is through
#pragma unroll
        for (int u = 0; u < FUSED_KP; ++u) { const uint32_t i = (u * FUSED_BLOCK + tid) * 2; if (i < Sr) *reinterpret_cast<u32x4s*>(rk + i) = kq[u]; }
        if (tid * 2 < FUSED_STEP) *reinterpret_cast<u32x4s*>(rk + Sr + tid * 2) = kq[0];         // slots 0 .. FUSED_STEP - 1 once more, behind the last
        for (uint32_t i = tid; i < mark_words; i += blockDim.x) s_mark[i] = 0;
        __syncthreads();
        const uint32_t rn = r + gridDim.x < R ? r + gridDim.x : r;   // (the last region asks for itself again: the loads stay unconditional)
        prefetch(rn);                                          // the next one: in flight behind this region's work
        const uint64_t rbase = (uint64_t)r * Sr, sbase = (uint64_t)r * Ss;
        // ---- the streamed region: its occupied slots, 64 at a time ----
        // One streamed k-mer with its walk and its two accounts is ~170 instructions, and the kernel is bound by their issue.  A wave-item
        // of 64 consecutive slots holds as many k-mers as the table's load says: 19 of an assembly's (load 0.3), 40 of a read set's -- the
        // other lanes would sit through every instruction.  So the slots go through the wave's queue first (a ballot, a bit count and one
        // store each) and the walk takes them from there, every lane busy.
        uint32_t q_n = 0;                                      // entries in the queue (uniform in the wave)
        auto stream_drain = [&](uint32_t need) {
#pragma unroll 1
            while (q_n >= need) {
                const uint32_t cnt = q_n < 64 ? q_n : 64;
                q_n -= cnt;
                const bool occ = lane < cnt;
                const uint64_t w = occ ? q_w[q_n + lane] : 0ULL;
                CT cs = 0, cr = 0;                                           // count in the streamed table, in the resident one
                if (occ) {
                    cs = (CT)pk_count(w, cb);
                    if (OVF && ns_ovf) cs += (CT)ovf_get(ts, sbase + q_i[q_n + lane]);
                    const uint64_t rem = pk_rem(w, cb);
                    uint32_t s = place_offset(rem, pl, Sr);
                    // the walk, FUSED_STEP slots per LDS round trip (the region is followed by a copy of its first slots, so a step never
                    // wraps): what bounds a wave is its longest walk times the LDS latency -- 3.6 slots on average for a k-mer the resident
                    // table does not hold, some thirty for the unluckiest of 64 lanes
                    for (uint32_t probe = 0; probe < Sr; probe += FUSED_STEP) {
                        const u64x2a8 c01 = *reinterpret_cast<const u64x2a8*>(rk + s), c23 = *reinterpret_cast<const u64x2a8*>(rk + s + 2);
                        const unsigned long long c[FUSED_STEP] = {c01.x, c01.y, c23.x, c23.y};
                        int hit = -1; bool stop = false;                      // the first slot that is empty (stop) or holds the k-mer (hit)
#pragma unroll
                        for (int j = FUSED_STEP - 1; j >= 0; --j) {
                            const bool e = c[j] == 0, m = (c[j] >> cb) == rem;
                            if (e) { stop = true; hit = -1; } else if (m) { stop = true; hit = j; }
                        }
                        if (hit >= 0) {
                            const unsigned long long cur = hit == 0 ? c[0] : hit == 1 ? c[1] : hit == 2 ? c[2] : c[3];
                            uint32_t at = s + (uint32_t)hit; at = at >= Sr ? at - Sr : at;
                            cr = (CT)pk_count(cur, cb);
                            if (OVF && nr_ovf) cr += (CT)ovf_get(tr, rbase + at);
                            atomicOr(&s_mark[at >> 5], 1u << (at & 31));
                        }
                        if (stop) break;
                        s += FUSED_STEP; s = s >= Sr ? s - Sr : s;
                    }
                }
                if (SWAP) comp_account<1, true>(occ, cs, cr, a, s_tile, s_spec1, acc1, &hot_s);         // streamed = hash 1
                else {
                    comp_account<1>(occ && cr != 0, cr, cs, a, s_tile, s_spec1, acc1);                   // the pair, seen from hash 1
                    comp_account<2, true>(occ, cs, cr, a, s_tile, s_spec2, acc2, &hot_s);                // streamed = hash 2
                }
            }
        };
#pragma unroll
        for (int u = 0; u < JP; ++u) {
            const u32x4s x = wq[u];
            wq[u] = stream_pair(rn, u);
            if ((u * FUSED_BLOCK + wave0) * 2 < Ss) {                       // (else: nothing of the region left for this wave)
                const uint32_t i = (u * FUSED_BLOCK + tid) * 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint64_t w = i < Ss ? ((uint64_t)(h ? x.w : x.y) << 32) | (h ? x.z : x.x) : 0ULL;
                    const unsigned long long m = __ballot(w != 0);
                    if (m) {
                        const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                        if (w != 0) { q_w[at] = w; if (OVF) q_i[at] = i + h; }
                        q_n += (uint32_t)__popcll(m);
                    }
                }
            }
            stream_drain(u == JP - 1 ? 1u : 64u);                           // (the last pair leaves nothing behind)
        }
        __syncthreads();                                       // every mark is in
        // ---- the resident region: what of it the accounts want (hash 1 resident: the k-mers nobody found; hash 2: all), the same way ----
#pragma unroll 1
        for (uint32_t i0 = 0; ; i0 += blockDim.x) {
            const bool more = i0 + wave0 < Sr;                 // (uniform in the wave)
            if (more) {
                const uint32_t i = i0 + tid;
                const unsigned long long cur = i < Sr ? rk[i] : 0ULL;
                const bool marked = i < Sr && ((s_mark[i >> 5] >> (i & 31)) & 1u);
                const bool want = cur != 0 && (SWAP || !marked);
                const unsigned long long m = __ballot(want);
                if (m) {
                    const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (want) { q_w[at] = cur; if (OVF || SWAP) q_i[at] = i | (marked ? 0x80000000u : 0u); }
                    q_n += (uint32_t)__popcll(m);
                }
            }
#pragma unroll 1
            while (q_n >= (more ? 64u : 1u)) {
                const uint32_t cnt = q_n < 64 ? q_n : 64;
                q_n -= cnt;
                const bool occ = lane < cnt;
                const unsigned long long cur = occ ? q_w[q_n + lane] : 0ULL;
                const uint32_t qi = (OVF || SWAP) && occ ? q_i[q_n + lane] : 0u;
                CT cr = 0;
                if (occ) { cr = (CT)pk_count(cur, cb); if (OVF && nr_ovf) cr += (CT)ovf_get(tr, rbase + (qi & 0x7FFFFFFFu)); }
                if (SWAP) comp_account<2, true>(occ, cr, (CT)(qi >> 31), a, s_tile, s_spec2, acc2, &hot_r);            // resident = hash 2: all of it
                else comp_account<1, true>(occ, cr, (CT)0, a, s_tile, s_spec1, acc1, &hot_r);                          // resident = hash 1: what hash 2 lacks
            }
            if (!more) break;
        }
    }
    comp_hot_flush(hot_s, s_tile, SWAP ? s_spec1 : s_spec2);
    comp_hot_flush(hot_r, s_tile, SWAP ? s_spec2 : s_spec1);
    __syncthreads();
    comp_flush<1>(a, s_acc, s_tile, s_spec1, acc1);
    __syncthreads();
    comp_flush<2, false>(a, s_acc + 8, s_tile, s_spec2, acc2);
}

// ---- K5b: the third comp input (src/comp.cc:123-127,403-433,466-479) ----
// Pass over hash 1 again with both other tables probed: the scaled counts decide which of the ends / middle / mixed
// matrices the k-mer lands in.  Three 64x64 LDS tiles (the hot low-count corner), global atomics elsewhere.
struct Comp3Args {
    double d1_scale, d2_scale;
    uint32_t d1_bins, d2_bins;
    uint32_t canon2, canon3;         // probes canonicalised iff that input is canonical (src/comp.cc:401,404)
    unsigned long long* mx[3];       // 0 ends, 1 middle, 2 mixed: each d1_bins x d2_bins
};

template <bool W>
__global__ void __launch_bounds__(256)
k_comp3_pass1(DevTable t1, uint32_t n1_ovf, DevTable t2, uint32_t n2_ovf, DevTable t3, uint32_t n3_ovf, Comp3Args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_tiles[];      // 3 x 64 x 64
    for (uint32_t i = threadIdx.x; i < 3 * COMP_TILE * COMP_TILE; i += blockDim.x) s_tiles[i] = 0;
    __syncthreads();
    const uint32_t k = t1.k;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rounds = (t1.cap + 1 + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = first + r * stride;
        uint64_t key = EMPTY, c1 = 0;
        bool occ = false;
        if (i < t1.cap) {
            if constexpr (W) { key = t1.keys[i]; occ = key != EMPTY; if (occ) c1 = slot_count(t1, i, key, n1_ovf); }
            else { const SlotView v = slot_view(t1, i); key = v.key; occ = v.occ; if (occ) c1 = slot_total(t1, i, key, v.cnt, n1_ovf); }
        } else if (i == t1.cap) { c1 = t1.ctrs[CTR_ONES]; occ = c1 != 0; }
        uint32_t which = 0, cell = 0;
        bool in_tile = false;
        if (occ) {
            uint64_t c2, c3;
            if constexpr (W) {
                const KeyW kw{key, t1.keys_b[i]}, can = keyw_canonical(kw, k);
                c2 = table_get_w(t2, a.canon2 ? can : kw, n2_ovf);
                c3 = table_get_w(t3, a.canon3 ? can : kw, n3_ovf);
            } else {
                const uint64_t can = kmer_canonical(key, k);
                c2 = table_get(t2, a.canon2 ? can : key, n2_ovf);
                c3 = table_get(t3, a.canon3 ? can : key, n3_ovf);
            }
            uint64_t s1 = scale_count(c1, a.d1_scale), s2 = scale_count(c2, a.d2_scale), s3 = scale_count(c3, a.d2_scale);
            if (s1 >= a.d1_bins) s1 = a.d1_bins - 1;
            if (s2 >= a.d2_bins) s2 = a.d2_bins - 1;
            if (s3 >= a.d2_bins) s3 = a.d2_bins - 1;
            which = s2 == s3 ? 0u : (s3 > 0 ? 2u : 1u);                     // ends / mixed / middle (src/comp.cc:426-432)
            in_tile = s1 < COMP_TILE && s3 < COMP_TILE;
            cell = in_tile ? which * COMP_TILE * COMP_TILE + (uint32_t)(s1 * COMP_TILE + s3) : (uint32_t)(s1 * a.d2_bins + s3);
        }
        lds_inc_aggregated(s_tiles, cell, occ && in_tile);
        if (occ && !in_tile) atomicAdd(&a.mx[which][cell], 1ULL);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3 * COMP_TILE * COMP_TILE; i += blockDim.x) {
        const uint32_t v = s_tiles[i];
        if (!v) continue;
        const uint32_t which = i / (COMP_TILE * COMP_TILE), rc = i % (COMP_TILE * COMP_TILE), rr = rc / COMP_TILE, cc = rc % COMP_TILE;
        if (rr < a.d1_bins && cc < a.d2_bins) atomicAdd(&a.mx[which][(uint64_t)rr * a.d2_bins + cc], (unsigned long long)v);
    }
}

// updateHash3Counters (lib/src/comp_counters.cc:113-117): hash3_total, hash3_distinct
static __global__ void __launch_bounds__(256)
k_comp3_pass3(DevTable t3, uint32_t n3_ovf, unsigned long long* counters) {
    uint64_t tot = 0, dis = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t3.cap; i += stride) {
        if (i < t3.cap) {
            const uint64_t w = t3.keys[i];
            if (t3.cbits ? w != 0 : w != EMPTY) { tot += slot_total(t3, i, w, t3.cbits ? pk_count(w, t3.cbits) : (uint64_t)t3.counts[i], n3_ovf); ++dis; }
        }
        else { const uint64_t c = t3.ctrs[CTR_ONES]; if (c) { tot += c; ++dis; } }
    }
    for (int off = 32; off > 0; off >>= 1) { tot += __shfl_down(tot, off, 64); dis += __shfl_down(dis, off, 64); }
    if ((threadIdx.x & 63) == 0 && dis) { atomicAdd(&counters[CC_H3_TOTAL], (unsigned long long)tot); atomicAdd(&counters[CC_H3_DISTINCT], (unsigned long long)dis); }
}

// ---- batch lookup (JellyfishHelper::getCount, lib/src/jellyfish_helper.cc:189-194) ----
static __global__ void __launch_bounds__(256)
k_get(DevTable t, uint32_t n_ovf, const uint64_t* __restrict__ keys, uint64_t n, int canonicalise, uint64_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table_get(t, canonicalise ? kmer_canonical(keys[i], t.k) : keys[i], n_ovf);
}

// K8.  Per-position coverage of a sequence (kat sect / kat cold; src/sect.cc:516-535): out[i] = count of the k-window
// starting at base i, 0 when the window holds anything but ACGTacgt.  Same front end as k_count (16-byte loads, packed
// codes through LDS, a 96-bit register window slid 16 times); the back end is a read-only probe, so the table's
// cache lines are shared between waves and nothing is atomic.  Each lane produces 16 consecutive counts = one 128-byte
// line of `out`, written as 8 dwordx4 stores.
template <bool ALIGNED>
__global__ void __launch_bounds__(COUNT_BLOCK)
k_profile(DevTable t, uint32_t n_ovf, int canonicalise, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_chunks,
          uint64_t* __restrict__ out) {
    __shared__ uint32_t s_code[COUNT_BLOCK + 2];
    __shared__ uint32_t s_bad[COUNT_BLOCK + 2];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = t.k;
    const uint64_t n_out = n - k + 1;
    if (tid < 2) { s_code[COUNT_BLOCK + tid] = 0; s_bad[COUNT_BLOCK + tid] = 0xFFFF; }

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t off = chunk * CHUNK_STARTS + (uint64_t)tid * BASES_PER_LANE;
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
                    uint64_t i = off + q * 4 + b;
                    uint32_t c = i < n ? bases[i] : (uint32_t)'N';
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

        if (tid < LANES_WITH_STARTS && off < n_out) {
            uint64_t hi = ((uint64_t)s_code[tid] << 32) | s_code[tid + 1];
            uint64_t lo = (uint64_t)s_code[tid + 2] << 32;
            uint64_t m = ((uint64_t)s_bad[tid] << 48) | ((uint64_t)s_bad[tid + 1] << 32) | ((uint64_t)s_bad[tid + 2] << 16);
            const uint32_t kshift = 64 - 2 * k, mshift = 64 - k;
            uint64_t c[BASES_PER_LANE];
#pragma unroll
            for (int j = 0; j < BASES_PER_LANE; ++j) {
                c[j] = 0;
                if ((m >> mshift) == 0) {
                    uint64_t key = hi >> kshift;
                    if (canonicalise) key = kmer_canonical(key, k);
                    c[j] = table_get(t, key, n_ovf);
                }
                hi = (hi << 2) | (lo >> 62);
                lo <<= 2;
                m <<= 1;
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

// ---- export / owner partition ----
// mode 0: count records per part into sizes[]; mode 1: scatter records to cursors[part]++.
template <int MODE>
__global__ void __launch_bounds__(256)
k_partition(DevTable t, uint32_t n_ovf, uint32_t n_parts, unsigned long long* __restrict__ sizes_or_cursors,
            uint64_t* __restrict__ out_keys, uint64_t* __restrict__ out_counts) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t.cap; i += stride) {
        uint64_t key = EMPTY, c = 0;
        if (i < t.cap) { const SlotView v = slot_view(t, i); if (!v.occ) continue; key = v.key; c = slot_total(t, i, key, v.cnt, n_ovf); }
        else { c = t.ctrs[CTR_ONES]; if (!c) continue; }
        uint32_t part = n_parts > 1 ? owner_of(key, t.k, n_parts) : 0;
        unsigned long long at = atomicAdd(&sizes_or_cursors[part], 1ULL);
        if (MODE == 1) { out_keys[at] = key; out_counts[at] = c; }
    }
}

// ---- region-ordered extraction and merge: the multi-GPU exchange (kat_amd/dist.py) ----
// Every rank counts into a table of the same region grid, so the records of one region of one rank belong to the same
// region of the owner's table.  Extraction walks the table region by region and writes each owner's records in region
// order; the owner then applies, per region, the runs it received from every rank to that region in LDS -- no
// global atomic per record and no re-partitioning on the receiving side.
constexpr int EXTRACT_BLOCK = 256;
constexpr int EXTRACT_UNROLL = 4;            // 16-byte loads a lane of the extraction kernels has in flight (packed tables)
constexpr uint32_t MAX_EXCHANGE_PARTS = 256;
constexpr int MAX_MERGE_SRC = 16;
constexpr int MERGE_BATCH = 8;
constexpr int MERGE_BLOCK = 512;             // threads of a k_merge_apply workgroup               // records a thread of k_merge_apply fetches before it applies the first

// pass 1: rcnt[p * R + g] = number of records of region g owned by part p.  One workgroup per region.
static __global__ void __launch_bounds__(EXTRACT_BLOCK)
k_extract_count(DevTable t, uint32_t n_parts, uint32_t* __restrict__ rcnt) {
    __shared__ uint32_t s_cnt[MAX_EXCHANGE_PARTS];
    const uint32_t tid = threadIdx.x, S = t.region_slots;
    for (uint32_t g = blockIdx.x; g < t.n_regions; g += gridDim.x) {
        for (uint32_t p = tid; p < n_parts; p += EXTRACT_BLOCK) s_cnt[p] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)g * S;
        const RegionPlace rp = region_place(t, g);
        if (t.cbits) {
            // packed slots: four 16-byte loads per lane in flight before the first is looked at (the kernel waits for memory 0.63 of its
            // wave cycles with one 8-byte load per lane: profiles/r06_exchange_components.txt) -- regions are multiples of four slots
            const ulonglong2* w2 = reinterpret_cast<const ulonglong2*>(t.keys + base);
            const uint32_t n2 = S >> 1;
            for (uint32_t i0 = 0; i0 < n2; i0 += EXTRACT_BLOCK * EXTRACT_UNROLL) {
                ulonglong2 w[EXTRACT_UNROLL];
#pragma unroll
                for (int u = 0; u < EXTRACT_UNROLL; ++u) { const uint32_t i = i0 + u * EXTRACT_BLOCK + tid; w[u] = i < n2 ? w2[i] : make_ulonglong2(0, 0); }
#pragma unroll
                for (int u = 0; u < EXTRACT_UNROLL; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint64_t x = h ? w[u].y : w[u].x;
                        if (x) atomicAdd(&s_cnt[n_parts > 1 ? owner_of(key_in(pk_rem(x, t.cbits), rp), t.k, n_parts) : 0], 1u);
                    }
            }
        } else {
            for (uint32_t i = tid; i < S; i += EXTRACT_BLOCK) {
                const SlotView v = slot_view_in(t, rp, base + i);
                if (v.occ) atomicAdd(&s_cnt[n_parts > 1 ? owner_of(v.key, t.k, n_parts) : 0], 1u);
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < n_parts; p += EXTRACT_BLOCK) rcnt[(uint64_t)p * t.n_regions + g] = s_cnt[p];
        __syncthreads();
    }
}

// Exclusive scan of each row of a u32 matrix into u64 (row_base[r] added when given); one workgroup per row.
// off may be null (totals only); off rows have n_cols + 1 entries when `closed` (the last one is the row total).
static __global__ void __launch_bounds__(1024)
k_rows_scan(const uint32_t* __restrict__ m, uint32_t n_cols, uint64_t row_stride, const uint64_t* __restrict__ row_base,
            uint64_t* __restrict__ off, uint64_t off_stride, int closed, unsigned long long* __restrict__ totals) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_run;
    const uint32_t r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t* row = m + (uint64_t)r * row_stride;
    if (tid == 0) s_run = row_base ? row_base[r] : 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 1024) {
        const uint32_t c = c0 + tid;
        const uint64_t v = c < n_cols ? row[c] : 0;
        uint64_t x = v;                                            // inclusive scan within the wave
        for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d, 64); if ((int)lane >= d) x += y; }
        if (lane == 63) s_wave[w] = x;
        __syncthreads();
        uint64_t before = s_run;
        for (uint32_t q = 0; q < w; ++q) before += s_wave[q];
        if (off && c < n_cols) off[(uint64_t)r * off_stride + c] = before + x - v;
        __syncthreads();
        if (tid == 1023) s_run = before + x;
        __syncthreads();
    }
    if (tid == 0) {
        if (off && closed) off[(uint64_t)r * off_stride + n_cols] = s_run;
        if (totals) totals[r] = s_run - (row_base ? row_base[r] : 0);
    }
}

// pass 2: write the records.  off[p * R + g] = global index of the first record of (part p, region g).  Counts that do
// not fit 32 bits travel in the `big` list (their record carries count 0, which a merge skips).
// PACKED: a record is what a packed slot holds of the k-mer -- the remainder (rb <= 44 bits) -- below its count, 72 bits cut into a u32, a u8
// and a u32: 9 bytes where key + count take 12.  The third word's low rb - 40 bits (none for rb <= 40) are the
// remainder's top ones; the count has the 32 - (rb - 40) >= 28 above them, and what does not fit travels in the `big` list like a count
// beyond 32 bits does.  The region the record lies in says the rest of the k-mer: the form for owners that share the sender's region grid.
__device__ __host__ __forceinline__ uint32_t rec_xs(uint32_t rb) { return rb > 40 ? rb - 40 : 0; }
struct PackedRec { uint64_t rem; uint32_t c; };
__device__ __forceinline__ PackedRec packed_rec(const uint32_t* __restrict__ lo, const uint8_t* __restrict__ hi, const uint32_t* __restrict__ cw, uint64_t i, uint32_t xs) {
    const uint32_t w = cw[i];
    PackedRec r;
    r.c = w >> xs;
    r.rem = r.c ? (((uint64_t)(w & ((1u << xs) - 1)) << 40) | ((uint64_t)hi[i] << 32) | lo[i]) : 0;     // (count 0: a record whose count travels out of band, skipped)
    return r;
}
template <bool PACKED>
static __global__ void __launch_bounds__(EXTRACT_BLOCK)
k_extract_write(DevTable t, uint32_t n_ovf, uint32_t n_parts, const uint64_t* __restrict__ off, uint64_t* __restrict__ out_keys,
                uint32_t* __restrict__ out_rem_lo, uint8_t* __restrict__ out_rem_hi,
                uint32_t* __restrict__ out_counts, uint64_t* __restrict__ big_keys, uint64_t* __restrict__ big_counts,
                unsigned long long* __restrict__ big_n, uint32_t big_cap) {
    __shared__ uint32_t s_cur[MAX_EXCHANGE_PARTS];
    const uint32_t tid = threadIdx.x, S = t.region_slots;
    for (uint32_t g = blockIdx.x; g < t.n_regions; g += gridDim.x) {
        for (uint32_t p = tid; p < n_parts; p += EXTRACT_BLOCK) s_cur[p] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)g * S;
        const RegionPlace rp = region_place(t, g);
        auto emit = [&](uint64_t pos, uint64_t key, uint64_t in_slot) {
            const uint32_t p = n_parts > 1 ? owner_of(key, t.k, n_parts) : 0;
            const uint64_t at = off[(uint64_t)p * t.n_regions + g] + atomicAdd(&s_cur[p], 1u);
            uint64_t c = slot_total(t, pos, key, in_slot, n_ovf);
            const uint32_t xs = PACKED ? rec_xs(rp.pl.rb) : 0;
            if (c > (0xFFFFFFFFULL >> xs)) {
                const unsigned long long b = atomicAdd(big_n, 1ULL);
                if (b < big_cap) { big_keys[b] = key; big_counts[b] = c; }
                c = 0;
            }
            if constexpr (PACKED) {
                const uint64_t rem = rem_in(key, rp);
                out_rem_lo[at] = (uint32_t)rem; out_rem_hi[at] = (uint8_t)(rem >> 32);
                out_counts[at] = c ? ((uint32_t)c << xs) | (uint32_t)(rem >> 40) : 0u;
            } else { out_keys[at] = key; out_counts[at] = (uint32_t)c; }
        };
        if (t.cbits) {                                             // (packed slots: as k_extract_count, four 16-byte loads per lane in flight)
            const ulonglong2* w2 = reinterpret_cast<const ulonglong2*>(t.keys + base);
            const uint32_t n2 = S >> 1;
            for (uint32_t i0 = 0; i0 < n2; i0 += EXTRACT_BLOCK * EXTRACT_UNROLL) {
                ulonglong2 w[EXTRACT_UNROLL];
#pragma unroll
                for (int u = 0; u < EXTRACT_UNROLL; ++u) { const uint32_t i = i0 + u * EXTRACT_BLOCK + tid; w[u] = i < n2 ? w2[i] : make_ulonglong2(0, 0); }
#pragma unroll
                for (int u = 0; u < EXTRACT_UNROLL; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint64_t x = h ? w[u].y : w[u].x;
                        if (x) emit(base + 2 * (uint64_t)(i0 + u * EXTRACT_BLOCK + tid) + h, key_in(pk_rem(x, t.cbits), rp), pk_count(x, t.cbits));
                    }
            }
        } else {
            for (uint32_t i = tid; i < S; i += EXTRACT_BLOCK) {
                const SlotView v = slot_view_in(t, rp, base + i);
                if (v.occ) emit(base + i, v.key, v.cnt);
            }
        }
        __syncthreads();
    }
}

struct MergeSrc { const uint64_t* keys; const uint32_t* counts; const uint64_t* off; const uint32_t* rem_lo; const uint8_t* rem_hi; };   // off: u64[regions + 1], relative to the record arrays; keys == null: packed records (rem_lo / rem_hi)
struct MergeSrcs { MergeSrc s[MAX_MERGE_SRC]; uint32_t n; uint32_t src_p1, src_n1, src_l2; };      // src_*: the grid packed records were cut from (the table's, unless it has grown since)
// the k-mer of record i of region g of a source (packed records: from the SENDER's grid)
__device__ __forceinline__ uint64_t merge_src_key(const DevTable& t, const MergeSrcs& srcs, const MergeSrc& s, uint32_t g, uint64_t i, uint32_t& c) {
    if (s.keys) { c = s.counts[i]; return s.keys[i]; }
    const Place pl = place_make(t.k, srcs.src_p1, srcs.src_n1, srcs.src_l2);
    const PackedRec r = packed_rec(s.rem_lo, s.rem_hi, s.counts, i, rec_xs(pl.rb));
    c = r.c;
    return place_key_d(g >> srcs.src_l2, g & ((1u << srcs.src_l2) - 1), r.rem, pl);
}

// Owner side: regions [g_lo, g_hi).  LDS: keys[S] (u64) | counts[S] (u32) (KV12) or the S packed words (PK); the protocol of the
// apply kernels with arbitrary 32-bit amounts.  A region that could overflow (occupied + incoming > S) is not touched: its index
// goes to `deferred` and the host sends its runs through the direct path after making room.
template <int BLOCK, bool PK>
__global__ void __launch_bounds__(BLOCK, 4)     // 512 threads, four waves per SIMD (<= 128 VGPRs: a batch of records in registers): two workgroups per CU, a region of <= 77 KB each
k_merge_apply(DevTable t, uint32_t g_lo, uint32_t g_hi, MergeSrcs srcs, uint32_t* __restrict__ deferred, unsigned long long* __restrict__ n_deferred,
              uint32_t zero_fill /* PK: the regions hold whatever the memory held (katgpu_table::zero_from): each starts from zeros in LDS instead of being loaded,
                                    and EVERY region of [g_lo, g_hi) is written -- one that nothing arrives for, or that is deferred, as zeros */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ uint32_t s_occ;
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(lds_raw);
    uint32_t* rc = reinterpret_cast<uint32_t*>(lds_raw + (size_t)t.region_slots * 8);      // (KV12 only)
    const uint32_t tid = threadIdx.x, S = t.region_slots, cb = t.cbits;
    const uint64_t cmask = PK ? pk_cmask(cb) : 0, half = PK ? pk_half(cb) : 0;
    uint32_t new_distinct = 0;
    for (uint32_t g = g_lo + blockIdx.x; g < g_hi; g += gridDim.x) {
        const uint32_t j = g - g_lo;
        uint64_t incoming = 0;
        for (uint32_t s = 0; s < srcs.n; ++s) incoming += srcs.s[s].off[j + 1] - srcs.s[s].off[j];
        const uint64_t base = (uint64_t)g * S;
        if (incoming == 0 || (zero_fill && incoming > S)) {                      // uniform over the workgroup
            if (zero_fill) {                                                     // (S % 4 == 0: regions are 32-byte aligned)
                ulonglong2* z = reinterpret_cast<ulonglong2*>(t.keys + base);
                for (uint32_t i = tid; i < (S >> 1); i += BLOCK) z[i] = make_ulonglong2(0, 0);
            }
            if (incoming && tid == 0) deferred[atomicAdd(n_deferred, 1ULL)] = g;
            continue;
        }
        if (tid == 0) s_occ = 0;
        __syncthreads();
        const RegionPlace rp = region_place(t, g);
        uint32_t occ = 0;
        if (PK && zero_fill) {
            for (uint32_t i = tid; i < S; i += BLOCK) rk[i] = 0;
        } else {
            for (uint32_t i = tid; i < S; i += BLOCK) {
                const uint64_t key = t.keys[base + i];
                rk[i] = key;
                if constexpr (PK) occ += key != 0; else { rc[i] = t.counts[base + i]; occ += key != EMPTY; }
            }
        }
        for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d, 64);
        if ((tid & 63) == 0 && occ) atomicAdd(&s_occ, occ);
        __syncthreads();
        if ((uint64_t)s_occ + incoming > S) {                                    // uniform
            if (tid == 0) deferred[atomicAdd(n_deferred, 1ULL)] = g;
            __syncthreads();
            continue;
        }
        for (uint32_t s = 0; s < srcs.n; ++s) {
            const uint64_t beg = srcs.s[s].off[j], end = srcs.s[s].off[j + 1];
            // A thread has a handful of records per region and source, and applying one is a chain of LDS round trips behind three loads: the
            // loads of MERGE_BATCH records are issued before the first is applied (the kernel waited for memory 0.71 of its wave cycles
            // fetching them one at a time).
            // A source's run of a region is in slot order (the extraction walks the region): lanes that took CONSECUTIVE records would all claim
            // slots of one neighbourhood at once and a wave would wait for the longest chain of claims among them, step after step (24 of the
            // kernel's 28 ms).  A thread takes a BLOCK of consecutive records instead: a wave's lanes work a dozen slots apart.
            const uint64_t per_thread = (end - beg + BLOCK - 1) / BLOCK;
            const uint64_t t_beg = beg + (uint64_t)tid * per_thread, t_end = t_beg + per_thread < end ? t_beg + per_thread : end;
            for (uint64_t i0 = t_beg; i0 < t_end; i0 += MERGE_BATCH) {
              uint32_t bc[MERGE_BATCH]; uint64_t brem[MERGE_BATCH];
#pragma unroll
              for (int u = 0; u < MERGE_BATCH; ++u) {
                const uint64_t i = i0 + (uint64_t)u;
                bc[u] = 0; brem[u] = 0;
                if (i < t_end) {
                    if (PK && !srcs.s[s].keys) { const PackedRec r = packed_rec(srcs.s[s].rem_lo, srcs.s[s].rem_hi, srcs.s[s].counts, i, rec_xs(rp.pl.rb)); bc[u] = r.c; brem[u] = r.rem; }   // (packed records: this table's grid is the sender's -- the host checked)
                    else { bc[u] = srcs.s[s].counts[i]; brem[u] = PK ? rem_in(srcs.s[s].keys[i], rp) : srcs.s[s].keys[i]; }
                }
              }
#pragma unroll
              for (int u = 0; u < MERGE_BATCH; ++u) {
                const uint32_t c = bc[u];
                uint64_t rem = brem[u];
                if (!c) continue;
                if constexpr (PK) {
                    uint32_t slot = place_offset(rem, rp.pl, S);
                    uint64_t q, r;
                    pk_split((uint64_t)c, cb, q, r);
                    const uint64_t in = r ? r : half, qq0 = r ? q : q - 1;
                    const unsigned long long claim = (unsigned long long)((rem << cb) | in);
                    for (uint32_t probe = 0; probe < S; ++probe) {               // cannot fail: occupied + incoming <= S
                        // the claim is tried outright: what it returns is what a read would have (one LDS round trip per step, not two)
                        unsigned long long w = atomicCAS(&rk[slot], 0ULL, claim);
                        if (w == 0) { ++new_distinct; if (qq0) ovf_add(t, base + slot, qq0 * half); break; }
                        if ((w >> cb) == rem) {
                            if (r) {
                                for (;;) {
                                    uint64_t cc = (w & cmask) + r, qq = q;
                                    if (cc > half) { cc -= half; ++qq; }                  // (the invariant of packed slots: 1 .. half)
                                    const unsigned long long got = atomicCAS(&rk[slot], w, (unsigned long long)((w & ~cmask) | cc));
                                    if (got == w) { q = qq; break; }
                                    w = got;
                                }
                            }
                            if (q) ovf_add(t, base + slot, q * half);
                            break;
                        }
                        slot = slot + 1 == S ? 0 : slot + 1;
                    }
                } else {
                    const unsigned long long key = rem;                          // (KV12 tables: never packed records; brem holds the key)
                    uint32_t slot = home_offset_in(key, rp);
                    for (uint32_t probe = 0; probe < S; ++probe) {               // cannot fail: occupied + incoming <= S
                        unsigned long long cur = rk[slot];
                        if (cur == EMPTY) {
                            cur = atomicCAS(&rk[slot], (unsigned long long)EMPTY, key);
                            if (cur == EMPTY) { ++new_distinct; cur = key; }
                        }
                        if (cur == key) {
                            const uint32_t old = atomicAdd(&rc[slot], c);
                            if ((uint32_t)(old + c) < old) ovf_add(t, key, 1ULL << 32);
                            break;
                        }
                        slot = slot + 1 == S ? 0 : slot + 1;
                    }
                }
              }
            }
        }
        __syncthreads();
        if constexpr (PK) {                                        // (16-byte stores: S % 4 == 0)
            ulonglong2* o2 = reinterpret_cast<ulonglong2*>(t.keys + base);
            const ulonglong2* r2 = reinterpret_cast<const ulonglong2*>(rk);
            for (uint32_t i = tid; i < (S >> 1); i += BLOCK) o2[i] = r2[i];
        } else
        for (uint32_t i = tid; i < S; i += BLOCK) { t.keys[base + i] = rk[i]; if constexpr (!PK) t.counts[base + i] = rc[i]; }
        __syncthreads();
    }
    flush_distinct(t, new_distinct);
}

// The runs of the regions k_merge_apply deferred, through the direct path: one workgroup per deferred region, every source's
// run of that region.  (One launch for all of them: when a whole table is too small, every region is on the list.)
static __global__ void __launch_bounds__(256)
k_merge_deferred(DevTable t, uint32_t g_lo, MergeSrcs srcs, const uint32_t* __restrict__ deferred, uint32_t n_deferred) {
    uint32_t new_distinct = 0;
    for (uint32_t d = blockIdx.x; d < n_deferred; d += gridDim.x) {
        const uint32_t j = deferred[d] - g_lo;
        for (uint32_t s = 0; s < srcs.n; ++s) {
            const uint64_t beg = srcs.s[s].off[j], end = srcs.s[s].off[j + 1];
            for (uint64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
                uint32_t c;
                const uint64_t key = merge_src_key(t, srcs, srcs.s[s], deferred[d], i, c);
                if (c) table_add(t, key, (uint64_t)c, new_distinct);
            }
        }
    }
    flush_distinct(t, new_distinct);
}

// records with 32-bit counts through the direct path (sources of another grid, deferred regions)
static __global__ void __launch_bounds__(256)
k_merge32(DevTable dst, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i]) table_add(dst, keys[i], (uint64_t)counts[i], new_distinct);
    flush_distinct(dst, new_distinct);
}

// ---- synthetic workload (bench / tests), bit-identical to kat_amd/synth.py ----
__device__ __forceinline__ uint32_t genome_code(uint64_t seed, uint64_t i) {
    return (uint32_t)(rng2(seed, i >> 5) >> (2 * (i & 31))) & 3;
}
// contig_len == 0: n bases.  contig_len > 0: the assembly's base stream, an 'N' after every contig_len bases
// (n counts output bytes, separators included).
static __global__ void __launch_bounds__(256)
k_synth_genome(uint8_t* __restrict__ out, uint64_t n, uint64_t seed, uint64_t contig_len) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += stride) {
        uint64_t i = x;
        if (contig_len) {
            uint64_t c = x / (contig_len + 1), j = x - c * (contig_len + 1);
            if (j == contig_len) { out[x] = 'N'; continue; }
            i = c * contig_len + j;
        }
        out[x] = "ACGT"[genome_code(seed, i)];
    }
}
// one lane per base: reads are laid out with stride read_len+1 ('N' separator closes every record)
static __global__ void __launch_bounds__(256)
k_synth_reads(const uint8_t* __restrict__ genome, uint64_t genome_len, uint8_t* __restrict__ out, uint64_t first_read,
              uint64_t n_reads, uint32_t read_len, uint32_t frag_len, uint32_t err_thresh, uint64_t seed) {
    const uint64_t rec = (uint64_t)read_len + 1, total = n_reads * rec;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride) {
        uint64_t rl = x / rec;
        uint32_t j = (uint32_t)(x - rl * rec);
        if (j == read_len) { out[x] = 'N'; continue; }
        uint64_t r = first_read + rl, pair = r >> 1, mate = r & 1;
        uint64_t u = rng2(seed, pair);
        uint64_t start = __umul64hi(u, genome_len - frag_len + 1);
        uint64_t strand = splitmix64(u) >> 63;
        bool fwd = (strand ^ mate) == 0;
        uint8_t g = genome[fwd ? start + j : start + frag_len - 1 - j];
        uint32_t code = ((g >> 1) & 3) ^ ((g >> 2) & 1);
        if (!fwd) code = 3 - code;
        uint64_t e = rng2(seed ^ 0x5EED5EED5EED5EEDULL, r * 1024 + j);
        if ((uint32_t)e < err_thresh) code = (code + 1 + (uint32_t)((e >> 32) % 3)) & 3;
        out[x] = "ACGT"[code];
    }
}

}  // namespace kg
