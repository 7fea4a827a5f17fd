// kg_partition.hpp -- the partitioned counter: K1 without a DRAM/L2 atomic per k-mer.
//
// Measured on MI355X (tools/ubench_count.hip, profiles/): extracting + canonicalising + hashing k-mers runs at
// ~490 G k-mers/s, random 8-byte loads at ~55 G/s, but device-scope atomic adds saturate the L2 atomic units at
// ~22 G/s whatever the table size -- so the direct kernel (k_count: one atomic per instance) tops out near 15 G k-mers/s.
// This path removes the per-instance global atomic.  The table is made of regions of `region_slots` slots with
// region-local probing (kg_device.hpp: Probe); a round of the partitioned counter
//   P1  radix-partitions the round's k-mers by the high part of their region index into P1 buckets  (8 B out / k-mer)
//   P2  splits every bucket by the low part of the region index -> one contiguous run per region     (8 B in, 4 + HB B out)
//   P3  loads a region (96 KB) into LDS, applies its run with LDS atomics, writes the region back     (4 + HB B in + 24 B/slot)
// What level 2 writes is not the k-mer but the REMAINDER of its placement hash (kg_device.hpp "placement": the hash is one to one
// and the region already says its digits): rb = 2k - log2(regions) bits, kept as two streams -- the low 32 bits and HB = 0, 1, 2
// or 4 bytes of high bits (35 bits at the bench size: 5 bytes per k-mer instead of 8).
// so HBM sees only streaming traffic.  Level 1 is exact two-pass (histogram, scan, scatter) with per-workgroup running
// cursors held in LDS: no global atomic per k-mer, deterministic placement.  Level 2 normally runs in ONE pass
// (k_p2_fast: equal-capacity runs sized from the uniform hash, an overflow list for what does not fit, the exact
// histogram/scan/scatter kernel k_p2 as the fall back).  A region that fills up in P3 (more distinct k-mers than slots)
// spills the k-mer to a list (held in the then-free P1 buffer, so it can never overflow) that is inserted through the
// direct path after a regrow.
#pragma once
#include "kg_kernels.hpp"

namespace kg {

constexpr int PART_BLOCK = 1024;                              // 16 waves, one workgroup per CU
constexpr int PART_ITEMS = 16;                                // k-mers per lane per tile
constexpr int TILE_ITEMS = PART_BLOCK * PART_ITEMS;           // 16384
constexpr int L1_TILE_BYTES = TILE_ITEMS;                     // bytes staged per tile (16 per lane)
constexpr int L1_TILE_STARTS = L1_TILE_BYTES - CHUNK_OVERLAP; // 16352 window starts per tile
constexpr int L1_LANES_WITH_STARTS = L1_TILE_STARTS / PART_ITEMS;   // 1022
constexpr int MAX_PARTS = 1024;                               // buckets per level (one lane per bucket in the scans)

// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a workgroup-scope release fence, which on gfx950
// lowers to s_waitcnt vmcnt(0): every barrier placed after a run of global STORES (copy-out, region write-back) would
// stall the whole workgroup until those stores are acknowledged by L2.  Nothing in these kernels hands global data from
// one lane to another inside a launch, so only LDS traffic has to be complete here.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PartGeom {
    uint32_t R, S;     // regions, slots per region (the table's)
    uint32_t P1, P2;   // region r = b1 * P2 + b2 (the table's p1, p2): level-1 bucket b1, level-2 bucket b2
    uint32_t l2;       // P2 == 1 << l2
    uint32_t hb;       // bytes of a level-2 item beyond its low 32 bits: 0, 1, 2 or 4 (from pl.rb)
    Place pl;          // the placement hash's bit budget for this table (kg_device.hpp)
};
// the two streams of a level-2 buffer of `cap` items: low words first, high parts behind them
__device__ __host__ __forceinline__ const void* l2_hi_of(const uint32_t* lo, uint64_t cap) { return lo + cap; }
__device__ __host__ __forceinline__ void* l2_hi_of(uint32_t* lo, uint64_t cap) { return lo + cap; }
__device__ __host__ __forceinline__ uint32_t l2_hi_bytes(uint32_t rb) { return rb <= 32 ? 0u : rb <= 40 ? 1u : rb <= 48 ? 2u : 4u; }
template <int HB> struct HiWord { typedef uint32_t type; };
template <> struct HiWord<1> { typedef uint8_t type; };
template <> struct HiWord<2> { typedef uint16_t type; };
template <int HB>
__device__ __forceinline__ void l2_store(uint32_t* __restrict__ lo, void* __restrict__ hi, uint64_t at, uint64_t rem) {
    lo[at] = (uint32_t)rem;
    if (HB) reinterpret_cast<typename HiWord<HB>::type*>(hi)[at] = (typename HiWord<HB>::type)(rem >> 32);
}
template <int HB>
__device__ __forceinline__ uint64_t l2_load(const uint32_t* __restrict__ lo, const void* __restrict__ hi, uint64_t at) {
    const uint32_t l = lo[at];
    const uint32_t h = HB ? (uint32_t)reinterpret_cast<const typename HiWord<HB>::type*>(hi)[at] : 0u;
    return ((uint64_t)h << 32) | l;
}
// (slow paths: the width is a run-time number)
__device__ __forceinline__ uint64_t l2_load_any(uint32_t hb, const uint32_t* __restrict__ lo, const void* __restrict__ hi, uint64_t at) {
    return hb == 0 ? l2_load<0>(lo, hi, at) : hb == 1 ? l2_load<1>(lo, hi, at) : hb == 2 ? l2_load<2>(lo, hi, at) : l2_load<4>(lo, hi, at);
}

// LDS carve of the partition kernels (dynamic shared memory, 16-byte aligned base)
struct PartLds {
    uint64_t staging[TILE_ITEMS];      // 128 KB: the tile's k-mers grouped by bucket
    uint64_t cursor[MAX_PARTS];        // running output position of each bucket for THIS workgroup
    uint32_t hist[MAX_PARTS];          // per-tile (scatter) or accumulated (count) bucket sizes
    uint32_t off[MAX_PARTS];           // exclusive scan of hist
    uint32_t wave_tot[32];
    uint32_t code[PART_BLOCK + 2];
    uint32_t bad[PART_BLOCK + 2];
};

// exclusive scan of v over the first MAX_PARTS lanes of a 1024-thread block (lane b holds bucket b); returns the
// exclusive prefix, *total gets the grand total.  Two barriers inside.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    uint32_t prefix = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PART_BLOCK / 64; ++w) { uint32_t x = wave_tot[w]; if ((uint32_t)w < wave) prefix += x; tot += x; }
    *total = tot;
    lds_barrier();
    return prefix + inc - v;
}

// 64-bit variant (scratch: 16 u64 in LDS)
__device__ __forceinline__ uint64_t block_exclusive_scan64(uint64_t v, uint64_t* wave_tot) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint64_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    lds_barrier();                                  // every lane has read its own histogram word before scratch is written
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    uint64_t prefix = 0;
#pragma unroll
    for (int w = 0; w < PART_BLOCK / 64; ++w) { uint64_t x = wave_tot[w]; if ((uint32_t)w < wave) prefix += x; }
    lds_barrier();
    return prefix + inc - v;
}

// One tile of the base stream is 16 bytes per lane.  The load is split from the staging so that the NEXT tile's load can be
// in flight while the current tile is processed (one workgroup per CU: nothing else hides the ~2 us HBM latency).
__device__ __forceinline__ void tile_load(const uint8_t* __restrict__ bases, uint64_t n, uint64_t tile_off, uint32_t (&w)[4]) {
    const uint64_t off = tile_off + (uint64_t)threadIdx.x * PART_ITEMS;
    if (off + PART_ITEMS <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(bases + off);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { uint64_t i = off + q * 4 + b; x |= (i < n ? (uint32_t)bases[i] : (uint32_t)'N') << (8 * b); }
            w[q] = x;
        }
    }
}

// 2-bit codes + validity flags of the loaded tile into LDS.  Ends with a barrier.
__device__ __forceinline__ void tile_stage(PartLds& L, const uint32_t (&w)[4]) {
    const uint32_t tid = threadIdx.x;
    uint32_t code, bad;
    encode16(w, code, bad);
    L.code[tid] = code;
    L.bad[tid] = bad;
    if (tid < 2) { L.code[PART_BLOCK + tid] = 0; L.bad[PART_BLOCK + tid] = 0xFFFF; }
    lds_barrier();
}

// The 16 k-mers whose windows start in this lane's 16 positions (canonical if asked); bit j of the result = window j valid.
__device__ __forceinline__ uint32_t lane_kmers(const PartLds& L, uint32_t k, bool canonical, uint64_t (&key)[PART_ITEMS]) {
    const uint32_t tid = threadIdx.x;
    uint32_t valid = 0;
    if (tid >= L1_LANES_WITH_STARTS) return 0;
    uint64_t hi = ((uint64_t)L.code[tid] << 32) | L.code[tid + 1];
    uint64_t lo = (uint64_t)L.code[tid + 2] << 32;
    uint64_t m = ((uint64_t)L.bad[tid] << 48) | ((uint64_t)L.bad[tid + 1] << 32) | ((uint64_t)L.bad[tid + 2] << 16);
    const uint32_t kshift = 64 - 2 * k, mshift = 64 - k;
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        uint64_t fwd = hi >> kshift;
        uint64_t kk = fwd;
        if (canonical) { uint64_t rc = kmer_revcomp(fwd, k); kk = rc < fwd ? rc : fwd; }
        key[j] = kk;
        if ((m >> mshift) == 0) valid |= 1u << j;
        hi = (hi << 2) | (lo >> 62);
        lo <<= 2;
        m <<= 1;
    }
    return valid;
}

// ---- level 1 scan: offs[w][b] = start of workgroup w's run inside bucket b; l1_off[b] = start of bucket b; l1_off[P1] = items ----
__global__ void __launch_bounds__(PART_BLOCK)
k_p1_scan(PartGeom g, uint32_t n_wg, const uint32_t* __restrict__ hist1, uint64_t* __restrict__ offs, uint64_t* __restrict__ l1_off) {
    __shared__ uint64_t s_base[MAX_PARTS + 1];
    const uint32_t b = threadIdx.x;
    uint64_t tot = 0;
    if (b < g.P1) for (uint32_t w = 0; w < n_wg; ++w) tot += hist1[(uint64_t)w * g.P1 + b];
    if (b < g.P1) s_base[b] = tot;
    lds_barrier();
    if (b == 0) {                                   // P1 <= 1024 entries: a serial scan is a few microseconds
        uint64_t run = 0;
        for (uint32_t i = 0; i < g.P1; ++i) { uint64_t v = s_base[i]; s_base[i] = run; run += v; }
        s_base[g.P1] = run;
    }
    lds_barrier();
    if (b < g.P1) l1_off[b] = s_base[b];
    if (b == 0) l1_off[g.P1] = s_base[g.P1];
    if (b < g.P1) {
        uint64_t run = s_base[b];
        for (uint32_t w = 0; w < n_wg; ++w) { offs[(uint64_t)w * g.P1 + b] = run; run += hist1[(uint64_t)w * g.P1 + b]; }
    }
}

// =====================================================================================================================
// Level 1, second edition: 512-thread workgroups that stage 2-byte tile POSITIONS instead of 8-byte k-mers.  The k-mer is
// recomputed from the LDS-resident 2-bit codes when its run is copied out (three LDS words + shifts + v_bfrev, cheap),
// which shrinks the workgroup's LDS from 156 KB to 36 KB: three or four workgroups share a CU and overlap each
// other's barrier-separated phases (the 1024-thread edition above sits parked on barriers / memory 51 % of the time).
constexpr int P1_BLOCK = 512;
constexpr int P1_TILE_BYTES = P1_BLOCK * PART_ITEMS;              // 8192
constexpr int P1_TILE_STARTS = P1_TILE_BYTES - CHUNK_OVERLAP;     // 8160
constexpr int P1_LANES_WITH_STARTS = P1_TILE_STARTS / PART_ITEMS; // 510

struct P1Lds {
    uint64_t cursor[MAX_PARTS];
    uint32_t hist[MAX_PARTS];
    uint32_t off[MAX_PARTS];
    uint32_t wave_tot[16];
    uint32_t code[P1_BLOCK + 2];
    uint32_t bad[P1_BLOCK + 2];
    uint32_t pos[P1_TILE_BYTES];        // per staged k-mer: bucket << 16 | tile position  (52 KB in all: three workgroups per CU)
};

struct LaneWindow {                       // the 96-bit sliding window of kg_kernels.hpp's K1, as an object
    uint64_t hi, lo, m;
    uint32_t kshift, mshift;
    __device__ __forceinline__ void init(const uint32_t* code, const uint32_t* bad, uint32_t w, uint32_t k) {
        hi = ((uint64_t)code[w] << 32) | code[w + 1];
        lo = (uint64_t)code[w + 2] << 32;
        m = ((uint64_t)bad[w] << 48) | ((uint64_t)bad[w + 1] << 32) | ((uint64_t)bad[w + 2] << 16);
        kshift = 64 - 2 * k; mshift = 64 - k;
    }
    __device__ __forceinline__ bool valid() const { return (m >> mshift) == 0; }
    __device__ __forceinline__ uint64_t fwd() const { return hi >> kshift; }
    __device__ __forceinline__ void step() { hi = (hi << 2) | (lo >> 62); lo <<= 2; m <<= 1; }
};

__device__ __forceinline__ uint64_t canon_if(uint64_t fwd, uint32_t k, bool canonical) {
    if (!canonical) return fwd;
    const uint64_t rc = kmer_revcomp(fwd, k);
    return rc < fwd ? rc : fwd;
}

// the k-mer whose window starts at tile position p, from the staged codes
__device__ __forceinline__ uint64_t kmer_at(const uint32_t* code, uint32_t p, uint32_t k, bool canonical) {
    const uint32_t w = p >> 4, o = p & 15;
    uint64_t hi = ((uint64_t)code[w] << 32) | code[w + 1];
    if (o) hi = (hi << (2 * o)) | ((uint64_t)code[w + 2] >> (32 - 2 * o));
    return canon_if(hi >> (64 - 2 * k), k, canonical);
}

__device__ __forceinline__ void p1_tile_load(const uint8_t* __restrict__ bases, uint64_t n, uint64_t tile_off, uint32_t (&w)[4]) {
    const uint64_t off = tile_off + (uint64_t)threadIdx.x * PART_ITEMS;
    if (off + PART_ITEMS <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(bases + off);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { uint64_t i = off + q * 4 + b; x |= (i < n ? (uint32_t)bases[i] : (uint32_t)'N') << (8 * b); }
            w[q] = x;
        }
    }
}

__device__ __forceinline__ void p1_tile_stage(P1Lds& L, const uint32_t (&w)[4]) {
    const uint32_t tid = threadIdx.x;
    uint32_t code, bad;
    encode16(w, code, bad);
    L.code[tid] = code;
    L.bad[tid] = bad;
    if (tid < 2) { L.code[P1_BLOCK + tid] = 0; L.bad[P1_BLOCK + tid] = 0xFFFF; }
    lds_barrier();
}

// exclusive scan over 2 * P1_BLOCK logical entries (entry b and b + 512 per lane); three barriers
__device__ __forceinline__ void p1_scan_pair(uint32_t v0, uint32_t v1, uint32_t* wave_tot, uint32_t& e0, uint32_t& e1) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t i0 = v0, i1 = v1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t a = __shfl_up(i0, d, 64), b = __shfl_up(i1, d, 64);
        if (lane >= (uint32_t)d) { i0 += a; i1 += b; }
    }
    lds_barrier();
    if (lane == 63) { wave_tot[wave] = i0; wave_tot[8 + wave] = i1; }
    lds_barrier();
    uint32_t p0 = 0, p1 = 0, t0 = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { uint32_t x = wave_tot[w], y = wave_tot[8 + w]; t0 += x; if ((uint32_t)w < wave) { p0 += x; p1 += y; } }
    e0 = p0 + i0 - v0;
    e1 = t0 + p1 + i1 - v1;
}

__global__ void __launch_bounds__(P1_BLOCK)
k_p1v2_count(DevTable t, PartGeom g, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
             uint32_t* __restrict__ hist1) {
    __shared__ uint32_t s_hist[MAX_PARTS];
    __shared__ uint32_t s_code[P1_BLOCK + 2];
    __shared__ uint32_t s_bad[P1_BLOCK + 2];
    const uint32_t tid = threadIdx.x;
    const bool canonical = t.canonical != 0;
    for (uint32_t b = tid; b < MAX_PARTS; b += P1_BLOCK) s_hist[b] = 0;
    uint32_t ones = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    uint32_t w[4], wn[4];
    if (t0 < t1) p1_tile_load(bases, n, t0 * P1_TILE_STARTS, w);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) p1_tile_load(bases, n, (tile + 1) * P1_TILE_STARTS, wn);
        lds_barrier();
        uint32_t code, bad;
        encode16(w, code, bad);
        s_code[tid] = code; s_bad[tid] = bad;
        if (tid < 2) { s_code[P1_BLOCK + tid] = 0; s_bad[P1_BLOCK + tid] = 0xFFFF; }
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = wn[q];
        if (tid < P1_LANES_WITH_STARTS) {
            LaneWindow lw;
            lw.init(s_code, s_bad, tid, t.k);
#pragma unroll 4
            for (int j = 0; j < PART_ITEMS; ++j, lw.step()) {
                if (!lw.valid()) continue;
                const uint64_t key = canon_if(lw.fwd(), t.k, canonical);
                if (key == EMPTY) { ++ones; continue; }
                atomicAdd(&s_hist[place_digit1(place_stage1(key, g.pl), g.pl)], 1u);
            }
        }
    }
    lds_barrier();
    for (uint32_t b = tid; b < g.P1; b += P1_BLOCK) hist1[(uint64_t)blockIdx.x * g.P1 + b] = s_hist[b];
    for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
    if ((tid & 63) == 0 && ones) atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)ones);
}

// SEG = false: the exact edition -- every workgroup's share of every bucket was counted first (k_p1v2_count, k_p1_scan), `offs`
// holds where it starts.  SEG = true: no counting pass -- bucket b is cut into one SEGMENT of seg_cap k-mers per workgroup
// (segment (b, w) starts at (b * gridDim + w) * seg_cap); the hash spreads a workgroup's k-mers evenly, so a capacity of the
// expected share + 1/24 + 64 holds them (5 sigma at the bench size); what a full segment cannot take goes to the overflow list
// (-> direct path; if that overflows too the host redoes the round with the exact edition), what a segment has left at the
// end is padded with EMPTY, which level 2 skips (the all-ones k-mer never is an item).  Saves the second decode + hash of the
// whole input (k_p1v2_count: 70 ms of the bench step) for ~4 % more level-1 bytes.
template <bool SEG>
__global__ void __launch_bounds__(P1_BLOCK, 6)                // six waves per SIMD = three workgroups per CU (g_p1_wgs): at most 80 VGPRs
k_p1v2_scatter(DevTable t, PartGeom g, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
               const uint64_t* __restrict__ offs, uint64_t* __restrict__ l1_buf, uint64_t seg_cap, uint64_t* __restrict__ ovf_buf,
               unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap) {
    __shared__ __attribute__((aligned(16))) P1Lds L;
    const uint32_t tid = threadIdx.x, P = g.P1, k = t.k;
    const bool canonical = t.canonical != 0;
    uint32_t ones = 0;
    auto seg_base = [&](uint32_t b) -> uint64_t { return ((uint64_t)b * gridDim.x + blockIdx.x) * seg_cap; };
    for (uint32_t b = tid; b < P; b += P1_BLOCK) L.cursor[b] = SEG ? seg_base(b) : offs[(uint64_t)blockIdx.x * P + b];
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    uint32_t w[4], wn[4];
    if (t0 < t1) p1_tile_load(bases, n, t0 * P1_TILE_STARTS, w);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) p1_tile_load(bases, n, (tile + 1) * P1_TILE_STARTS, wn);
        lds_barrier();                                   // previous tile's copy-out / cursor update done
        for (uint32_t b = tid; b < MAX_PARTS; b += P1_BLOCK) L.hist[b] = 0;
        p1_tile_stage(L, w);                               // ends with a barrier
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = wn[q];
        // sweep 1: bucket and rank of every valid window of this lane
        uint32_t br[PART_ITEMS];
        uint32_t valid = 0;
        if (tid < P1_LANES_WITH_STARTS) {
            LaneWindow lw;
            lw.init(L.code, L.bad, tid, k);
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j, lw.step()) {
                br[j] = 0;
                if (!lw.valid()) continue;
                const uint64_t key = canon_if(lw.fwd(), k, canonical);
                if (key == EMPTY) { if (SEG) ++ones; continue; }               // exact edition: tallied by the count pass
                const uint32_t b = place_digit1(place_stage1(key, g.pl), g.pl);
                br[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
                valid |= 1u << j;
            }
        }
        lds_barrier();
        uint32_t e0, e1;
        p1_scan_pair(tid < P ? L.hist[tid] : 0, tid + P1_BLOCK < P ? L.hist[tid + P1_BLOCK] : 0, L.wave_tot, e0, e1);
        L.off[tid] = e0;
        L.off[tid + P1_BLOCK] = e1;
        lds_barrier();
        // sweep 2: park the tile position of every k-mer in its bucket's run
#pragma unroll
        for (int j = 0; j < PART_ITEMS; ++j)
            if (valid >> j & 1) L.pos[L.off[br[j] >> 16] + (br[j] & 0xFFFF)] = (br[j] & 0xFFFF0000u) | (tid * PART_ITEMS + j);
        lds_barrier();
        // copy-out, one staged k-mer per lane and step: its bucket travels with its position, so no lane idles on a short run
        // and the steps are independent of each other (a loop over buckets serialised ~24 LDS round trips per lane group and
        // was 57 % of this kernel: cycle stamps; same-box A/B 228 -> 216 ms).  Neighbouring lanes still write neighbouring addresses
        // inside a run.
        const uint32_t total = L.off[P - 1] + L.hist[P - 1];
        for (uint32_t idx = tid; idx < total; idx += P1_BLOCK) {
            const uint32_t v = L.pos[idx], b = v >> 16;
            const uint64_t dst = L.cursor[b] + (idx - L.off[b]);
            const uint64_t key1 = kmer_at(L.code, v & 0xFFFF, k, canonical);
            if (!SEG || dst < seg_base(b) + seg_cap) l1_buf[dst] = key1;
            else {                                                             // the segment is full: the overflow list
                const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                if (at < ovf_cap) ovf_buf[at] = key1;
            }
        }
        lds_barrier();
        for (uint32_t b = tid; b < P; b += P1_BLOCK) {
            uint64_t c = L.cursor[b] + L.hist[b];
            if (SEG) { const uint64_t lim = seg_base(b) + seg_cap; c = c < lim ? c : lim; }
            L.cursor[b] = c;
        }
    }
    if (SEG) {
        lds_barrier();
        // what the segments have left is padded; a 16-lane group per bucket
        const uint32_t grp = tid >> 4, l16 = tid & 15;
        for (uint32_t b = grp; b < P; b += P1_BLOCK / 16) {
            const uint64_t lim = seg_base(b) + seg_cap;
            for (uint64_t i = L.cursor[b] + l16; i < lim; i += 16) l1_buf[i] = EMPTY;
        }
        for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
        if ((tid & 63) == 0 && ones) atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)ones);
    }
}

// where bucket b1 of the level-1 buffer lies: exact layout (l1_off) or segmented (seg_slots = workgroups x seg_cap slots per
// bucket, EMPTY-padded)
__device__ __forceinline__ void l1_bucket_range(const uint64_t* __restrict__ l1_off, uint64_t seg_slots, uint32_t b1, uint64_t& beg, uint64_t& end) {
    if (seg_slots) { beg = (uint64_t)b1 * seg_slots; end = beg + seg_slots; }
    else { beg = l1_off[b1]; end = l1_off[b1 + 1]; }
}

// ---- level 2 ----
// Counting-sort one tile's k-mers of bucket b1 by their level-2 digit through LDS and append every digit's run at this workgroup's
// cursor.  What is staged and written is the placement hash below the level-1 digit (y2 = digit : remainder, kg_device.hpp), so
// the copy-out reads its digit off the staged word instead of hashing again, and what reaches HBM is the remainder alone.
// BOUNDED: a run has a capacity (lim[]); what does not fit goes to the overflow list as a k-mer (through the inverse hash).
// All 1024 lanes must call it (barriers inside).
template <int HB, bool BOUNDED>
__device__ __forceinline__ void scatter_tile2(PartLds& L, const PartGeom g, const uint64_t base1, const uint64_t (&key)[PART_ITEMS], uint32_t valid,
                                              uint32_t* __restrict__ out_lo, void* __restrict__ out_hi, const uint64_t* lim, uint64_t* __restrict__ ovf_buf,
                                              unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap) {
    const uint32_t tid = threadIdx.x;
    const uint32_t P = g.P2;
    if (tid < MAX_PARTS) L.hist[tid] = 0;
    lds_barrier();
    uint32_t br[PART_ITEMS];                                  // digit << 16 | rank inside the tile's run
    uint64_t y2[PART_ITEMS];
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        br[j] = 0; y2[j] = 0;
        if (valid >> j & 1) {
            y2[j] = place_stage2(place_stage1(key[j], g.pl) - base1, g.pl);
            const uint32_t b = place_digit2(y2[j], g.pl);
            br[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
        }
    }
    lds_barrier();
    uint32_t total;
    const uint32_t mine = tid < P ? L.hist[tid] : 0;
    const uint32_t excl = block_exclusive_scan(mine, L.wave_tot, &total);
    if (tid < MAX_PARTS) L.off[tid] = excl;
    lds_barrier();
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j)
        if (valid >> j & 1) L.staging[L.off[br[j] >> 16] + (br[j] & 0xFFFF)] = y2[j];
    lds_barrier();
    // copy-out, one staged k-mer per lane and step (as in k_p1v2_scatter): the steps are independent, where a loop over the runs
    // kept a third of the lanes busy
    for (uint32_t idx = tid; idx < total; idx += PART_BLOCK) {
        const uint64_t staged = L.staging[idx];
        const uint32_t b = place_digit2(staged, g.pl);
        const uint64_t dst = L.cursor[b] + (idx - L.off[b]);
        if (!BOUNDED || dst < lim[b]) l2_store<HB>(out_lo, out_hi, dst, place_rem(staged, g.pl));
        else {                                                             // beyond the run's capacity: the overflow list
            const unsigned long long at = atomicAdd(ovf_n, 1ULL);
            if (at < ovf_cap) ovf_buf[at] = place_key(base1, staged, g.pl);
        }
    }
    lds_barrier();
    if (tid < P) {
        if (BOUNDED) { const uint64_t room = lim[tid] - L.cursor[tid]; L.cursor[tid] += (uint64_t)L.hist[tid] <= room ? L.hist[tid] : room; }
        else L.cursor[tid] += L.hist[tid];
    }
    // (the next tile's first barrier orders this update before the next use)
}

// the 16 k-mers of a lane for one tile of bucket [beg, end) of the level-1 buffer; bit j of `valid` = item j is a k-mer (not past
// the end, not segment padding)
__device__ __forceinline__ uint32_t p2_tile_load(const uint64_t* __restrict__ l1_buf, uint64_t tbeg, uint64_t end, bool padded, uint64_t (&key)[PART_ITEMS]) {
    uint32_t valid = 0;
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        const uint64_t i = tbeg + (uint64_t)j * PART_BLOCK + threadIdx.x;
        key[j] = 0;
        if (i < end) { key[j] = l1_buf[i]; valid |= 1u << j; }
    }
    if (padded) {                                        // segment padding (looked at only after all 16 loads are in flight)
#pragma unroll
        for (int j = 0; j < PART_ITEMS; ++j) if (key[j] == EMPTY) valid &= ~(1u << j);
    }
    return valid;
}

// The exact edition: one workgroup per level-1 bucket: histogram by digit, scan, scatter.  off2[r] = start of region r's run.
template <int HB>
__global__ void __launch_bounds__(PART_BLOCK)
k_p2(PartGeom g, const uint64_t* __restrict__ l1_off, const uint64_t* __restrict__ l1_buf, uint32_t* __restrict__ l2_lo, void* __restrict__ l2_hi,
     uint64_t* __restrict__ off2, uint64_t seg_slots, uint64_t* __restrict__ bend /* end of bucket b1's last run (segmented layout) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    PartLds& L = *reinterpret_cast<PartLds*>(lds_raw);
    const uint32_t tid = threadIdx.x;
    for (uint32_t b1 = blockIdx.x; b1 < g.P1; b1 += gridDim.x) {
        uint64_t beg, end;
        l1_bucket_range(l1_off, seg_slots, b1, beg, end);
        const uint64_t base1 = place_base1(b1, g.pl.n, g.pl.p1);
        lds_barrier();
        // pass A histogram in 64 bits (a heavy-hitter k-mer may put more than 2^32 items of a round into one region):
        // the cursor array is free until the scan, so it doubles as the histogram
        unsigned long long* h64 = reinterpret_cast<unsigned long long*>(L.cursor);
        if (tid < MAX_PARTS) h64[tid] = 0;
        lds_barrier();
        for (uint64_t i0 = beg; i0 < end; i0 += (uint64_t)8 * PART_BLOCK) {                   // 8 coalesced loads in flight per lane
            uint64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint64_t i = i0 + (uint64_t)u * PART_BLOCK + tid; v[u] = i < end ? l1_buf[i] : EMPTY; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] != EMPTY) atomicAdd(&h64[place_digit2(place_stage2(place_stage1(v[u], g.pl) - base1, g.pl), g.pl)], 1ULL);
        }
        lds_barrier();
        const uint64_t mine = tid < g.P2 ? h64[tid] : 0;
        const uint64_t excl = block_exclusive_scan64(mine, reinterpret_cast<uint64_t*>(L.staging));
        if (tid < g.P2) {
            L.cursor[tid] = beg + excl;
            off2[(uint64_t)b1 * g.P2 + tid] = beg + excl;
        }
        if (b1 == g.P1 - 1 && tid == 0) off2[(uint64_t)g.P1 * g.P2] = end;
        if (bend && tid == g.P2 - 1) bend[b1] = beg + excl + mine;                             // the runs of a bucket stop short of the next bucket (padding)
        for (uint64_t tbeg = beg; tbeg < end; tbeg += TILE_ITEMS) {                            // pass B
            uint64_t key[PART_ITEMS];
            const uint32_t valid = p2_tile_load(l1_buf, tbeg, end, seg_slots != 0, key);
            lds_barrier();
            scatter_tile2<HB, false>(L, g, base1, key, valid, l2_lo, l2_hi, nullptr, nullptr, nullptr, 0);
        }
    }
}

// ---- level 2 without the histogram pass ----
// The exact k_p2 reads every bucket twice: once to size the P2 sub-runs, once to scatter.  With a uniform hash the sizes
// are known in advance up to noise (n_b / P2 k-mers per region, sigma = sqrt of that), so the fast edition gives every
// region the same capacity -- mean + 1/16 + 16 -- scatters in ONE pass and reports what it really wrote (cnt2).  K-mers
// beyond a region's capacity (heavy hitters, or a very unlucky region) go to a small overflow list that the host
// inserts through the direct path; if even that list overflows, the host redoes the round with the exact kernel
// (the level-1 buffer is only read here) and stays exact for the rest of the call.
__device__ __host__ __forceinline__ uint64_t p2_out_base(uint64_t beg, uint32_t b1, uint32_t P2) { return beg + (beg >> 4) + (uint64_t)b1 * P2 * 16; }
__device__ __host__ __forceinline__ uint64_t p2_region_cap(uint64_t n_b, uint32_t P2) { return (n_b + (n_b >> 4)) / P2 + 16; }

template <int HB>
__global__ void __launch_bounds__(PART_BLOCK)
k_p2_fast(PartGeom g, const uint64_t* __restrict__ l1_off, const uint64_t* __restrict__ l1_buf, uint32_t* __restrict__ l2_lo, void* __restrict__ l2_hi,
          uint64_t* __restrict__ off2, uint32_t* __restrict__ cnt2, uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n,
          uint64_t ovf_cap, uint64_t seg_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    PartLds& L = *reinterpret_cast<PartLds*>(lds_raw);
    uint64_t* lim = reinterpret_cast<uint64_t*>(L.code);                   // code / bad are level-1 only: 8 KB for the run limits
    const uint32_t tid = threadIdx.x;
    for (uint32_t b1 = blockIdx.x; b1 < g.P1; b1 += gridDim.x) {
        uint64_t beg, end;
        l1_bucket_range(l1_off, seg_slots, b1, beg, end);
        const uint64_t cap = p2_region_cap(end - beg, g.P2), obase = p2_out_base(beg, b1, g.P2);
        const uint64_t base1 = place_base1(b1, g.pl.n, g.pl.p1);
        lds_barrier();
        if (tid < g.P2) {
            const uint64_t start = obase + (uint64_t)tid * cap;
            L.cursor[tid] = start;
            lim[tid] = start + cap;
            off2[(uint64_t)b1 * g.P2 + tid] = start;
        }
        for (uint64_t tbeg = beg; tbeg < end; tbeg += TILE_ITEMS) {
            uint64_t key[PART_ITEMS];
            const uint32_t valid = p2_tile_load(l1_buf, tbeg, end, seg_slots != 0, key);
            lds_barrier();
            scatter_tile2<HB, true>(L, g, base1, key, valid, l2_lo, l2_hi, lim, ovf_buf, ovf_n, ovf_cap);
        }
        lds_barrier();
        if (tid < g.P2) cnt2[(uint64_t)b1 * g.P2 + tid] = (uint32_t)(L.cursor[tid] - (obase + (uint64_t)tid * cap));
    }
}

// ---- level 3: apply a region's run to the region, in LDS ----
// LDS: keys[S] (u64) | counts[S] (u32).  Insert = LDS CAS claim + LDS add (same protocol as table_inc, minus the HBM).
// SPT = slots per lane held in registers while a region is prefetched (region_slots <= SPT * BLOCK).
// Software pipeline: while region r's run is applied in LDS, region r' (the workgroup's next one) is already on its way
// from HBM into registers, and r's write-back drains behind it -- the CU's memory pipe stays busy through the LDS phase.
template <int BLOCK, int SPT, int BATCH = 4, bool TEST_SPILL = false /* honours spill_mod: instantiated for the test suite only */>
__global__ void __launch_bounds__(BLOCK)
k_p3_apply(DevTable t, PartGeom g, const uint64_t* __restrict__ off2, const uint32_t* __restrict__ l2_lo, const void* __restrict__ l2_hi,
           uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n, uint32_t spill_mod,
           const uint32_t* __restrict__ cnt2 /* run lengths when k_p2_fast laid the runs out; null: off2[r + 1] ends run r */,
           const uint64_t* __restrict__ bend /* exact level 2 over a chunked level 1: where the last run of each bucket ends */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(lds_raw);
    uint32_t* rc = reinterpret_cast<uint32_t*>(lds_raw + (size_t)g.S * 8);
    const uint32_t tid = threadIdx.x, S = g.S;
    uint32_t new_distinct = 0;
    uint64_t kk[SPT]; uint32_t cc[SPT];

    auto run_end = [&](uint32_t r) -> uint64_t {                  // exact layouts: run r ends where the next one starts
        if (bend && (r + 1) % g.P2 == 0) return bend[r / g.P2];
        return off2[r + 1];
    };
    auto next_region = [&](uint32_t from) {                       // first region >= from (stride gridDim) that received k-mers
        uint32_t r = from;
        while (r < g.R && (cnt2 ? cnt2[r] == 0 : off2[r] == run_end(r))) r += gridDim.x;
        return r;
    };
    auto prefetch = [&](uint32_t r) {
        const uint64_t base = (uint64_t)r * S;
#pragma unroll
        for (int u = 0; u < SPT; ++u) { const uint32_t i = u * BLOCK + tid; kk[u] = i < S ? t.keys[base + i] : 0; cc[u] = i < S ? t.counts[base + i] : 0; }
    };

    uint32_t r = next_region(blockIdx.x);
    if (r < g.R) prefetch(r);
    while (r < g.R) {
        const uint64_t beg = off2[r], end = cnt2 ? beg + cnt2[r] : run_end(r);
        const uint64_t base = (uint64_t)r * S;
        // an item is the remainder of its k-mer's placement hash; the region supplies the digits (kg_device.hpp "placement")
        const uint64_t base1 = place_base1(r >> g.l2, g.pl.n, g.pl.p1), d2_hi = g.pl.rb < 64 ? (uint64_t)(r & (g.P2 - 1)) << g.pl.rb : 0ULL;
        auto item = [&](uint64_t i) -> unsigned long long { return place_key(base1, d2_hi | l2_load_any(g.hb, l2_lo, l2_hi, i), g.pl); };
#pragma unroll
        for (int u = 0; u < SPT; ++u) { const uint32_t i = u * BLOCK + tid; if (i < S) { rk[i] = kk[u]; rc[i] = cc[u]; } }
        lds_barrier();
        const uint32_t rn = next_region(r + gridDim.x);
        bool prefetched = false;
        // Each lane walks ITS k-mers of the batch on its own: a lane that has placed one k-mer starts probing for its next
        // while its neighbours are still on longer probe chains.  (With a per-k-mer loop the wave waits for the longest of
        // 64 chains for every k-mer -- about 10 probes at load 0.6 -- and this LDS-latency-bound loop ran at a fifth of the
        // speed; the lane-independent walk waits once, for the largest SUM of BATCH chains.)  The next batch is already on
        // its way from HBM.
        unsigned long long cur[BATCH], nxt[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) { const uint64_t i = beg + (uint64_t)u * BLOCK + tid; cur[u] = i < end ? item(i) : EMPTY; }
        for (uint64_t i0 = beg; i0 < end; i0 += (uint64_t)BATCH * BLOCK) {
          const uint64_t i1 = i0 + (uint64_t)BATCH * BLOCK;
#pragma unroll
          for (int u = 0; u < BATCH; ++u) { const uint64_t i = i1 + (uint64_t)u * BLOCK + tid; nxt[u] = i < end ? item(i) : EMPTY; }
          if (!prefetched) { if (rn < g.R) prefetch(rn); prefetched = true; }     // issued AFTER the first batches: their wait does not cover these
          uint32_t nv = 0;                                                          // this lane's k-mers in the batch (EMPTY only pads the tail)
#pragma unroll
          for (int u = 0; u < BATCH; ++u) nv += cur[u] != EMPTY;
          uint32_t u = 0, slot = 0, probes = 0;
          unsigned long long key = EMPTY;
          auto start = [&]() {                                                      // load k-mer u into the walk state
              while (u < nv) {
                  key = cur[0];
#pragma unroll
                  for (int q = 1; q < BATCH; ++q) key = u == (uint32_t)q ? cur[q] : key;
                  slot = home_offset(key, t);
                  probes = 0;
                  if (!(TEST_SPILL && spill_mod && __umulhi((uint32_t)(mix64(key) >> 32), spill_mod) == 0)) break;   // test hook: 1 k-mer in spill_mod takes the spill path
                  spill[atomicAdd(spill_n, 1ULL)] = key;
                  ++u;
              }
          };
          start();
          while (__any(u < nv)) {
              if (u < nv) {
                  unsigned long long c0 = rk[slot];
                  if (c0 == EMPTY) {
                      c0 = atomicCAS(&rk[slot], (unsigned long long)EMPTY, key);
                      if (c0 == EMPTY) { ++new_distinct; c0 = key; }
                  }
                  bool fin = false;
                  if (c0 == key) {
                      // LDS returning add: a 32-bit wrap is seen right here and chained into the side table, so a round may
                      // carry any number of copies of one k-mer and needs no host-side overflow guard
                      if (atomicAdd(&rc[slot], 1u) == 0xFFFFFFFFu) ovf_add(t, key, 1ULL << 32);
                      fin = true;
                  } else {
                      slot = slot + 1 == S ? 0 : slot + 1;
                      if (++probes == S) { spill[atomicAdd(spill_n, 1ULL)] = key; fin = true; }      // region full: direct path later
                  }
                  if (fin) { ++u; start(); }
              }
          }
#pragma unroll
          for (int q = 0; q < BATCH; ++q) cur[q] = nxt[q];
        }
        lds_barrier();
        for (uint32_t i = tid; i < S; i += BLOCK) { t.keys[base + i] = rk[i]; t.counts[base + i] = rc[i]; }
        lds_barrier();                                          // LDS is overwritten with the next region at the loop top
        r = rn;
    }
    flush_distinct(t, new_distinct);
}

// ---- level 3, second edition: the walk as straight-line batches ----
// k_p3_apply's walk is bound by dependent LDS round trips, not by LDS or VALU throughput (profiles/r01_partitioned_sq_counters.txt:
// waves parked 67 % of their cycles, LDS array 15 % busy): every lane runs one probe chain at a time inside a divergent loop.
// Here a wave takes U k-mers per lane and runs NR probe rounds over all of them in straight-line code: U independent
// ds_read_b64 in flight per wave and one wait per round; a k-mer whose slot holds its key gets a NO-RETURN ds_add and is done.
// What is left after NR rounds -- k-mers that met an EMPTY slot (a new key: needs the CAS claim) or a chain longer than NR --
// goes to a small per-wave queue in LDS (key + slot + remaining probe budget), which the wave drains 64 entries at a time with
// the dependent claim/add loop: dense, and only for the minority that needs it.  Waves never meet at a barrier inside a run.
// No-return adds cannot report a 32-bit wrap, so none may happen: before a walk, counters >= 2^31 give 2^31 to the side table,
// and a walk covers fewer than 2^31 k-mers (a longer run -- one region, one round, exact level 2 only -- is walked in segments).
// Region fill and write-back move 16 bytes per lane and instruction (8- and 4-byte stores were store-issue-bound).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a native vector: stays in registers where HIP's uint4 struct went to scratch
constexpr int AP2_QCAP = 256;                                 // straggler queue entries per wave (12 bytes each)
constexpr int AP2_LANE_PROBES = 12;                           // probes a queue entry gets from its own lane before the wave takes it over
constexpr uint64_t AP2_SEGMENT = 0x7FF00000ULL;               // k-mers per walk: < 2^31

template <int BLOCK, int KP /* 16-byte key loads per lane that cover a region */, int U, int NR, int HB /* high bytes of an item */, bool STAMP = false, bool INLINE_CLAIM = false, bool DYN = true>
__global__ void __launch_bounds__(BLOCK)
k_p3_apply2(DevTable t, PartGeom g, const uint64_t* __restrict__ off2, const uint32_t* __restrict__ l2_lo, const void* __restrict__ l2_hi,
            uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n,
            const uint32_t* __restrict__ cnt2, const uint64_t* __restrict__ bend, unsigned long long* __restrict__ stamps = nullptr) {
    // STAMP: cycle stamps of wave 0 (tools/ab_apply.sh): [0] fill + sweep, [1] chunk loads + hash, [2] probe rounds, [3] queue push + drains,
    // [4] wait for the other waves, [5] write-back, [6] regions
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NW = BLOCK / 64, CP = (KP + 1) / 2;
    constexpr uint32_t CH = 64 * U;
    const uint32_t S = g.S;                                   // S % 4 == 0 (host-checked): every region is 16-byte aligned in both arrays
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(lds_raw);
    uint32_t* rc = reinterpret_cast<uint32_t*>(lds_raw + (size_t)S * 8);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* wqk = reinterpret_cast<unsigned long long*>(lds_raw + (size_t)S * 12) + (size_t)wave * AP2_QCAP;
    uint32_t* wqs = reinterpret_cast<uint32_t*>(lds_raw + (size_t)S * 12 + (size_t)NW * AP2_QCAP * 8) + (size_t)wave * AP2_QCAP;
    uint32_t new_distinct = 0;
    u32x4 kq[KP], cq[CP];
    __shared__ unsigned long long s_next_chunk;               // chunks of the run are handed out to the waves as they come free

    auto run_end = [&](uint32_t r) -> uint64_t {
        if (bend && (r + 1) % g.P2 == 0) return bend[r / g.P2];
        return off2[r + 1];
    };
    auto next_region = [&](uint32_t from) {
        uint32_t r = from;
        while (r < g.R && (cnt2 ? cnt2[r] == 0 : off2[r] == run_end(r))) r += gridDim.x;
        return r;
    };
    auto prefetch = [&](uint32_t r) {
        const uint64_t base = (uint64_t)r * S;
#pragma unroll
        for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; kq[u] = *reinterpret_cast<const u32x4*>(t.keys + base + (i < S ? i : 0)); }    // clamped, unconditional: stays in registers
#pragma unroll
        for (int u = 0; u < CP; ++u) { const uint32_t i = (u * BLOCK + tid) * 4; cq[u] = *reinterpret_cast<const u32x4*>(t.counts + base + (i < S ? i : 0)); }
    };

    uint32_t r = next_region(blockIdx.x);
    if (r < g.R) prefetch(r);
    while (r < g.R) {
        const uint64_t beg = off2[r], end = cnt2 ? beg + cnt2[r] : run_end(r);
        const uint64_t base = (uint64_t)r * S;
        const unsigned long long t_top = now();
        // ---- fill: registers -> LDS ----
#pragma unroll
        for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; if (i < S) *reinterpret_cast<u32x4*>(rk + i) = kq[u]; }
#pragma unroll
        for (int u = 0; u < CP; ++u) { const uint32_t i = (u * BLOCK + tid) * 4; if (i < S) *reinterpret_cast<u32x4*>(rc + i) = cq[u]; }
        const uint32_t rn = next_region(r + gridDim.x);
        // an item is the remainder of its k-mer's placement hash; the region supplies the digits (kg_device.hpp "placement"):
        // the home slot comes straight from the remainder, the k-mer (what the slots hold) through the inverse hash
        const uint64_t base1 = place_base1(r >> g.l2, g.pl.n, g.pl.p1), d2_hi = g.pl.rb < 64 ? (uint64_t)(r & (g.P2 - 1)) << g.pl.rb : 0ULL;

        for (uint64_t sbeg = beg; sbeg < end; sbeg += AP2_SEGMENT) {        // one segment, normally
            const uint64_t n_run = (end - sbeg < AP2_SEGMENT ? end - sbeg : AP2_SEGMENT);
            if (tid == 0) s_next_chunk = NW;                  // chunks 0 .. NW-1 are the waves' first ones
            lds_barrier();
            // counters that could wrap during this walk hand 2^31 to the side table (each lane looks at the quads it filled)
#pragma unroll 1
            for (int u = 0; u < CP; ++u) {
                const uint32_t i = (u * BLOCK + tid) * 4;
                if (i >= S) break;
                const u32x4 c = *reinterpret_cast<const u32x4*>(rc + i);
                if (!((c.x | c.y | c.z | c.w) & 0x80000000u)) continue;
#pragma unroll 1
                for (uint32_t j = 0; j < 4; ++j)
                    if (rc[i + j] & 0x80000000u) { rc[i + j] -= 0x80000000u; ovf_add(t, rk[i + j], 0x80000000ULL); }
            }
            lds_barrier();
            st[0] += now() - t_top;

            // ---- the walk ----
            uint32_t q_n = 0;                                     // entries in this wave's queue (wave-uniform)
            auto add1 = [&](uint32_t slot) { (void)__hip_atomic_fetch_add(&rc[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            // One pass over up to 64 queue entries (taken from the tail).  Phase 1, one entry per lane: dependent probes with the
            // claim, while at least 8 lanes are busy and for at most AP2_LANE_PROBES probes.  Phase 2: what is left is on a long
            // chain (the longest of a region at load 0.6 runs to ~80 slots, and a lane walks it one LDS round trip per slot -- that
            // lane made the whole workgroup wait at the barrier): the WAVE finishes such a k-mer, 64 consecutive slots per read.
            auto drain_pass = [&](bool fin /* nothing will follow: leave no entry behind */) {
                const uint32_t take = q_n < 64 ? q_n : 64;
                q_n -= take;
                bool live = lane < take;
                unsigned long long key = EMPTY; uint32_t slot = 0, budget = 0;
                if (live) { key = wqk[q_n + lane]; const uint32_t s = wqs[q_n + lane]; slot = s & 0xFFFF; budget = s >> 16; }
#pragma unroll 1
                for (int rr = 0; rr < AP2_LANE_PROBES; ++rr) {
                    const int busy = __popcll(__ballot(live));
                    if (busy == 0 || (busy < 8 && (fin || rr >= 4))) break;
                    if (live) {
                        unsigned long long c0 = rk[slot];
                        if (c0 == EMPTY) {
                            c0 = atomicCAS(&rk[slot], (unsigned long long)EMPTY, key);
                            if (c0 == EMPTY) { ++new_distinct; c0 = key; }
                        }
                        if (c0 == key) { add1(slot); live = false; }
                        else {
                            slot = slot + 1 == S ? 0 : slot + 1;
                            if (--budget == 0) { spill[atomicAdd(spill_n, 1ULL)] = key; live = false; }     // region full: direct path later
                        }
                    }
                }
                // the wave takes over what has had its AP2_LANE_PROBES (all that is left, when nothing follows); the rest goes back
                const bool lng = live && (fin || S - budget >= (uint32_t)(AP2_LANE_PROBES + NR));
                {
                    const bool back = live && !lng;
                    const unsigned long long m = __ballot(back);
                    if (m) {
                        const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                        if (back) { wqk[at] = key; wqs[at] = slot | (budget << 16); }
                        q_n += (uint32_t)__popcll(m);
                    }
                }
                unsigned long long todo = __ballot(lng);
#pragma unroll 1
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const unsigned long long ck = __shfl(key, src, 64);                      // wave-uniform from here on
                    uint32_t cs = __shfl(slot, src, 64);
                    int cb = (int)__shfl(budget, src, 64);
#pragma unroll 1
                    for (;;) {
                        uint32_t idx = cs + lane; if (idx >= S) idx -= S;                      // S >= 64 on this path (host-checked)
                        const unsigned long long c0 = rk[idx];
                        const unsigned long long mk = __ballot(c0 == ck), me = __ballot(c0 == EMPTY);
                        if (!(mk | me)) {                                                      // 64 foreign keys
                            cb -= 64; cs = cs + 64 >= S ? cs + 64 - S : cs + 64;
                            if (cb <= 0) { if (lane == 0) spill[atomicAdd(spill_n, 1ULL)] = ck; break; }
                            continue;
                        }
                        const int first = __ffsll((long long)(mk | me)) - 1;
                        if (first >= cb) { if (lane == 0) spill[atomicAdd(spill_n, 1ULL)] = ck; break; }   // beyond the region's last unprobed slot
                        unsigned long long got = ck;                                           // what the slot holds after this step
                        if (!((mk >> first) & 1)) {                                            // EMPTY comes first: claim it
                            unsigned long long old = EMPTY;
                            if ((int)lane == first) old = atomicCAS(&rk[idx], (unsigned long long)EMPTY, ck);
                            old = __shfl(old, first, 64);
                            if (old == EMPTY) { if ((int)lane == first) ++new_distinct; }
                            else got = old;
                        }
                        if (got == ck) { if ((int)lane == first) add1(idx); break; }
                        cb -= first; cs = cs + first >= S ? cs + first - S : cs + first;       // someone else's key landed there: go on from that slot
                    }
                }
            };

            const uint64_t n_chunks = (n_run + CH - 1) / CH;
            unsigned long long cur[U], nxt[U];                    // remainders (then k-mers) of this chunk, remainders of the next
            uint32_t cur_n = 0, nxt_n = 0;                        // bit u: item u exists
#pragma unroll
            for (int u = 0; u < U; ++u) { const uint64_t i = (uint64_t)wave * CH + (uint64_t)u * 64 + lane; cur[u] = l2_load<HB>(l2_lo, l2_hi, sbeg + (i < n_run ? i : 0)); cur_n |= i < n_run ? 1u << u : 0u; }
            // (static round-robin left the workgroup waiting ~12 K cycles per region for its slowest wave: the drains vary)
            auto grab = [&]() -> uint64_t {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(&s_next_chunk, 1ULL);
                return __shfl(v, 0, 64);
            };
            for (uint64_t c = wave; c < n_chunks;) {
                const unsigned long long t_a = now();
                const uint64_t c_next = DYN ? grab() : c + NW;
#pragma unroll
                for (int u = 0; u < U; ++u) {                 // next chunk: in flight behind this one (unconditional loads from a clamped index: a load inside a branch is waited for at the end of the branch)
                    const uint64_t i = c_next * CH + (uint64_t)u * 64 + lane;
                    nxt[u] = l2_load<HB>(l2_lo, l2_hi, sbeg + (i < n_run ? i : 0));
                    nxt_n |= i < n_run ? 1u << u : 0u;
                }
                uint32_t slot[U];
                bool pend[U];                                     // k-mer u still to be placed (lane masks in SGPRs)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    pend[u] = (cur_n >> u) & 1;
                    slot[u] = place_offset(cur[u], g.pl, S);
                    cur[u] = place_key(base1, d2_hi | cur[u], g.pl);
                }
                const unsigned long long t_b = now();
                // Probe rounds: U reads in flight, one wait; a match adds 1 and is done, a foreign key moves on, an EMPTY slot is
                // claimed.  Lanes that are done take no part in the LDS operations.
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    unsigned long long seen[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { seen[u] = EMPTY; if (pend[u]) seen[u] = rk[slot[u]]; }
                    bool claim[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool hit = pend[u] && seen[u] == cur[u];
                        if (hit) add1(slot[u]);
                        pend[u] = pend[u] && !hit;
                        claim[u] = pend[u] && seen[u] == EMPTY;
                        const uint32_t nx = slot[u] + 1 == S ? 0 : slot[u] + 1;
                        slot[u] = (pend[u] && !claim[u]) ? nx : slot[u];
                    }
                    // INLINE_CLAIM: new keys claimed right here, U CAS in flight, instead of through the queue.  Measured (same box,
                    // bench config): 217 ms against 196 without -- the extra dependent round trip per probe round costs more than
                    // the queue traffic it saves (a first round on an empty table gains, every later round loses).  Off.
                    bool any_claim = false;
#pragma unroll
                    for (int u = 0; u < U; ++u) any_claim = any_claim || claim[u];
                    if (INLINE_CLAIM && __any(any_claim)) {
                        unsigned long long got[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { got[u] = 0; if (claim[u]) got[u] = atomicCAS(&rk[slot[u]], (unsigned long long)EMPTY, cur[u]); }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (claim[u]) {
                                if (got[u] == EMPTY) ++new_distinct;
                                const bool mine = got[u] == EMPTY || got[u] == cur[u];
                                if (mine) add1(slot[u]);
                                pend[u] = !mine;
                                if (!mine) slot[u] = slot[u] + 1 == S ? 0 : slot[u] + 1;    // someone else's key landed there
                            }
                        }
                    }
                }
                if (STAMP) { __builtin_amdgcn_s_waitcnt(0); }
                const unsigned long long t_c = now();
                // survivors -> queue (q_n <= 64 here).  Normal case: one wave-wide prefix sum; a chunk with more survivors than the
                // queue holds (a nearly empty table: every k-mer is new) goes in one k-mer column at a time.
                uint32_t mine = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) mine += pend[u] ? 1u : 0u;
                uint32_t tot = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(tot, d, 64); if (lane >= (uint32_t)d) tot += o; }
                const uint32_t total = __shfl(tot, 63, 64);
                if (total <= AP2_QCAP - 64) {
                    uint32_t at = q_n + tot - mine;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (pend[u]) { wqk[at] = cur[u]; wqs[at] = slot[u] | ((S - NR) << 16); ++at; }
                    q_n += total;
                } else {
                    uint32_t pm = 0;
#pragma unroll
                    for (int u = 0; u < U; ++u) pm |= pend[u] ? 1u << u : 0u;
#pragma unroll 1
                    for (int it = 0; it < U; ++it) {
                        const bool p = pm & 1;
                        const unsigned long long m = __ballot(p);
                        if (m) {
                            while (q_n > AP2_QCAP - 64) drain_pass(false);
                            const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            if (p) { wqk[at] = cur[0]; wqs[at] = slot[0] | ((S - NR) << 16); }
                            q_n += (uint32_t)__popcll(m);
                        }
                        pm >>= 1;
#pragma unroll
                        for (int u = 0; u + 1 < U; ++u) { cur[u] = cur[u + 1]; slot[u] = slot[u + 1]; }
                    }
                }
                const bool last = c_next >= n_chunks;                     // the wave's last chunk empties the queue
                while (q_n > (last ? 0u : 64u)) drain_pass(last);
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
                cur_n = nxt_n; nxt_n = 0;
                c = c_next;
                if (STAMP) { const unsigned long long t_d = now(); st[1] += t_b - t_a; st[2] += t_c - t_b; st[3] += t_d - t_c; }
            }
        }
        if (rn < g.R) prefetch(rn);                           // in flight behind the write-back
        const unsigned long long t_w = now();

        // ---- write-back: LDS -> HBM, 16 bytes per lane and store ----
        lds_barrier();
        const unsigned long long t_x = now();
#pragma unroll
        for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; if (i < S) *reinterpret_cast<u32x4*>(t.keys + base + i) = *reinterpret_cast<const u32x4*>(rk + i); }
#pragma unroll
        for (int u = 0; u < CP; ++u) { const uint32_t i = (u * BLOCK + tid) * 4; if (i < S) *reinterpret_cast<u32x4*>(t.counts + base + i) = *reinterpret_cast<const u32x4*>(rc + i); }
        lds_barrier();
        if (STAMP) { st[4] += t_x - t_w; st[5] += now() - t_x; st[6] += 1; }
        r = rn;
    }
    if (STAMP && tid == 0 && stamps) for (int i = 0; i < 7; ++i) atomicAdd(&stamps[i], st[i]);
    flush_distinct(t, new_distinct);
}

// spilled k-mers (count 1 each) through the direct path.  Checked adds (table_add sees a 32-bit wrap itself): these lists
// are short, and the unchecked table_inc would oblige the host to sweep the whole table first (katgpu.hip: maybe_sweep).
__global__ void __launch_bounds__(256)
k_insert_keys(DevTable t, const uint64_t* __restrict__ keys, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) table_add(t, keys[i], 1ULL, new_distinct);
    flush_distinct(t, new_distinct);
}

}  // namespace kg
