// kg_partition.hpp -- the partitioned counter: K1 without a DRAM/L2 atomic per k-mer.
//
// Measured on MI355X (tools/ubench_count.hip, profiles/): extracting + canonicalising + hashing k-mers runs at
// ~490 G k-mers/s, random 8-byte loads at ~55 G/s, but device-scope atomic adds saturate the L2 atomic units at
// ~22 G/s whatever the table size -- so the direct kernel (k_count: one atomic per instance) tops out near 15 G k-mers/s.
// This path removes the per-instance global atomic.  The table is made of regions of `region_slots` slots with
// region-local probing (kg_device.hpp: Probe); a round of the partitioned counter
//   P1  radix-partitions the round's k-mers by the high part of their region index into P1 buckets  (4 + HB1 B out / k-mer)
//   P2  splits every bucket by the low part of the region index -> one contiguous run per region     (4 + HB1 B in, 4 + HB B out)
//   P3  loads a region into LDS, applies its run with LDS atomics, writes the region back            (4 + HB B in + 16 B/slot packed, 24 KV12)
// An item is not the k-mer but what its position does not already say (kg_device.hpp "placement": the map k-mer -> (digit 1, digit 2,
// remainder) is one to one and each level peels its digit off): level 1 writes the k-mer's low n1 = 2k - log2(p1) bits, level 2 its
// low rb = n1 - log2(p2) bits, each as the low 32 bits + HB = 0, 1, 2 or 4 bytes of high bits (45 and 35 bits at the bench size: 6
// and 5 bytes per k-mer instead of 8 and 8).
// so HBM sees only streaming traffic.  Level 1 is exact two-pass (histogram, scan, scatter) with per-workgroup running
// cursors held in LDS: no global atomic per k-mer, deterministic placement.  Level 2 normally runs in ONE pass
// (k_p2_fast: equal-capacity runs sized from the uniform hash, an overflow list for what does not fit, the exact
// histogram/scan/scatter kernel k_p2 as the fall back).  A region that fills up in P3 (more distinct k-mers than slots)
// spills the k-mer to a list (held in the then-free P1 buffer, so it can never overflow) that is inserted through the
// direct path after a regrow.
#pragma once
#include "kg_kernels.hpp"
#include "kg_l1_lean.hpp"

#include <cstddef>
#include <type_traits>

namespace kg {

constexpr int PART_BLOCK = 1024;                              // 16 waves, one workgroup per CU
constexpr int PART_ITEMS = 16;                                // k-mers per lane per tile
constexpr int TILE_ITEMS = PART_BLOCK * PART_ITEMS;           // 16384
constexpr int L1_TILE_BYTES = TILE_ITEMS;                     // bytes staged per tile (16 per lane)
constexpr int L1_TILE_STARTS = L1_TILE_BYTES - CHUNK_OVERLAP; // 16352 window starts per tile
constexpr int L1_LANES_WITH_STARTS = L1_TILE_STARTS / PART_ITEMS;   // 1022

// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a workgroup-scope release fence, which on gfx950
// lowers to s_waitcnt vmcnt(0): every barrier placed after a run of global STORES (copy-out, region write-back) would
// stall the whole workgroup until those stores are acknowledged by L2.  Nothing in these kernels hands global data from
// one lane to another inside a launch, so only LDS traffic has to be complete here.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// a k-mer that found no room in its region -> the pass's spill list
__device__ __forceinline__ void spill_put(uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n, uint64_t cap, uint64_t key) {
    const unsigned long long at = atomicAdd(spill_n, 1ULL);
    if (at < cap) spill[at] = key;
}

struct PartGeom {
    uint32_t R, S;     // regions, slots per region (the table's)
    uint32_t P1, P2;   // region r = b1 * P2 + b2 (the table's p1, p2): level-1 bucket b1, level-2 bucket b2
    uint32_t b_lo, b_hi; // this launch's share of a round: level-1 buckets [b_lo, b_hi) = regions [b_lo * P2, b_hi * P2) (a round goes
                       // through level 2 + apply in as many passes as there are whole sets of n_CU buckets: the level-2 buffer holds one pass)
    uint32_t l2;       // P2 == 1 << l2
    uint32_t hb;       // bytes of a level-2 item beyond its low 32 bits: 0, 1, 2 or 4 (from pl.rb; "the level-2 buffer" below)
    uint32_t hb1;      // ... of a level-1 item (from pl.n1; "the level-1 buffer" below)
    uint64_t l1_stride;// segmented level 1: bytes from one bucket's first group to the next's in the level-1 buffer; 0: the exact layout
    uint64_t l1_real;  // segmented level 1: k-mers a bucket holds at most (workgroups x the segments' k-mer capacity; its slots -- group padding included -- are more): what level 2's runs and the spill list are sized by
    uint64_t spill_cap;// k-mers the apply's spill list holds (this pass's part of the level-1 buffer); what is beyond is counted, not written: the host fails the call
    uint32_t cbits;    // the table's (kg_device.hpp: packed slots); 0: KV12
    Place pl;          // the placement hash's bit budget for this table (kg_device.hpp)
};
// ---- the level-2 buffer ----
// Items of 4 + HB bytes (low word + HB = 0, 1, 2 or 4 bytes of high bits) in GROUPS of four: [4 low words | 4 high parts] =
// 16 + 4 HB contiguous bytes (20 at the bench size).  Level 2 stores, and the apply loads, a group with two instructions, and what
// a sub-bucket receives from one tile is ONE contiguous piece.  That is what the layout is for: the copy-out is bound by how many
// pieces (cache-line visits) reach the memory system and how long their acknowledgements take -- waves parked 75 % of their cycles
// (profiles/) -- not by bytes: plain per-item stores into a low-word stream and a high-part stream measured 301 ms for level 2 at
// the bench size, groups with the two streams 128 bytes apart 285, against 205 for 8-byte k-mers.  Runs start on group
// boundaries; the all-ones item is "no item" (padding), which is why HB leaves the remainder at least one spare code.  Groups are
// only 4-byte aligned (gfx950 takes 16-byte accesses at any dword).
__device__ __host__ __forceinline__ uint32_t l2_hi_bytes(uint32_t rb) { return rb <= 31 ? 0u : rb <= 39 ? 1u : rb <= 47 ? 2u : 4u; }   // rb = 64: no partition path (part_geometry)
template <int HB> struct L2Fmt {
    static constexpr int N = HB == 4 ? 12 : 16;              // k-mers per lane and tile (what the padded tile costs in LDS)
    static constexpr int TILE = N * 1024;
    static constexpr int NS = TILE + 3 * 1024;               // staged slots of a tile: every sub-bucket's run padded to whole groups
    static constexpr uint32_t GS = 16 + 4 * HB;              // bytes of a group
    static constexpr uint64_t NONE = HB == 4 ? ~0ULL : (1ULL << (32 + 8 * HB)) - 1;
};
__device__ __host__ __forceinline__ uint32_t l2_tile_items(uint32_t hb) { return hb == 4 ? 12 * 1024 : 16 * 1024; }
// ---- 5-byte items (HB = 1: the bench's tables) live in 64-BYTE BLOCKS OF TWELVE (round 6) ----
// block = [12 low words: 48 B | 12 high bytes | 4 B unused], 64-byte aligned; group g of the buffer (four items, still the unit of the
// readers: a lane of the apply takes one) is part g % 3 of block g / 3: its low words 16 bytes at 16 (g % 3), its high bytes the dword at
// 48 + 4 (g % 3).  Why: what a memory request LOOKS like decides what a store costs here (tools/ubench_l1_tlb.hip, ubench_l2_layout.hip, round
// 6).  A sub-bucket's piece of a tile as 20-byte groups at dword alignment (two store instructions per group, every piece ending in
// partial lines) moves level 2's pattern at 22.2 ms per launch; whole 64-byte blocks stored by ONE lane each (four instructions per line:
// round 5's "C") 19.0; the same blocks stored by a QUAD of lanes in one instruction -- sixteen bytes each, a whole line per request --
// 16.6, stores alone 7.4 against 13.0.  So the one-pass level 2 keeps an incomplete block per sub-bucket in LDS until it is full
// (scatter_tile2_blk) and everything else addresses groups through these two functions.
__device__ __host__ __forceinline__ constexpr bool l2_blocked(uint32_t hb) { return hb == 1; }
constexpr uint32_t L2_BLOCK_ITEMS = 12, L2_BLOCK_BYTES = 64;
template <int HB> __device__ __host__ __forceinline__ uint64_t l2_lo_at(uint64_t grp) {          // byte offset of group grp's four low words
    if (l2_blocked(HB)) { const uint64_t b = grp / 3; return b * L2_BLOCK_BYTES + (grp - 3 * b) * 16; }
    return grp * L2Fmt<HB>::GS;
}
template <int HB> __device__ __host__ __forceinline__ uint64_t l2_hi_at(uint64_t grp) {          // ... of its four high parts
    if (l2_blocked(HB)) { const uint64_t b = grp / 3; return b * L2_BLOCK_BYTES + 48 + (grp - 3 * b) * 4; }
    return grp * L2Fmt<HB>::GS + 16;
}
// bytes the level-2 buffer takes for `items` items (a multiple of 4) + the granule runs are aligned to, in items
__device__ __host__ __forceinline__ uint64_t l2_buffer_bytes(uint32_t hb, uint64_t items) { return l2_blocked(hb) ? (items / L2_BLOCK_ITEMS + 2) * L2_BLOCK_BYTES : items / 4 * (16 + 4 * hb); }
__device__ __host__ __forceinline__ double l2_bytes_per_item(uint32_t hb) { return l2_blocked(hb) ? (double)L2_BLOCK_BYTES / L2_BLOCK_ITEMS : 4.0 + hb; }
__device__ __host__ __forceinline__ uint32_t l2_run_align(uint32_t hb) { return l2_blocked(hb) ? L2_BLOCK_ITEMS : 4u; }
template <int HB> struct HiWord { typedef uint32_t type; };
template <> struct HiWord<1> { typedef uint8_t type; };
template <> struct HiWord<2> { typedef uint16_t type; };
// one item (slow paths: the exact level 2, the first-edition apply)
template <int HB>
__device__ __forceinline__ void l2_put(uint8_t* __restrict__ buf, uint64_t i, uint64_t rem) {
    reinterpret_cast<uint32_t*>(buf + l2_lo_at<HB>(i >> 2))[i & 3] = (uint32_t)rem;
    if (HB) reinterpret_cast<typename HiWord<HB>::type*>(buf + l2_hi_at<HB>(i >> 2))[i & 3] = (typename HiWord<HB>::type)(rem >> 32);
}
template <int HB>
__device__ __forceinline__ uint64_t l2_get(const uint8_t* __restrict__ buf, uint64_t i) {
    const uint32_t l = reinterpret_cast<const uint32_t*>(buf + l2_lo_at<HB>(i >> 2))[i & 3];
    const uint32_t h = HB ? (uint32_t)reinterpret_cast<const typename HiWord<HB>::type*>(buf + l2_hi_at<HB>(i >> 2))[i & 3] : 0u;
    return ((uint64_t)h << 32) | l;
}
// one group: four low words + four high parts (HB = 4: the high parts are a second 16-byte piece)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a native vector: stays in registers where HIP's uint4 struct went to scratch
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));           // in the level-2 buffer: dword-aligned
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
template <int HB> struct HiGroup { typedef uint32_t type; typedef uint32_t mem; };          // HB = 0: unused; HB = 1: four bytes
template <> struct HiGroup<2> { typedef u32x2 type; typedef u32x2_a4 mem; };
template <> struct HiGroup<4> { typedef u32x4 type; typedef u32x4_a4 mem; };
template <int HB>
__device__ __forceinline__ uint32_t hi_of_group(const typename HiGroup<HB>::type& h, int q);
template <> __device__ __forceinline__ uint32_t hi_of_group<0>(const uint32_t&, int) { return 0; }
template <> __device__ __forceinline__ uint32_t hi_of_group<1>(const uint32_t& h, int q) { return (h >> (8 * q)) & 0xFF; }
template <> __device__ __forceinline__ uint32_t hi_of_group<2>(const u32x2& h, int q) { return ((q < 2 ? h.x : h.y) >> (16 * (q & 1))) & 0xFFFF; }
template <> __device__ __forceinline__ uint32_t hi_of_group<4>(const u32x4& h, int q) { return q == 0 ? h.x : q == 1 ? h.y : q == 2 ? h.z : h.w; }
// a group in the PLAIN layout (16 + 4 HB contiguous bytes): the level-1 buffer's groups, and the level-2 buffer's unless its items are blocked
template <int HB>
__device__ __forceinline__ void l2_store_group(uint8_t* __restrict__ buf, uint64_t grp, const u32x4& lo, const typename HiGroup<HB>::type& hi) {
    uint8_t* p = buf + grp * L2Fmt<HB>::GS;
    *reinterpret_cast<u32x4_a4*>(p) = lo;
    if (HB) *reinterpret_cast<typename HiGroup<HB>::mem*>(p + 16) = hi;
}
// A run of the level-2 buffer as the apply kernels read it: groups counted from group g0 of the buffer (32-bit from there: a walk covers fewer
// than 2^31 k-mers).  Blocked items: g0 = 3 q0 + r0, group gi of the run is part (r0 + gi) % 3 of block q0 + (r0 + gi) / 3 -- a multiply-high.
template <int HB>
struct L2Run {
    const uint8_t* p;
    uint32_t r0;
    __device__ __forceinline__ L2Run(const uint8_t* __restrict__ buf, uint64_t g0) {
        if constexpr (l2_blocked(HB)) { const uint64_t q0 = g0 / 3; r0 = (uint32_t)(g0 - 3 * q0); p = buf + q0 * L2_BLOCK_BYTES; }
        else { r0 = 0; p = buf + g0 * L2Fmt<HB>::GS; }
    }
    __device__ __forceinline__ void load(uint32_t gi, u32x4& lo, typename HiGroup<HB>::type& hi) const {
        if constexpr (l2_blocked(HB)) {
            const uint32_t x = r0 + gi, blk = __umulhi(x, 0xAAAAAAABu) >> 1, part = x - 3 * blk;
            const uint8_t* q = p + ((uint64_t)blk << 6);
            lo = *reinterpret_cast<const u32x4*>(q + 16 * part);                    // (16-byte aligned: the buffer starts on a 64-byte boundary)
            hi = *reinterpret_cast<const uint32_t*>(q + 48 + 4 * part);
        } else {
            const uint8_t* q = p + (uint64_t)gi * L2Fmt<HB>::GS;
            lo = *reinterpret_cast<const u32x4_a4*>(q);
            if (HB) hi = *reinterpret_cast<const typename HiGroup<HB>::mem*>(q + 16);
        }
    }
};

// LDS carve of the exact level-2 kernel k_p2 (dynamic shared memory, 16-byte aligned base; the one-pass edition: P2FLds below)
template <int HB>
struct P2Lds {
    uint64_t cursor[MAX_PARTS];        // next free item of each sub-bucket's run (64 bits: a heavy hitter may put more than 2^32 items of a round into one run)
    uint32_t hist[MAX_PARTS];          // k-mers of the tile per sub-bucket
    uint32_t goff[MAX_PARTS];          // where the sub-bucket's k-mers of this tile are staged: first slot
    uint32_t wave_tot[32];
    uint32_t pad_[32];                 // (st_lo starts on a 16-byte boundary)
    uint32_t st_lo[L2Fmt<HB>::TILE];   // staged remainders, grouped by sub-bucket: low words ...
    typename HiWord<HB>::type st_hi[HB ? L2Fmt<HB>::TILE : 4];   // ... high parts
};
static_assert(sizeof(P2Lds<0>) <= 160 * 1024 && sizeof(P2Lds<1>) <= 160 * 1024 && sizeof(P2Lds<2>) <= 160 * 1024 && sizeof(P2Lds<4>) <= 160 * 1024, "LDS");

// exclusive scan of v over the first MAX_PARTS lanes of a 1024-thread block (lane b holds bucket b); returns the
// exclusive prefix, *total gets the grand total.  Two barriers inside.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    // the sixteen wave totals: lane w of every wave takes wave w's and the same scan gives every wave all the prefixes (a loop over the
    // waves with a comparison each kept sixteen lane masks alive across the kernel: scalar registers spilled into vector ones)
    const uint32_t sc = wave_inclusive_scan(lane < PART_BLOCK / 64 ? wave_tot[lane] : 0u);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    *total = lane_value(sc, PART_BLOCK / 64 - 1);
    const uint32_t prefix = wv ? lane_value(sc, (int)wv - 1) : 0u;
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

// ---- level 1 scan: offs[w][b] = start of workgroup w's run inside bucket b; l1_off[b] = start of bucket b; l1_off[P1] = items ----
static __global__ void __launch_bounds__(PART_BLOCK)
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

// ---- the level-1 buffer ----
// A level-1 item is r1 = the k-mer's low n1 bits (kg_device.hpp "placement": the bucket says the rest), kept like a level-2 item: the
// low 32 bits + HB1 = 0, 1 or 2 bytes of high bits, in groups of four = 16 + 4 HB1 contiguous bytes (24 at the bench size: 6 bytes
// per k-mer where the k-mer itself took 8; level 2 loads a group with two instructions); items of more than 48 bits (HB1 = 4) are plain
// 64-bit words, one store each.  Bucket b's groups start at byte
// l1_bucket_base(first item of b, b): the buffer keeps 8 bytes per item whatever the group size, because a pass's part of it, once
// its level 2 is through, is the spill list of that pass's apply -- 8-byte k-mers, as many as there were items in the worst case.
// The all-ones item is "no item" (segment padding): n1 < 8 (4 + HB1) wherever this path runs (part_geometry).
__device__ __host__ __forceinline__ uint64_t l1_bucket_base(uint64_t first_item, uint32_t b) { return 8 * first_item + 32ULL * b; }   // (+ 32 b: a bucket's last group may hold up to three items more than the bucket)
template <uint32_t HB1>
__device__ __forceinline__ void l1_put(uint8_t* __restrict__ bucket, uint64_t group_byte, uint32_t q, uint32_t lo, uint32_t hi) {
    uint8_t* grp = bucket + group_byte;
    if (HB1 == 4) { reinterpret_cast<uint64_t*>(grp)[q] = ((uint64_t)hi << 32) | lo; return; }      // 8-byte items: as they are, one store (a "group" is four of them)
    reinterpret_cast<uint32_t*>(grp)[q] = lo;
    if (HB1 == 1) grp[16 + q] = (uint8_t)hi;
    else if (HB1 == 2) reinterpret_cast<uint16_t*>(grp + 16)[q] = (uint16_t)hi;
}
__device__ __forceinline__ void l1_put_any(uint8_t* __restrict__ bucket, uint64_t group_byte, uint32_t q, uint32_t hb1, uint32_t lo, uint32_t hi) {
    if (hb1 == 0) l1_put<0>(bucket, group_byte, q, lo, hi);
    else if (hb1 == 1) l1_put<1>(bucket, group_byte, q, lo, hi);
    else if (hb1 == 2) l1_put<2>(bucket, group_byte, q, lo, hi);
    else l1_put<4>(bucket, group_byte, q, lo, hi);
}

// LEAN (k_p1v2_scatter<SEG, true>): the tile's reverse-complement stream is staged next to the codes (rcode), so that the copy-out reads
// the k-mer of the strand the ranking sweep chose instead of recomputing the canonical form; the segmented edition's cursors count
// inside a segment and fit 32 bits, which pays for rcode (three workgroups per CU: 3 x 1280-byte granules to spare).
// PB: bucket capacity of the per-bucket arrays (512 when the table has at most 512 level-1 digits -- the bench's tables --, else
// MAX_PARTS): with 512 the segmented editions stay at three workgroups per CU although their staged runs are padded to whole groups.
template <bool LEAN = false, bool SEG = false, int PB = MAX_PARTS>
struct P1LdsT {
    typedef typename std::conditional<SEG, uint32_t, uint64_t>::type cursor_t;
    cursor_t cursor[PB];                // next item of each bucket's run, counted from the bucket's first item (exact edition) / inside the segment (segmented)
    uint32_t hist[PB + (LEAN ? 64 : 0)];   // (LEAN: + one dump counter per lane of a wave, for the windows that hold no k-mer)
    uint32_t real[SEG ? PB : 1];        // SEG: k-mers in each bucket's segment so far (the cursor counts slots: k-mers + group padding)
    uint32_t off[PB + (LEAN ? 64 : 0)];     // (LEAN: the dump counters' "runs" all start at the dump slots behind pos[]: a window without a k-mer is staged like any other, there)
    uint32_t wave_tot[16];
    uint32_t code[P1_BLOCK + 2];
    uint16_t bad[P1_BLOCK + 2];        // (sixteen flags per word of codes; 16-bit entries: the kilobyte that keeps three workgroups on a CU beside the dump slots)
    uint32_t rcode[LEAN ? P1_BLOCK + 2 : 1];   // LEAN: word v = reverse complement of code word 511 - v (the tile read backwards on the other strand)
    uint32_t pad_[2];                   // (pos starts on a 16-byte boundary: the grouped copy-out reads it four entries at a time)
    uint32_t pos[P1_TILE_BYTES + (SEG ? 3 * PB : 0) + (LEAN ? 128 : 0)];  // per staged k-mer: bucket << 16 | tile position (LEAN: | where the copy-out reads the k-mer from, kg_l1_lean.hpp lean_entry_*); SEG: a bucket's run padded to whole groups of four; LEAN: + 128 dump slots (a dump counter ranks at most 8 lanes x 16 windows per tile)
};
static_assert(P1_BLOCK == (int)LEAN_BLOCK, "kg_l1_lean.hpp: the staged entries' arithmetic is written for this tile");
constexpr uint32_t P1_RCW = LEAN_RCW;     // LEAN: L.rcode lies this many words behind L.code (code | bad | rcode are one array to the copy-out: lean_entry_*)
constexpr uint32_t P1_PAD = 0xFFFF0000u;   // "no k-mer" in pos[]: the bucket field of an entry is at most 1023, so the top bit says it; read as an entry it is tile position 0

struct LaneWindow {                       // the 96-bit sliding window of kg_kernels.hpp's K1, as an object
    uint64_t hi, lo, m;
    uint32_t kshift, mshift;
    template <typename BadT>
    __device__ __forceinline__ void init(const uint32_t* code, const BadT* bad, uint32_t w, uint32_t k) {
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
    const uint32_t c0 = code[w], c1 = code[w + 1], c2 = code[w + 2];
    // the 64 bits that start 2 o bits into c0 : c1 : c2 -- two funnel shifts (v_alignbit_b32 shifts right by its amount mod 32)
    const uint32_t sh = (32 - 2 * o) & 31;
    const uint32_t h1 = o ? __builtin_amdgcn_alignbit(c0, c1, sh) : c0, h0 = o ? __builtin_amdgcn_alignbit(c1, c2, sh) : c1;
    const uint64_t hi = ((uint64_t)h1 << 32) | h0;
    return canon_if(hi >> (64 - 2 * k), k, canonical);
}

// The same in two steps, so that the load of the NEXT tile can be in flight while this one is worked on: `issue` is one
// unconditional 16-byte load (a lane whose window crosses the end of the stream loads from the start of it instead -- a load inside a
// branch is waited for at the end of the branch, which is what the one-step form below amounts to), `fix` looks at the result when it
// is needed and takes the byte-wise path for those (rare) lanes.
__device__ __forceinline__ u32x4 p1_tile_issue(const uint8_t* __restrict__ bases, uint64_t n, uint64_t tile_off) {
    const uint64_t off = tile_off + (uint64_t)threadIdx.x * PART_ITEMS;
    return *reinterpret_cast<const u32x4*>(bases + (off + PART_ITEMS <= n ? off : 0));
}
__device__ __forceinline__ void p1_tile_fix(const uint8_t* __restrict__ bases, uint64_t n, uint64_t tile_off, const u32x4& v, uint32_t (&w)[4]) {
    const uint64_t off = tile_off + (uint64_t)threadIdx.x * PART_ITEMS;
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    if (off + PART_ITEMS > n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { uint64_t i = off + q * 4 + b; x |= (i < n ? (uint32_t)bases[i] : (uint32_t)'N') << (8 * b); }
            w[q] = x;
        }
    }
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

template <bool LEAN, bool SEG, int PB>
__device__ __forceinline__ void p1_tile_stage(P1LdsT<LEAN, SEG, PB>& L, const uint32_t (&w)[4]) {
    const uint32_t tid = threadIdx.x;
    uint32_t code, bad;
    encode16(w, code, bad);
    L.code[tid] = code;
    L.bad[tid] = (uint16_t)bad;
    if (LEAN) L.rcode[P1_BLOCK - 1 - tid] = lean_revcomp16(code);
    if (tid < 2) { L.code[P1_BLOCK + tid] = 0; L.bad[P1_BLOCK + tid] = 0xFFFF; if (LEAN) L.rcode[P1_BLOCK + tid] = 0; }
    lds_barrier();
}

// exclusive scan over 2 * P1_BLOCK logical entries (entry b and b + 512 per lane); three barriers
__device__ __forceinline__ void p1_scan_pair(uint32_t v0, uint32_t v1, uint32_t* wave_tot, uint32_t& e0, uint32_t& e1) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = wave_inclusive_scan(v0), i1 = wave_inclusive_scan(v1);
    lds_barrier();
    if (lane == 63) { wave_tot[wave] = i0; wave_tot[8 + wave] = i1; }
    lds_barrier();
    // the sixteen wave totals (eight of the first half, eight of the second) in one scan: lane w of every wave takes entry w
    const uint32_t sc = wave_inclusive_scan(lane < 16 ? wave_tot[lane] : 0u);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t p0 = wv ? lane_value(sc, (int)wv - 1) : 0u;           // first-half totals of the waves before this one
    const uint32_t p1 = lane_value(sc, 7 + (int)wv);                       // all of the first half + second-half totals of the waves before
    e0 = p0 + i0 - v0;
    e1 = p1 + i1 - v1;
}

static __global__ void __launch_bounds__(P1_BLOCK)
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
    u32x4 raw = p1_tile_issue(bases, n, (t0 < t1 ? t0 : 0) * P1_TILE_STARTS);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        uint32_t w[4];
        p1_tile_fix(bases, n, tile * P1_TILE_STARTS, raw, w);
        raw = p1_tile_issue(bases, n, (tile + 1 < t1 ? tile + 1 : tile) * P1_TILE_STARTS);    // the next tile: in flight behind this one
        lds_barrier();
        uint32_t code, bad;
        encode16(w, code, bad);
        s_code[tid] = code; s_bad[tid] = bad;
        if (tid < 2) { s_code[P1_BLOCK + tid] = 0; s_bad[P1_BLOCK + tid] = 0xFFFF; }
        lds_barrier();
        if (tid < P1_LANES_WITH_STARTS) {
            LaneWindow lw;
            lw.init(s_code, s_bad, tid, t.k);
#pragma unroll 4
            for (int j = 0; j < PART_ITEMS; ++j, lw.step()) {
                if (!lw.valid()) continue;
                const uint64_t key = canon_if(lw.fwd(), t.k, canonical);
                if (key == EMPTY) { ++ones; continue; }
                atomicAdd(&s_hist[place_digit1_of(key, g.pl)], 1u);
            }
        }
    }
    lds_barrier();
    for (uint32_t b = tid; b < g.P1; b += P1_BLOCK) hist1[(uint64_t)blockIdx.x * g.P1 + b] = s_hist[b];
    for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
    if ((tid & 63) == 0 && ones) atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)ones);
}

// SEG = false: the exact edition -- every workgroup's share of every bucket was counted first (k_p1v2_count, k_p1_scan): `offs`
// holds where it starts, l1_off where the bucket does.  SEG = true: no counting pass -- bucket b is cut into one SEGMENT of seg_cap
// items per workgroup (segment (b, w) = items [w * seg_cap, (w + 1) * seg_cap) of bucket b; seg_cap a multiple of 4: whole groups);
// the placement spreads a workgroup's k-mers evenly, so a capacity of the expected share + 1/24 + 64 holds them (5 sigma at the bench
// size); what a full segment cannot take goes to the overflow list as a k-mer (-> direct path; if that overflows too the host redoes
// the round with the exact edition), what a segment has left at the end is padded with "no item", which level 2 skips.  Saves
// the second decode + hash of the whole input (k_p1v2_count: 70 ms of the bench step) for ~4 % more level-1 bytes.
// LEAN: the ranking sweep on 32-bit halves and a copy-out that reads the chosen strand's k-mer off the staged stream of that strand
// (kg_l1_lean.hpp; host-checked arithmetic).
// The segmented edition writes WHOLE GROUPS (HB1 <= 2): a bucket's k-mers of a tile are padded to a multiple of four in LDS ("no item")
// and leave as groups, one lane per group and two stores per four k-mers -- 16 bytes of low words + 4 HB1 of high parts -- where
// item-by-item stores (a dword and a short per k-mer, 16-byte pieces 24 bytes apart) measured 178 ms for the bench's level 1
// against 157 with one 8-byte store per k-mer: the kernel is bound by how many write requests reach the memory system, not by bytes
// or instructions.  The padding (1.5 items per tile and bucket, 11 % at the bench's 13 k-mers per tile and bucket) is room the 8
// bytes per item of the buffer have.  HB1 = 4 (8-byte items: n1 > 48) keeps one item per lane: there a padded group would not fit.
template <bool SEG, bool LEAN = false, int PB = MAX_PARTS>
__global__ void __launch_bounds__(P1_BLOCK, SEG && PB == 512 ? 6 : 4)   // six waves per SIMD = three workgroups per CU (g_p1_wgs): at most 80 VGPRs; the other shapes' LDS holds two workgroups
k_p1v2_scatter(DevTable t, PartGeom g, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
               const uint64_t* __restrict__ offs, const uint64_t* __restrict__ l1_off, uint8_t* __restrict__ l1_buf, uint32_t seg_cap /* SEG: < 2^24, a multiple of 4 */,
               uint32_t bucket_stride /* SEG: bytes from one bucket's first group to the next's, < 2^32 */,
               uint32_t seg_real /* SEG: k-mers a segment takes at most (<= seg_cap; the slots beyond are room for group padding) */,
               uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap) {
    __shared__ __attribute__((aligned(16))) P1LdsT<LEAN, SEG, PB> L;
    typedef typename P1LdsT<LEAN, SEG, PB>::cursor_t cursor_t;
    typedef P1LdsT<LEAN, SEG, PB> P1Lds_t;
    static_assert(!LEAN || offsetof(P1Lds_t, rcode) - offsetof(P1Lds_t, code) == 4 * P1_RCW, "the copy-out reads code | bad | rcode as one array");
    const uint32_t tid = threadIdx.x, P = g.P1, k = t.k;           // P <= PB (host-checked)
    const bool canonical = t.canonical != 0;
    const uint32_t hb1 = g.hb1, gs1 = 16 + 4 * hb1;
    const bool grouped = SEG && hb1 != 4;                          // runs staged and written as whole groups
    uint32_t ones = 0;
    // SEG: this workgroup's segment of bucket b starts seg_off bytes into the bucket (whole groups)
    const uint64_t seg_off = SEG ? (uint64_t)blockIdx.x * (seg_cap >> 2) * gs1 : 0;
    for (uint32_t b = tid; b < P; b += P1_BLOCK) { L.cursor[b] = SEG ? (cursor_t)0 : (cursor_t)(offs[(uint64_t)blockIdx.x * P + b] - l1_off[b]); if (SEG) L.real[b] = 0; }
    constexpr uint32_t DUMP0 = LEAN ? sizeof(L.pos) / 4 - 128 : 0;                 // LEAN: first dump slot of pos[] (a multiple of 4: a group boundary)
    static_assert(DUMP0 % 4 == 0, "dump slots");
    if (LEAN && tid < 64) L.off[PB + tid] = grouped ? DUMP0 / 4 : DUMP0;
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    u32x4 raw = p1_tile_issue(bases, n, (t0 < t1 ? t0 : 0) * P1_TILE_STARTS);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        uint32_t w[4];
        p1_tile_fix(bases, n, tile * P1_TILE_STARTS, raw, w);
        lds_barrier();                                   // previous tile's copy-out / cursor update done
        for (uint32_t b = tid; b < PB + (LEAN ? 64 : 0); b += P1_BLOCK) L.hist[b] = 0;     // (the dump counters too: their ranks are slots of the dump area)
        p1_tile_stage(L, w);                               // ends with a barrier
        // sweep 1: bucket and rank of every valid window of this lane
        uint32_t br[PART_ITEMS];
        uint32_t valid = 0;
        if (LEAN) {                                           // (every lane: the tile's last two hold no window start -- sixteen windows "without a k-mer" each, like any other such window; sweep 2 is straight-line for all)
            // Straight-line: every window is worked whatever its flags say (a lane-divergent branch saves nothing), a window that
            // holds a flagged base ranks in one of 64 dump counters (hist[MAX_PARTS + lane]) instead of a bucket, and a rank is
            // looked at two windows after its atomic was issued: up to three LDS round trips in flight per lane where a branch per
            // window waited for each (sixteen serialised round trips per lane and tile).
            const LeanGeom lg = lean_geom(k, canonical, g.pl.n1);
            const uint32_t v16 = tid < P1_LANES_WITH_STARTS ? lean_valid16(L.bad[tid], L.bad[tid + 1], L.bad[tid + 2], k) : 0u;
            LeanWin w{L.code[tid], L.code[tid + 1], L.code[tid + 2], 0, 0};
            uint32_t f_hi, f_lo;
            lean_fwd(w, lg, f_hi, f_lo);
            { const uint64_t rc0 = kmer_revcomp(((uint64_t)f_hi << 32) | f_lo, k); w.rc_hi = (uint32_t)(rc0 >> 32); w.rc_lo = (uint32_t)rc0; }
            valid = v16;                                                     // bit 15 - j: window j (k <= 31: no k-mer is the all-ones word)
            const uint32_t dump = PB + (tid & 63);
            uint32_t rk[PART_ITEMS];
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j) {
                if (j) { lean_step(w); lean_fwd(w, lg, f_hi, f_lo); lean_rc_roll(w, lg, f_lo); }
                uint32_t key_hi, key_lo;
                bool took_rc;                                                // the reverse complement is the canonical form
                const uint32_t b = lean_digit1(w, lg, g.pl, f_hi, f_lo, key_hi, key_lo, took_rc);
                const bool ok = (v16 & (0x8000u >> j)) != 0;
                const uint32_t sel = ok ? b : dump;                          // a window without a k-mer: a dump counter is its bucket from here on (sweep 2 stages it in the dump slots, nobody copies it out)
                rk[j] = atomicAdd(&L.hist[sel], 1u);
                br[j] = (sel << 16) | (took_rc ? 0x8000u : 0u);
                if (j >= 2) br[j - 2] |= rk[j - 2];
            }
#pragma unroll
            for (int j = PART_ITEMS - 2; j < PART_ITEMS; ++j) br[j] |= rk[j];
        }
        if (!LEAN && tid < P1_LANES_WITH_STARTS) {
            LaneWindow lw;
            lw.init(L.code, L.bad, tid, k);
            // the reverse complement rolls along with the window: out goes its last base, in comes the complement of the window's
            // new last base at the top (seven operations instead of the sixteen of a bit-reversal per window)
            uint64_t rc = kmer_revcomp(lw.fwd(), k);
            const uint32_t top = 2 * k - 2;
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j) {
                if (j) { lw.step(); rc = (rc >> 2) | ((uint64_t)(3u ^ ((uint32_t)lw.fwd() & 3u)) << top); }
                br[j] = 0;
                if (!lw.valid()) continue;
                const uint64_t fw = lw.fwd();
                const uint64_t key = canonical ? (rc < fw ? rc : fw) : fw;
                if (key == EMPTY) { if (SEG) ++ones; continue; }               // exact edition: tallied by the count pass
                const uint32_t b = place_digit1_of(key, g.pl);
                br[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
                valid |= 1u << j;
            }
        }
        raw = p1_tile_issue(bases, n, (tile + 1 < t1 ? tile + 1 : tile) * P1_TILE_STARTS);    // the next tile: in flight behind the rest of this one (issued here, not before the sweep: four registers the sweep needs)
        lds_barrier();
        uint32_t e0, e1;
        {
            const uint32_t h0 = tid < P ? L.hist[tid] : 0, h1 = PB > P1_BLOCK && tid + P1_BLOCK < P ? L.hist[tid + P1_BLOCK] : 0;
            p1_scan_pair(grouped ? (h0 + 3) >> 2 : h0, grouped ? (h1 + 3) >> 2 : h1, L.wave_tot, e0, e1);     // grouped: off[] counts groups
            L.off[tid] = e0;
            if (PB > P1_BLOCK) L.off[tid + P1_BLOCK] = e1;
            if (grouped) {                                   // what a run leaves of its last group is "no k-mer"
                for (uint32_t q = h0; q & 3; ++q) L.pos[4 * e0 + q] = P1_PAD;
                if (PB > P1_BLOCK) for (uint32_t q = h1; q & 3; ++q) L.pos[4 * e1 + q] = P1_PAD;
            }
        }
        lds_barrier();
        // sweep 2: park the tile position of every k-mer in its bucket's run
        if (LEAN) {
            // (straight-line like sweep 1: every window reads its bucket's run start and writes -- a window without a k-mer ranked in a dump
            // counter, whose "run" is the dump slots behind the array: no test, no select)
            uint32_t run0[PART_ITEMS];
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j) run0[j] = L.off[br[j] >> 16];
            // What is parked is not the window's tile position but WHERE ITS K-MER IS READ FROM at copy-out (kg_l1_lean.hpp: lean_entry_*): the word
            // in front of its first code word and the funnel shift, on the stream of the strand sweep 1 chose.  Both are affine in the
            // lane (a lane's windows start 16 tid + j bases into the tile; the other strand's stream runs backwards, so its word index falls
            // with tid where the forward one rises, and its shift does not depend on the lane at all: 16 tid has no low nibble), so an entry
            // costs sweep 2 four operations more than the position did -- and saves the copy-out eleven per staged k-mer (round 5: the
            // copy-out derived all this from the position, per entry: 22 of its 26 VALU instructions per k-mer).
            const uint32_t gshift = grouped ? 2u : 0u;          // (off[] counts groups when runs are staged as whole groups)
            uint32_t base_f = tid * 32u;
            asm volatile("" : "+v"(base_f));                    // (opaque per tile: sixteen per-lane constants hoisted out of the tile loop cost sixteen registers -- five of them spilled)
            const uint32_t Q = (uint32_t)P1_TILE_BYTES - 1u - k;
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j) {
                asm volatile("" : "+v"(run0[j]));               // (keeps the read where it is)
                const uint32_t cj = lean_entry_fwd(j), sj = lean_entry_rc(Q - (uint32_t)j);     // (wave-uniform)
                uint32_t on_rc = (uint32_t)((int32_t)(br[j] << 16) >> 31);                      // all ones: the other strand's stream (v_bfe_i32)
                asm volatile("" : "+v"(on_rc));                 // (a mask, not a condition: one v_bfi_b32 picks the strand's entry)
                const uint32_t e = ((sj - base_f) & on_rc) | ((base_f + cj) & ~on_rc);
                L.pos[(run0[j] << gshift) + (br[j] & 0x7FFFu)] = (br[j] & 0xFFFF0000u) | (e & 0xFFFFu);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j)
                if (valid >> j & 1) {
                    const uint32_t run0 = L.off[br[j] >> 16];
                    L.pos[(grouped ? 4 * run0 : run0) + (br[j] & 0xFFFFu)] = (br[j] & 0xFFFF0000u) | (tid * PART_ITEMS + j);
                }
        }
        lds_barrier();
        // the k-mer of a staged entry (the chosen strand's, read off that strand's stream) from three code words
        // A staged entry's window starts at base p of its strand's stream.  With q = p - 1: word wd = q >> 4 (-1 for p = 0: the word
        // before the array, whose value does not matter) and a funnel shift by sh = 30 - 2 (q & 15) = 0 ... 30 -- no special case for a
        // window that starts on a word boundary (v_alignbit_b32 takes its amount mod 32: a shift by 32 it cannot do).
        auto entry_src = [&](uint32_t v, const uint32_t*& src, int32_t& wd, uint32_t& sh) {
            if (LEAN) {                                          // sweep 2 parked the answer (kg_l1_lean.hpp: lean_entry_*): two operations instead of thirteen
                src = L.code - 1;                                // (word 0 = the word in front of L.code: a window that starts on the tile's first base shifts by 0 and never looks at it)
                wd = (int32_t)((v >> 5) & 0x7FFu); sh = v;       // (v_alignbit_b32 takes its amount mod 32)
                return;
            }
            const uint32_t p = v & 0xFFFFu;
            const int32_t q = (int32_t)p - 1;
            src = L.code;
            wd = q >> 4; sh = 2u * (~(uint32_t)q & 15u);
        };
        auto entry_key = [&](uint32_t c0, uint32_t c1, uint32_t c2, uint32_t sh) -> uint64_t {
            // the 64 bits that start 32 - sh bits into c0 : c1 : c2
            const uint32_t h1 = __builtin_amdgcn_alignbit(c0, c1, sh), h0 = __builtin_amdgcn_alignbit(c1, c2, sh);
            uint64_t key1 = (((uint64_t)h1 << 32) | h0) >> (64 - 2 * k);
            if (!LEAN) key1 = canon_if(key1, k, canonical);
            return key1;
        };
        const uint32_t lastb = P - 1;
        const uint32_t total = L.off[lastb] + (grouped ? (L.hist[lastb] + 3) >> 2 : L.hist[lastb]);      // groups (grouped) / k-mers
        // copy-out, grouped: one staged group per lane and step
        auto copy_out_groups = [&](auto hb1_tag) {
            constexpr int HB1 = decltype(hb1_tag)::value;
            const uint32_t seg_groups = seg_cap >> 2;
            for (uint32_t gi = tid; gi < total; gi += P1_BLOCK) {
                const u32x4 e = *reinterpret_cast<const u32x4*>(&L.pos[4 * gi]);
                const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
                const uint32_t b = e.x >> 16;                      // (a group's first entry is a k-mer)
                uint32_t c0[4], c1[4], c2[4], o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                  // (a padding entry reads tile position 0 of the forward stream: harmless)
                    const uint32_t* src; int32_t wd;
                    entry_src(ev[q], src, wd, o[q]);
                    c0[q] = src[wd]; c1[q] = src[wd + 1]; c2[q] = src[wd + 2];
                }
                const uint32_t gahead = gi - L.off[b];
                const uint32_t grel = ((uint32_t)L.cursor[b] >> 2) + gahead;              // group of the segment
                uint32_t re = L.real[b];
                asm volatile("" : "+v"(re));                   // (read with the other two, not in a branch of the test below)
                // a segment takes seg_real k-mers at most (the groups of a run before this one are full ones)
                const bool room = (grel < seg_groups) & (re + 4 * gahead + 4 <= seg_real);
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t pad = (uint32_t)((int32_t)ev[q] >> 31);                // all ones for a padding entry
                    if (LEAN) {
                        // the item = the k-mer's low n1 bits, n1 >= 32 on this path (lean_applies): the low word whole, n1 - 32 bits of the high
                        // one -- on 32-bit halves: a funnel shift and a bit-field extract where the 64-bit form shifts and masks twice
                        const uint32_t h1 = __builtin_amdgcn_alignbit(c0[q], c1[q], o[q]), h0 = __builtin_amdgcn_alignbit(c1[q], c2[q], o[q]);
                        lo[q] = __builtin_amdgcn_alignbit(h1, h0, 64 - 2 * k) | pad;
                        hi[q] = __builtin_amdgcn_ubfe(h1, 64 - 2 * k, g.pl.n1 - 32) | pad;
                    } else {
                        const uint64_t r1 = entry_key(c0[q], c1[q], c2[q], o[q]) & g.pl.m1;
                        lo[q] = (uint32_t)r1 | pad;
                        hi[q] = (uint32_t)(r1 >> 32) | pad;
                    }
                }
                if (room) {
                    const u32x4 glo = {lo[0], lo[1], lo[2], lo[3]};
                    typename HiGroup<HB1>::type ghi{};
                    if constexpr (HB1 == 1) ghi = (hi[0] & 0xFFu) | ((hi[1] & 0xFFu) << 8) | ((hi[2] & 0xFFu) << 16) | (hi[3] << 24);
                    if constexpr (HB1 == 2) { ghi.x = (hi[0] & 0xFFFFu) | (hi[1] << 16); ghi.y = (hi[2] & 0xFFFFu) | (hi[3] << 16); }
                    uint8_t* segp = l1_buf + ((uint64_t)b * bucket_stride + seg_off);
                    l2_store_group<HB1>(segp, grel, glo, ghi);
                } else {                                             // the segment is full: the overflow list
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (ev[q] != P1_PAD) {                        // (the k-mer back from the item and the bucket)
                            const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                            if (at < ovf_cap) ovf_buf[at] = place_key_r1(b, ((uint64_t)hi[q] << 32) | lo[q], g.pl);
                        }
                }
            }
        };
        // copy-out, one staged k-mer per lane and step (the exact edition; 8-byte items): its bucket travels with its position, so
        // no lane idles on a short run and the steps are independent of each other (a loop over buckets serialised ~24 LDS round
        // trips per lane group and was 57 % of this kernel: cycle stamps; same-box A/B 228 -> 216 ms).  Neighbouring lanes still
        // write neighbouring items inside a run.  Four steps at a time: the three dependent LDS round trips of a step (position;
        // code words; cursor and run start) overlap with those of the other three.
        auto copy_out = [&](auto hb1_tag) {
            constexpr uint32_t HB1 = decltype(hb1_tag)::value, GS1 = 16 + 4 * HB1;
            constexpr int CU = 4;
            for (uint32_t idx0 = tid; idx0 < total; idx0 += CU * P1_BLOCK) {
                uint32_t v[CU], c0[CU], c1[CU], c2[CU], o[CU], run0[CU];
                cursor_t cur[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) { const uint32_t i = idx0 + u * P1_BLOCK; v[u] = L.pos[i < total ? i : idx0]; }
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const uint32_t* src; int32_t wd;
                    entry_src(v[u], src, wd, o[u]);
                    c0[u] = src[wd]; c1[u] = src[wd + 1]; c2[u] = src[wd + 2];
                    const uint32_t b = v[u] >> 16;
                    cur[u] = L.cursor[b]; run0[u] = L.off[b];
                }
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const uint32_t idx = idx0 + u * P1_BLOCK;
                    if (idx >= total) break;
                    const uint32_t b = v[u] >> 16;
                    const uint64_t key1 = entry_key(c0[u], c1[u], c2[u], o[u]);
                    const uint64_t r1 = key1 & g.pl.m1;
                    const uint32_t ahead = idx - run0[u];
                    if (SEG) {
                        const uint32_t rel = (uint32_t)cur[u] + ahead;      // item of the segment (32-bit: seg_cap < 2^24, a tile adds < 2^13)
                        if (rel < seg_real) l1_put<HB1>(l1_buf + ((uint64_t)b * bucket_stride + seg_off), __umul24(rel >> 2, GS1), rel & 3, (uint32_t)r1, (uint32_t)(r1 >> 32));
                        else {                                               // the segment is full: the overflow list
                            const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                            if (at < ovf_cap) ovf_buf[at] = key1;
                        }
                    } else {
                        const uint64_t i = (uint64_t)cur[u] + ahead;        // item of the bucket
                        l1_put<HB1>(l1_buf + l1_bucket_base(l1_off[b], b), (i >> 2) * GS1, (uint32_t)i & 3, (uint32_t)r1, (uint32_t)(r1 >> 32));
                    }
                }
            }
        };
        switch (hb1) {                                                     // (wave-uniform; decided outside the loops)
        case 0: if (SEG) copy_out_groups(std::integral_constant<int, 0>{}); else copy_out(std::integral_constant<uint32_t, 0>{}); break;
        case 1: if (SEG) copy_out_groups(std::integral_constant<int, 1>{}); else copy_out(std::integral_constant<uint32_t, 1>{}); break;
        case 2: if (SEG) copy_out_groups(std::integral_constant<int, 2>{}); else copy_out(std::integral_constant<uint32_t, 2>{}); break;
        default: copy_out(std::integral_constant<uint32_t, 4>{}); break;
        }
        lds_barrier();
        for (uint32_t b = tid; b < P; b += P1_BLOCK) {
            if (!SEG) L.cursor[b] = (cursor_t)((uint64_t)L.cursor[b] + L.hist[b]);
            else if (!grouped) { const uint32_t c = (uint32_t)L.cursor[b] + L.hist[b]; L.cursor[b] = (cursor_t)(c < seg_real ? c : seg_real); }
            else {                                           // the groups that were written: a prefix of the run's, by both limits (copy_out_groups)
                const uint32_t h = L.hist[b], gc = (h + 3) >> 2, cur = (uint32_t)L.cursor[b], re = L.real[b];
                const uint32_t by_slots = (seg_cap - cur) >> 2, by_real = (seg_real - re) >> 2;
                const uint32_t w = gc < by_slots ? (gc < by_real ? gc : by_real) : (by_slots < by_real ? by_slots : by_real);
                L.cursor[b] = (cursor_t)(cur + 4 * w);
                L.real[b] = re + (h < 4 * w ? h : 4 * w);
            }
        }
    }
    if (SEG) {
        lds_barrier();
        // what the segments have left is padded with "no item"; a 16-lane group per bucket
        const uint32_t grp = tid >> 4, l16 = tid & 15;
        for (uint32_t b = grp; b < P; b += P1_BLOCK / 16) {
            uint8_t* base = l1_buf + ((uint64_t)b * bucket_stride + seg_off);
            for (uint32_t i = (uint32_t)L.cursor[b] + l16; i < seg_cap; i += 16) l1_put_any(base, __umul24(i >> 2, gs1), i & 3, hb1, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
        if ((tid & 63) == 0 && ones) atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)ones);
    }
}

}  // namespace kg
#include "kg_l1_blocks.hpp"
namespace kg {

// bucket b1 of the level-1 buffer: its first item and number of items (in items of the round; segment and group padding
// included) and the byte its groups start at.  Exact layout (l1_off) or segmented (seg_slots = workgroups x seg_cap items per bucket,
// "no item"-padded, buckets l1_stride bytes apart).
// first / n_real: the k-mers before this bucket / in it (segmented: their bounds) -- what level 2 places and sizes its runs by
__device__ __forceinline__ uint64_t l1_bucket_range(const PartGeom& g, const uint64_t* __restrict__ l1_off, uint64_t seg_slots, uint32_t b1, uint64_t& first, uint64_t& n_items, uint64_t& n_real) {
    if (seg_slots) { first = (uint64_t)b1 * g.l1_real; n_items = seg_slots; n_real = g.l1_real; return (uint64_t)b1 * g.l1_stride; }
    first = l1_off[b1]; n_items = n_real = l1_off[b1 + 1] - first;
    return l1_bucket_base(first, b1);
}

// ---- level 2 ----
// A level-1 item is r1, the part of the k-mer below the level-1 digit (kg_device.hpp "placement"): its top bits, displaced by a hash of
// the rest, are the level-2 digit, which sorts; the rest alone -- the remainder -- reaches HBM.
// The N items of a lane for one tile (items [tbeg, tbeg + N * 1024), tbeg a multiple of 4) of a bucket of n_items items whose groups
// start at `bucket`: lane t takes groups t, t + 1024, ... of the tile: N / 4 loads of 16 bytes + N / 4 of 4 HB1.  Bit j of the result
// = item j is there (not past the end, not padding).
// A tile's items in registers: the low words and the high parts, the latter two 16-bit halves to a register when an item has at most
// 48 bits (W1 = false: HB1 <= 2; 24 registers for 16 items where 64-bit k-mers took 32), one register each otherwise.
template <int N, bool W1>
struct TileItems {
    uint32_t lo[N];
    uint32_t hi[W1 ? N : N / 2];
    __device__ __forceinline__ uint64_t r1(int j) const {
        const uint32_t h = W1 ? hi[j] : (hi[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        return ((uint64_t)h << 32) | lo[j];
    }
};
// the wide form (HB1 = 4): 64-bit items, two to a 16-byte load
template <int N>
__device__ __forceinline__ uint32_t p2_tile_load_wide(const uint8_t* __restrict__ bucket, uint64_t tbeg, uint64_t n_items, TileItems<N, true>& it) {
    // Unconditional loads (a load inside a branch is waited for at the end of the branch: the compiler once pulled a padding test
    // into the branch of each load -- sixteen serialised round trips to HBM, 29 K of a tile's 54 K cycles), and not even a clamped
    // index: what lies behind a bucket's last item is the next bucket, and behind the level-1 buffer the level-2 buffer (the arena
    // carve, kg_count.hip), so a tile's loads are always of mapped memory; what they hold is masked below.
    static_assert(N % 2 == 0, "pairs");
    const uint64_t left = n_items - tbeg;                                      // (tbeg < n_items)
    const uint32_t t_items = left < (uint64_t)N * PART_BLOCK ? (uint32_t)left : (uint32_t)N * PART_BLOCK;
    const uint8_t* mine = bucket + (tbeg + 2 * threadIdx.x) * 8;
    u32x4 w[N / 2];
#pragma unroll
    for (int u = 0; u < N / 2; ++u) w[u] = *reinterpret_cast<const u32x4_a4*>(mine + (uint64_t)u * PART_BLOCK * 16);
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < N / 2; ++u) {
        const uint32_t i0 = 2 * ((uint32_t)u * PART_BLOCK + threadIdx.x);
        it.lo[2 * u] = w[u].x; it.hi[2 * u] = w[u].y; it.lo[2 * u + 1] = w[u].z; it.hi[2 * u + 1] = w[u].w;
        uint32_t t0 = (i0 < t_items && (w[u].x & w[u].y) != 0xFFFFFFFFu) ? 1u : 0u, t1 = (i0 + 1 < t_items && (w[u].z & w[u].w) != 0xFFFFFFFFu) ? 1u : 0u;
        asm volatile("" : "+v"(t0), "+v"(t1));                // (0 / 1 shifted into place, as in the narrow form)
        valid |= (t0 << (2 * u)) | (t1 << (2 * u + 1));
    }
    return valid;
}
// The narrow forms (HB1 = 0, 1, 2) in ONE code path -- a three-way switch over the templated loader left the compiler with three
// sets of values to keep (18 spilled registers at the bench's shape): 16 + 8 bytes are loaded per group whatever HB1 is (what lies
// behind a shorter group is the next group, behind a bucket's last one the next bucket, behind the level-1 buffer the level-2 buffer
// -- mapped, ignored; loads are unconditional because a load inside a branch is waited for at the end of the branch), and one
// v_perm_b32 per register puts the high parts where TileItems
// wants them, its selector chosen by HB1 (wave-uniform).
// In two steps, so that a tile's loads can be in flight while the tile before it is worked on (the block edition of the one-pass level 2):
// `issue` is the unconditional loads, `decode` looks at them when they are needed.
template <int N> struct RawTile { u32x4 lo[N / 4]; u32x2 hi[N / 4]; };
template <int N>
__device__ __forceinline__ void p2_tile_issue_narrow(uint32_t hb1, const uint8_t* __restrict__ bucket, uint64_t tbeg, RawTile<N>& r) {
    static_assert(N % 4 == 0, "whole groups");
    const uint32_t gs1 = 16 + 4 * hb1;
    const uint8_t* tile = bucket + (tbeg >> 2) * gs1;                          // (wave-uniform; the lane's part is a 32-bit offset: one register to keep, not two)
    const uint32_t mine = threadIdx.x * gs1;
#pragma unroll
    for (int u = 0; u < N / 4; ++u) {
        const uint8_t* p = tile + (mine + (uint32_t)u * PART_BLOCK * gs1);
        r.lo[u] = *reinterpret_cast<const u32x4_a4*>(p);
        r.hi[u] = *reinterpret_cast<const u32x2_a4*>(p + 16);
    }
}
template <int N>
__device__ __forceinline__ uint32_t p2_tile_decode_narrow(uint32_t hb1, uint64_t tbeg, uint64_t n_items, const RawTile<N>& r, TileItems<N, false>& it) {
    const uint32_t hi_none = hb1 == 0 ? 0u : hb1 == 1 ? 0xFFu : 0xFFFFu;
    // v_perm_b32(y, x, sel): selector bytes 0-3 take x's bytes, 4-7 y's, 0x0C gives 0.  HB1 = 2: (x, y) are the pairs already; HB1 = 1:
    // x holds four bytes -> {x0, 0, x1, 0}, {x2, 0, x3, 0}; HB1 = 0: nothing
    const uint32_t sel0 = hb1 == 2 ? 0x03020100u : hb1 == 1 ? 0x0C010C00u : 0x0C0C0C0Cu, sel1 = hb1 == 2 ? 0x07060504u : hb1 == 1 ? 0x0C030C02u : 0x0C0C0C0Cu;
    const uint64_t left = tbeg < n_items ? n_items - tbeg : 0;
    const uint32_t t_items = left < (uint64_t)N * PART_BLOCK ? (uint32_t)left : (uint32_t)N * PART_BLOCK;
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < N / 4; ++u) {
        const uint32_t i0 = 4 * ((uint32_t)u * PART_BLOCK + threadIdx.x);
        const uint32_t h01 = __builtin_amdgcn_perm(r.hi[u].y, r.hi[u].x, sel0), h23 = __builtin_amdgcn_perm(r.hi[u].y, r.hi[u].x, sel1);
        it.hi[2 * u] = h01; it.hi[2 * u + 1] = h23;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l = q == 0 ? r.lo[u].x : q == 1 ? r.lo[u].y : q == 2 ? r.lo[u].z : r.lo[u].w;
            const uint32_t h = ((q < 2 ? h01 : h23) >> (16 * (q & 1))) & 0xFFFFu;
            it.lo[4 * u + q] = l;
            uint32_t there = (i0 + q < t_items && !(l == 0xFFFFFFFFu && h == hi_none)) ? 1u : 0u;
            asm volatile("" : "+v"(there));                   // (a 0 / 1 shifted into place: as a select of 1 << j the sixteen masks sat in sixteen registers for the whole kernel)
            valid |= there << (4 * u + q);
        }
    }
    return valid;
}
template <int N>
__device__ __forceinline__ uint32_t p2_tile_load_narrow(uint32_t hb1, const uint8_t* __restrict__ bucket, uint64_t tbeg, uint64_t n_items, TileItems<N, false>& it) {
    RawTile<N> r;
    p2_tile_issue_narrow<N>(hb1, bucket, tbeg, r);
    return p2_tile_decode_narrow<N>(hb1, tbeg, n_items, r, it);
}
// The level-1 buffer's BLOCKED form (kg_l1_blocks.hpp: 6-byte items in 64-byte blocks of five pair records): a lane takes N / 2 pairs, each with
// ONE 12-byte load -- {low a, low b, high a | high b << 16} is what TileItems keeps: no permute.  Pair p of the bucket lies 12 (p % 5) bytes into
// block p / 5.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
template <int N> struct RawTileB { u32x3 v[N / 2]; };
template <int N>
__device__ __forceinline__ void p2_tile_issue_l1b(const uint8_t* __restrict__ bucket, uint64_t tbeg, RawTileB<N>& r) {
    const uint32_t p0 = (uint32_t)(tbeg >> 1) + threadIdx.x;                   // (a bucket holds fewer than 2^32 bytes)
    uint32_t blk = __umulhi(p0, 0xCCCCCCCDu) >> 2, part = p0 - 5 * blk;
#pragma unroll
    for (int u = 0; u < N / 2; ++u) {
        // (a 32-bit offset from the bucket's wave-uniform base: one register per load in flight, not two)
        r.v[u] = *reinterpret_cast<const u32x3_a4*>(bucket + (uint32_t)((blk << 6) + 12 * part));
        blk += PART_BLOCK / 5; part += PART_BLOCK % 5;                          // the lane's next pair is 1024 pairs on: 204 blocks and 4 pairs
        if (part >= 5) { part -= 5; ++blk; }
    }
}
template <int N>
__device__ __forceinline__ uint32_t p2_tile_decode_l1b(uint64_t tbeg, uint64_t n_items, const RawTileB<N>& r, TileItems<N, false>& it) {
    const uint64_t left = tbeg < n_items ? n_items - tbeg : 0;
    const uint32_t t_items = left < (uint64_t)N * PART_BLOCK ? (uint32_t)left : (uint32_t)N * PART_BLOCK;
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < N / 2; ++u) {
        const uint32_t i0 = 2 * ((uint32_t)u * PART_BLOCK + threadIdx.x);
        it.lo[2 * u] = r.v[u].x; it.lo[2 * u + 1] = r.v[u].y; it.hi[u] = r.v[u].z;
        uint32_t ta = (i0 < t_items && !(r.v[u].x == 0xFFFFFFFFu && (r.v[u].z & 0xFFFFu) == 0xFFFFu)) ? 1u : 0u;
        uint32_t tb = (i0 + 1 < t_items && !(r.v[u].y == 0xFFFFFFFFu && (r.v[u].z >> 16) == 0xFFFFu)) ? 1u : 0u;
        asm volatile("" : "+v"(ta), "+v"(tb));
        valid |= (ta << (2 * u)) | (tb << (2 * u + 1));
    }
    return valid;
}
template <int N, bool W1, bool L1B = false>
__device__ __forceinline__ uint32_t p2_tile_load(uint32_t hb1, const uint8_t* __restrict__ bucket, uint64_t tbeg, uint64_t n_items, TileItems<N, W1>& it) {
    if constexpr (W1) return p2_tile_load_wide<N>(bucket, tbeg, n_items, it);
    else if constexpr (L1B) { RawTileB<N> r; p2_tile_issue_l1b<N>(bucket, tbeg, r); return p2_tile_decode_l1b<N>(tbeg, n_items, r, it); }
    else return p2_tile_load_narrow<N>(hb1, bucket, tbeg, n_items, it);
}

// The exact edition's tile: counting-sort its k-mers by their level-2 digit through LDS and write them item by item at exact positions
// (cursor[]: the histogram pass sized every run).  All 1024 lanes must call it (barriers inside).
template <int HB, bool W1>
__device__ __forceinline__ void scatter_tile2(P2Lds<HB>& L, const PartGeom g, const TileItems<L2Fmt<HB>::N, W1>& key, uint32_t valid, uint8_t* __restrict__ out) {
    constexpr int N = L2Fmt<HB>::N;
    typedef typename HiWord<HB>::type hi_t;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = g.P2;
    if (tid < MAX_PARTS) L.hist[tid] = 0;
    lds_barrier();
    uint32_t br[N];                                           // digit << 16 | rank inside the tile's run
#pragma unroll
    for (int j = 0; j < N; ++j) {
        br[j] = 0;
        if (valid >> j & 1) {
            const uint32_t b = place_digit2_of(key.r1(j), g.pl);   // (one 32-bit multiply)
            br[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
        }
    }
    lds_barrier();
    uint32_t total;                                           // k-mers of the tile
    const uint32_t mine = tid < P ? L.hist[tid] : 0;
    const uint32_t excl = block_exclusive_scan(mine, L.wave_tot, &total);
    if (tid < MAX_PARTS) L.goff[tid] = excl;
    lds_barrier();
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (valid >> j & 1) {
            const uint32_t b = br[j] >> 16, slot = L.goff[b] + (br[j] & 0xFFFF);
            const uint64_t rem = key.r1(j) & g.pl.mr;
            L.st_lo[slot] = (uint32_t)rem;
            if (HB) L.st_hi[slot] = (hi_t)(rem >> 32);
        }
    lds_barrier();
    // copy-out, one k-mer per lane and step; its sub-bucket: the last one whose staged run starts at or before it
    for (uint32_t idx = tid; idx < total; idx += PART_BLOCK) {
        uint32_t lo_b = 0, hi_b = P - 1;
        while (lo_b < hi_b) { const uint32_t mid = (lo_b + hi_b + 1) >> 1; if (L.goff[mid] <= idx) lo_b = mid; else hi_b = mid - 1; }
        // (an empty sub-bucket shares its successor's start, so the last one found is never empty)
        const uint64_t rem = ((uint64_t)(HB ? (uint32_t)L.st_hi[idx] : 0u) << 32) | L.st_lo[idx];
        l2_put<HB>(out, L.cursor[lo_b] + (idx - L.goff[lo_b]), rem);
    }
    lds_barrier();
    if (tid < P) L.cursor[tid] += mine;
    // (the next tile's first barrier orders the cursor update before the next use)
}

// ---- the one-pass edition's tile ----
// LDS carve of k_p2_fast.  A sub-bucket's run has a fixed start and capacity (run of digit b = groups [b * capg, (b + 1) * capg) of the
// bucket's part of the level-2 buffer), so all a workgroup keeps per sub-bucket is how many groups it has written: 32 bits.
// CARRY (all but 8-byte items, whose tiles leave no room): what a sub-bucket's k-mers of a tile leave of a group of four -- up to
// three -- waits in LDS for the next tile instead of leaving as a padded group: the runs hold no padding but their last group's (it was
// 9 % of the level-2 buffer at the bench's 16 k-mers per tile and sub-bucket: slots the apply had to read and skip).
template <int HB>
struct P2FLds {
    static constexpr bool CARRY = HB != 4;
    uint32_t cur[MAX_PARTS];           // groups written into each sub-bucket's run so far
    uint32_t hist[MAX_PARTS + 64];     // k-mers of the tile per sub-bucket (CARRY: + those carried over); + one dump counter per lane of a wave
    uint32_t goff[MAX_PARTS];          // first staged group of the sub-bucket's k-mers of this tile (CARRY: | its whole groups << 16)
    uint32_t wave_tot[32];
    uint32_t pad_[32];                 // (st_lo starts on a 16-byte boundary)
    static constexpr uint32_t CBASE = L2Fmt<HB>::NS;           // CARRY: sub-bucket b's carried k-mers are entries CBASE + 3 b, + 1, + 2 of the staging arrays
    uint32_t st_lo[L2Fmt<HB>::NS + (CARRY ? 3 * MAX_PARTS : 0)];     // staged remainders, grouped by sub-bucket: low words ...
    typename HiWord<HB>::type st_hi[HB ? L2Fmt<HB>::NS + (CARRY ? 3 * MAX_PARTS : 0) : 4];   // ... high parts ...
    uint16_t grp_b[L2Fmt<HB>::NS / 4]; // ... and the sub-bucket of every staged group
};
static_assert(sizeof(P2FLds<0>) <= 160 * 1024 - 256 && sizeof(P2FLds<1>) <= 160 * 1024 - 256 && sizeof(P2FLds<2>) <= 160 * 1024 - 256 && sizeof(P2FLds<4>) <= 160 * 1024 - 256, "LDS");

// Counting-sort one tile's k-mers of a bucket by their level-2 digit through LDS and append every digit's whole groups to its run -- one
// 16-byte store of low words + one of high parts per four k-mers; what does not fit its run goes to the overflow list as a k-mer
// (through the inverse).  CARRY: `carry_n` is thread b's count of sub-bucket b's carried k-mers (in, and out for the next tile); they
// rank first.  Without: a digit's k-mers of the tile are padded to whole groups in LDS ("no item").  All 1024 lanes must call it
// (barriers inside).  The copy-out works two groups per lane at once, so that the dependent LDS round trips of the two overlap.
template <int HB, bool W1, bool STAMP = false>
__device__ __forceinline__ void scatter_tile2_fast(P2FLds<HB>& L, const PartGeom g, const uint32_t b1, const TileItems<L2Fmt<HB>::N, W1>& key, uint32_t valid,
                                                   uint8_t* __restrict__ out /* the bucket's first run */, uint32_t capg /* groups per run */,
                                                   uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap, uint32_t& carry_n,
                                                   unsigned long long* st = nullptr /* STAMP: cycles of [1] hash + rank, [2] scan, [3] staging, [4] copy-out */) {
    constexpr int N = L2Fmt<HB>::N;
    constexpr bool CARRY = P2FLds<HB>::CARRY;
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    const unsigned long long t0 = now();
    typedef typename HiWord<HB>::type hi_t;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = g.P2;
    if (tid < MAX_PARTS) L.hist[tid] = CARRY ? carry_n : 0u;
    lds_barrier();
    uint32_t br[N];                                           // digit << 16 | rank inside the tile's run
    // Straight-line, as level 1's sweep: every slot takes a rank -- those without a k-mer in one of 64 dump counters behind the histogram
    // -- and a rank is looked at two slots after its atomic was issued: three LDS round trips in flight per lane where a branch per slot
    // waited for each.  (It fits the registers since the block scan stopped keeping sixteen lane masks alive; before, it spilled
    // seventeen, and a spill costs this kernel a trip to memory per phase: 179 ms per step against 145-150.)
    {
        uint32_t rk[N];
        const uint32_t dump = MAX_PARTS + (tid & 63);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint32_t b = place_digit2_of(key.r1(j), g.pl) & (MAX_PARTS - 1);   // (one 32-bit multiply; the mask: a slot without a k-mer holds any bits)
            rk[j] = atomicAdd(&L.hist[(valid >> j & 1) ? b : dump], 1u);
            br[j] = b << 16;
            if (j >= 2) br[j - 2] |= rk[j - 2];
        }
#pragma unroll
        for (int j = N - 2; j < N; ++j) br[j] |= rk[j];
    }
    lds_barrier();
    const unsigned long long t1 = now();
    uint32_t total;                                           // staged groups of the tile
    const uint32_t mine = tid < P ? L.hist[tid] : 0;
    const uint32_t mygroups = CARRY ? mine >> 2 : (mine + 3) >> 2;      // CARRY: whole groups only
    const uint32_t excl = block_exclusive_scan(mygroups, L.wave_tot, &total);
    if (tid < MAX_PARTS) L.goff[tid] = CARRY ? excl | (mygroups << 16) : excl;      // (at most 4864 staged groups, at most 4096 of one sub-bucket)
    if (tid < P) {
        if (CARRY) {
            // the carried k-mers are the first of the run's first group -- when there is one; else they stay where they are
            if (mygroups) for (uint32_t q = 0; q < carry_n; ++q) { L.st_lo[excl * 4 + q] = L.st_lo[P2FLds<HB>::CBASE + 3 * tid + q]; if (HB) L.st_hi[excl * 4 + q] = L.st_hi[P2FLds<HB>::CBASE + 3 * tid + q]; }
        } else {
            // the last group of the run: what the k-mers leave is "no item"
            for (uint32_t q = mine; q & 3; ++q) { L.st_lo[excl * 4 + q] = 0xFFFFFFFFu; if (HB) L.st_hi[excl * 4 + q] = (hi_t)~(hi_t)0; }
        }
        for (uint32_t gq = 0; gq < mygroups; ++gq) L.grp_b[excl + gq] = (uint16_t)tid;      // whose groups these are
    }
    lds_barrier();
    const unsigned long long t2 = now();
    // (the run starts of four slots are read in one go -- a slot without a k-mer has a valid digit all the same --: a read per slot inside
    // its branch is a round trip per slot; all sixteen at once do not fit the registers)
    constexpr int SB = N % 8 == 0 && !W1 ? 8 : 4;
    static_assert(N % SB == 0, "batches");
    uint32_t go[SB];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j % SB == 0) {
#pragma unroll
            for (int q = 0; q < SB; ++q) go[q] = L.goff[br[j + q] >> 16];
#pragma unroll
            for (int q = 0; q < SB; ++q) asm volatile("" : "+v"(go[q]));          // (keeps the reads out of the branches below)
        }
        if (valid >> j & 1) {
            const uint32_t b = br[j] >> 16, r = br[j] & 0xFFFF, gf = go[j % SB];
            const uint64_t rem = key.r1(j) & g.pl.mr;
            if (CARRY) {
                // rank r of the run: in one of its whole groups, or one of the up to three k-mers behind them, which wait for the next tile
                const uint32_t whole = (gf >> 16) * 4u;
                const uint32_t slot = r < whole ? (gf & 0xFFFFu) * 4u + r : P2FLds<HB>::CBASE + 3 * b + (r - whole);
                L.st_lo[slot] = (uint32_t)rem;
                if (HB) L.st_hi[slot] = (hi_t)(rem >> 32);
            } else {
                const uint32_t slot = gf * 4u + r;
                L.st_lo[slot] = (uint32_t)rem;
                if (HB) L.st_hi[slot] = (hi_t)(rem >> 32);
            }
        }
    }
    lds_barrier();
    const unsigned long long t3 = now();
    // copy-out, one staged group per lane and step, two steps at a time: the steps are independent of each other
    for (uint32_t gi0 = tid; gi0 < total; gi0 += 2 * PART_BLOCK) {
        uint32_t b[2], ahead[2];
        u32x4 lo[2];
        typename HiGroup<HB>::type hi[2];
        bool in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const uint32_t gi = gi0 + u * PART_BLOCK; in[u] = gi < total; b[u] = L.grp_b[in[u] ? gi : gi0]; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t gi = in[u] ? gi0 + u * PART_BLOCK : gi0;
            ahead[u] = L.cur[b[u]] + (gi - (CARRY ? L.goff[b[u]] & 0xFFFFu : L.goff[b[u]]));
            lo[u] = *reinterpret_cast<const u32x4*>(&L.st_lo[gi * 4]);
            hi[u] = typename HiGroup<HB>::type{};
            if (HB) hi[u] = *reinterpret_cast<const typename HiGroup<HB>::type*>(&L.st_hi[gi * 4]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!in[u]) continue;
            if (ahead[u] < capg) l2_store_group<HB>(out, (uint64_t)b[u] * capg + ahead[u], lo[u], hi[u]);
            else {                                                             // beyond the run's capacity: the overflow list
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t rem = ((uint64_t)hi_of_group<HB>(hi[u], q) << 32) | (q == 0 ? lo[u].x : q == 1 ? lo[u].y : q == 2 ? lo[u].z : lo[u].w);
                    if (rem == L2Fmt<HB>::NONE) continue;
                    const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                    if (at < ovf_cap) ovf_buf[at] = place_key_d(b1, b[u], rem, g.pl);
                }
            }
        }
    }
    lds_barrier();
    if (STAMP && st) { const unsigned long long t4 = now(); st[1] += t1 - t0; st[2] += t2 - t1; st[3] += t3 - t2; st[4] += t4 - t3; }
    if (tid < P) { const uint32_t c = L.cur[tid] + mygroups; L.cur[tid] = c < capg ? c : capg; }
    if (CARRY) carry_n = mine & 3;
    // (the next tile's first barrier orders the cursor update before the next use)
}

// CARRY: what a bucket's last tile left waiting -- up to three k-mers per sub-bucket -- as the run's last group, padded with "no item"
template <int HB>
__device__ __forceinline__ void flush_carry2(P2FLds<HB>& L, const PartGeom g, const uint32_t b1, uint8_t* __restrict__ out, uint32_t capg,
                                             uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap, uint32_t carry_n) {
    const uint32_t tid = threadIdx.x;
    if (tid >= g.P2 || !carry_n) return;
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool have = (uint32_t)q < carry_n;                       // (q < 3 then)
        lo[q] = have ? L.st_lo[P2FLds<HB>::CBASE + 3 * tid + (q < 3 ? q : 0)] : 0xFFFFFFFFu;
        hi[q] = have && HB ? (uint32_t)L.st_hi[HB ? P2FLds<HB>::CBASE + 3 * tid + (q < 3 ? q : 0) : 0] : 0xFFFFFFFFu;
    }
    const uint32_t c = L.cur[tid];
    if (c < capg) {
        const u32x4 glo = {lo[0], lo[1], lo[2], lo[3]};
        typename HiGroup<HB>::type ghi{};
        if constexpr (HB == 1) ghi = (hi[0] & 0xFFu) | ((hi[1] & 0xFFu) << 8) | ((hi[2] & 0xFFu) << 16) | (hi[3] << 24);
        if constexpr (HB == 2) { ghi.x = (hi[0] & 0xFFFFu) | (hi[1] << 16); ghi.y = (hi[2] & 0xFFFFu) | (hi[3] << 16); }
        l2_store_group<HB>(out, (uint64_t)tid * capg + c, glo, ghi);
        L.cur[tid] = c + 1;
    } else {
        for (uint32_t q = 0; q < carry_n; ++q) {
            const uint64_t rem = ((uint64_t)(HB ? hi[q] & (HB == 1 ? 0xFFu : 0xFFFFu) : 0u) << 32) | lo[q];
            const unsigned long long at = atomicAdd(ovf_n, 1ULL);
            if (at < ovf_cap) ovf_buf[at] = place_key_d(b1, tid, rem, g.pl);
        }
    }
}

// ---- the one-pass edition for blocked items (HB = 1): whole 64-byte lines, a quad of lanes per line ----
// LDS carve.  Per sub-bucket a 64-byte BLOCK IMAGE (cb): the items that wait for their block to fill -- up to eleven -- already where the
// block wants them (low words 0 .. 11, high bytes at 48 .. 59); its last dword counts the blocks the sub-bucket's run has received.  A tile's
// k-mers rank behind the waiting ones; ranks below twelve complete the image, which then leaves as it stands (four lanes, sixteen bytes each,
// ONE store instruction: a whole line per request), ranks beyond fill whole blocks in the staging arrays (st_lo / st_hi, block-ordered, no
// padding), and what is left over -- again up to eleven -- goes into the image once it has left (phase C, after the copy-out's barrier).
struct P2BLds {
    static constexpr int TILE = L2Fmt<1>::TILE;
    uint32_t hist[MAX_PARTS + 64];     // k-mers of the tile per sub-bucket, counted from the waiting ones; + one dump counter per lane of a wave
    uint32_t goff[MAX_PARTS];          // first staged block of the sub-bucket's k-mers of this tile | blocks it completes this tile << 16
    uint32_t wave_tot[32];
    uint32_t pad_[32];                 // (cb starts on a 64-byte boundary)
    uint32_t cb[MAX_PARTS * 16];       // the block images
    uint32_t st_lo[TILE];              // staged whole blocks: low words (block s: entries 12 s ...)
    uint8_t st_hi[TILE + 16];          // ... high bytes
    uint16_t blk_b[TILE / 12 + 2];     // ... and whose they are
};
static_assert(sizeof(P2BLds) <= 160 * 1024 - 256 && offsetof(P2BLds, cb) % 64 == 0 && offsetof(P2BLds, st_lo) % 16 == 0 && offsetof(P2BLds, st_hi) % 4 == 0, "LDS");
template <int HB> struct P2FastLds { typedef P2FLds<HB> type; };
template <> struct P2FastLds<1> { typedef P2BLds type; };

// a block of run `b` that found the run full: its k-mers to the overflow list (lows in `lo`, the high byte of item q from `hi8(q)`)
template <typename HiFn>
__device__ __forceinline__ void blk_overflow(const PartGeom& g, uint32_t b1, uint32_t b, const u32x4& lo, HiFn hi8, uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t rem = ((uint64_t)hi8(q) << 32) | (q == 0 ? lo.x : q == 1 ? lo.y : q == 2 ? lo.z : lo.w);
        if (rem == L2Fmt<1>::NONE) continue;
        const unsigned long long at = atomicAdd(ovf_n, 1ULL);
        if (at < ovf_cap) ovf_buf[at] = place_key_d(b1, b, rem, g.pl);
    }
}

// One tile.  `carry_n` is thread b's count of sub-bucket b's waiting k-mers (in, and out for the next tile).  `out`: the bucket's first run;
// run b = blocks [b * capb, (b + 1) * capb) from there.  All 1024 lanes must call it (barriers inside).
template <bool W1, bool STAMP = false, typename Prefetch>
__device__ __forceinline__ void scatter_tile2_blk(P2BLds& L, const PartGeom g, const uint32_t b1, const TileItems<16, W1>& key, uint32_t valid,
                                                  uint8_t* __restrict__ out, uint32_t capb, uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap,
                                                  uint32_t& carry_n, unsigned long long* st, Prefetch prefetch /* issues the NEXT tile's loads: called once the ranking has let go of its registers */) {
    constexpr int N = 16;
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    const unsigned long long t0 = now();
    const uint32_t tid = threadIdx.x, P = g.P2;
    uint8_t* const cb8 = reinterpret_cast<uint8_t*>(L.cb);
    L.hist[tid] = carry_n;                                    // (PART_BLOCK == MAX_PARTS: thread b is sub-bucket b)
    lds_barrier();
    uint32_t br[N];                                           // digit << 16 | rank in the sub-bucket's line-up (the waiting ones first)
    {
        uint32_t rk[N];
        const uint32_t dump = MAX_PARTS + (tid & 63);
#pragma unroll
        for (int j = 0; j < N; ++j) {                         // (straight-line, as scatter_tile2_fast: slots without a k-mer rank in a dump counter)
            const uint32_t b = place_digit2_of(key.r1(j), g.pl) & (MAX_PARTS - 1);
            rk[j] = atomicAdd(&L.hist[(valid >> j & 1) ? b : dump], 1u);
            br[j] = b << 16;
            if (j >= 2) br[j - 2] |= rk[j - 2];
        }
#pragma unroll
        for (int j = N - 2; j < N; ++j) br[j] |= rk[j];
    }
    lds_barrier();
    const unsigned long long t1 = now();
    prefetch();
    const uint32_t avail = tid < P ? L.hist[tid] : 0;         // waiting + new (< 2^15)
    const uint32_t nblk = __umulhi(avail, 0xAAAAAAABu) >> 3;  // avail / 12: blocks that complete
    const uint32_t nstg = nblk ? nblk - 1 : 0;                // ... of which the first is the image, the others are staged
    uint32_t total;                                           // staged blocks of the tile
    const uint32_t excl = block_exclusive_scan(nstg, L.wave_tot, &total);
    L.goff[tid] = excl | (nblk << 16);
    for (uint32_t q = 0; q < nstg; ++q) L.blk_b[excl + q] = (uint16_t)tid;
    lds_barrier();
    const unsigned long long t2 = now();
    // phase A: ranks below twelve into the image, whole blocks' worth into the staging arrays; the rest waits for phase C
    // (a branch-free form -- one index into the carve for either destination, as level 1's block edition has it -- needs registers this kernel
    // does not have next to the tile in flight: nineteen spilled, some of them inside the tile loop)
    {
        constexpr int SB = 4;
        uint32_t go[SB];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j % SB == 0) {
#pragma unroll
                for (int q = 0; q < SB; ++q) go[q] = L.goff[br[j + q] >> 16];
#pragma unroll
                for (int q = 0; q < SB; ++q) asm volatile("" : "+v"(go[q]));
            }
            uint32_t later = 0;
            if (valid >> j & 1) {
                const uint32_t b = br[j] >> 16, v = br[j] & 0xFFFF, gf = go[j % SB], whole = (gf >> 16) * 12u;
                const uint64_t rem = key.r1(j) & g.pl.mr;
                if (v < 12) { L.cb[b * 16 + v] = (uint32_t)rem; cb8[b * 64 + 48 + v] = (uint8_t)(rem >> 32); }
                else if (v < whole) { const uint32_t slot = (gf & 0xFFFFu) * 12u + (v - 12); L.st_lo[slot] = (uint32_t)rem; L.st_hi[slot] = (uint8_t)(rem >> 32); }
                else later = 0x80000000u | (b << 4) | (v - whole);
            }
            br[j] = later;
        }
    }
    lds_barrier();
    const unsigned long long t3 = now();
    // copy-out: a quad of lanes per block.  First the images that are full ...
    for (uint32_t qi = tid; qi < 4 * P; qi += PART_BLOCK) {
        const uint32_t b = qi >> 2, w = qi & 3;
        const uint32_t nb = L.goff[b] >> 16, cur = L.cb[b * 16 + 15];
        const u32x4 v = *reinterpret_cast<const u32x4*>(&L.cb[b * 16 + 4 * w]);
        if (nb) {
            if (cur < capb) *reinterpret_cast<u32x4*>(out + (((uint64_t)b * capb + cur) << 6) + 16 * w) = v;
            else if (w < 3) blk_overflow(g, b1, b, v, [&](int q) -> uint32_t { return cb8[b * 64 + 48 + 4 * w + q]; }, ovf_buf, ovf_n, ovf_cap);
        }
    }
    // ... then the staged ones, which follow their sub-bucket's image
    for (uint32_t qi = tid; qi < 4 * total; qi += PART_BLOCK) {
        const uint32_t s = qi >> 2, w = qi & 3;
        const uint32_t b = L.blk_b[s];
        const uint32_t gf = L.goff[b], cur = L.cb[b * 16 + 15];
        const u32x4 lo = *reinterpret_cast<const u32x4*>(&L.st_lo[12 * s + 4 * (w < 3 ? w : 0)]);
        const uint32_t* h = reinterpret_cast<const uint32_t*>(&L.st_hi[12 * s]);
        const u32x4 hv = {h[0], h[1], h[2], 0xFFFFFFFFu};
        const uint32_t dst = cur + 1 + (s - (gf & 0xFFFFu));
        if (dst < capb) *reinterpret_cast<u32x4*>(out + (((uint64_t)b * capb + dst) << 6) + 16 * w) = w < 3 ? lo : hv;
        else if (w < 3) blk_overflow(g, b1, b, lo, [&](int q) -> uint32_t { return L.st_hi[12 * s + 4 * w + q]; }, ovf_buf, ovf_n, ovf_cap);
    }
    lds_barrier();
    // phase C: what is left over takes the image's first places; the run has its blocks
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (br[j] & 0x80000000u) {
            const uint32_t b = (br[j] >> 4) & (MAX_PARTS - 1), v = br[j] & 15u;
            const uint64_t rem = key.r1(j) & g.pl.mr;
            L.cb[b * 16 + v] = (uint32_t)rem; cb8[b * 64 + 48 + v] = (uint8_t)(rem >> 32);
        }
    if (tid < P) { const uint32_t c = L.cb[tid * 16 + 15] + nblk; L.cb[tid * 16 + 15] = c < capb ? c : capb; }
    carry_n = avail - 12 * nblk;
    if (STAMP && st) { const unsigned long long t4 = now(); st[1] += t1 - t0; st[2] += t2 - t1; st[3] += t3 - t2; st[4] += t4 - t3; }
    // (the next tile's first barrier orders phase C before the next ranks are looked at)
}

// what a bucket's last tile left waiting -- up to eleven k-mers per sub-bucket -- as the run's last block, padded with "no item";
// returns (to thread b) the blocks run b holds.  All 1024 lanes must call it.
__device__ __forceinline__ uint32_t flush_carry_blk(P2BLds& L, const PartGeom g, const uint32_t b1, uint8_t* __restrict__ out, uint32_t capb,
                                                    uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap, uint32_t carry_n) {
    const uint32_t tid = threadIdx.x, P = g.P2;
    uint8_t* const cb8 = reinterpret_cast<uint8_t*>(L.cb);
    lds_barrier();                                            // the last tile's phase C is through
    if (tid < P && carry_n) for (uint32_t q = carry_n; q < 12; ++q) { L.cb[tid * 16 + q] = 0xFFFFFFFFu; cb8[tid * 64 + 48 + q] = 0xFF; }
    L.goff[tid] = carry_n;
    lds_barrier();
    for (uint32_t qi = tid; qi < 4 * P; qi += PART_BLOCK) {
        const uint32_t b = qi >> 2, w = qi & 3;
        const uint32_t cn = L.goff[b], cur = L.cb[b * 16 + 15];
        const u32x4 v = *reinterpret_cast<const u32x4*>(&L.cb[b * 16 + 4 * w]);
        if (cn) {
            if (cur < capb) *reinterpret_cast<u32x4*>(out + (((uint64_t)b * capb + cur) << 6) + 16 * w) = v;
            else if (w < 3) blk_overflow(g, b1, b, v, [&](int q) -> uint32_t { return cb8[b * 64 + 48 + 4 * w + q]; }, ovf_buf, ovf_n, ovf_cap);
        }
    }
    lds_barrier();
    uint32_t blocks = 0;
    if (tid < P) { const uint32_t c = L.cb[tid * 16 + 15]; blocks = carry_n && c < capb ? c + 1 : c; }
    return blocks;
}

// first item of bucket b1's runs in the exact edition: every run may end on up to three padding items
__device__ __host__ __forceinline__ uint64_t p2_exact_base(uint64_t beg, uint32_t b1, uint32_t P2) { return (beg + 4 * ((uint64_t)P2 + 1) * b1 + 3) & ~3ULL; }

// The exact edition: one workgroup per level-1 bucket: histogram by digit, scan, scatter.  off2[r] = start of region r's run (a
// multiple of 4: runs begin on group boundaries; the up to three items between a run's last k-mer and the next run are "no item").
template <int HB, bool W1 /* level-1 items of more than 48 bits (HB1 = 4) */, bool L1B = false /* the level-1 buffer holds blocks of ten (kg_l1_blocks.hpp) */>
__global__ void __launch_bounds__(PART_BLOCK)
k_p2(PartGeom g, const uint64_t* __restrict__ l1_off, const uint8_t* __restrict__ l1_buf, uint8_t* __restrict__ l2_buf,
     uint64_t* __restrict__ off2, uint64_t seg_slots, uint64_t* __restrict__ bend /* end of bucket b1's last run: the next bucket's runs start later */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    P2Lds<HB>& L = *reinterpret_cast<P2Lds<HB>*>(lds_raw);
    constexpr int N = L2Fmt<HB>::N;
    const uint32_t tid = threadIdx.x;
    uint64_t beg0, n0, nr0;
    l1_bucket_range(g, l1_off, seg_slots, g.b_lo, beg0, n0, nr0);
    for (uint32_t b1 = g.b_lo + blockIdx.x; b1 < g.b_hi; b1 += gridDim.x) {
        uint64_t beg, n_items, n_real;
        const uint8_t* bucket = l1_buf + l1_bucket_range(g, l1_off, seg_slots, b1, beg, n_items, n_real);
        const uint64_t obeg = p2_exact_base(beg, b1, g.P2) - p2_exact_base(beg0, g.b_lo, g.P2);   // the level-2 buffer holds this pass only
        lds_barrier();
        // pass A histogram in 64 bits (a heavy-hitter k-mer may put more than 2^32 items of a round into one region):
        // the cursor array is free until the scan, so it doubles as the histogram
        unsigned long long* h64 = reinterpret_cast<unsigned long long*>(L.cursor);
        if (tid < MAX_PARTS) h64[tid] = 0;
        lds_barrier();
        for (uint64_t tbeg = 0; tbeg < n_items; tbeg += L2Fmt<HB>::TILE) {
            TileItems<N, W1> key;
            const uint32_t valid = p2_tile_load<N, W1, L1B>(g.hb1, bucket, tbeg, n_items, key);
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (valid >> j & 1) atomicAdd(&h64[place_digit2_of(key.r1(j), g.pl)], 1ULL);
        }
        lds_barrier();
        const uint64_t mine = tid < g.P2 ? h64[tid] : 0;
        const uint64_t excl = block_exclusive_scan64((mine + 3) & ~3ULL, reinterpret_cast<uint64_t*>(L.st_lo));
        if (tid < g.P2) {
            const uint64_t start = obeg + excl;
            L.cursor[tid] = start;
            off2[(uint64_t)b1 * g.P2 + tid] = start;
            for (uint64_t q = mine; q & 3; ++q) l2_put<HB>(l2_buf, start + q, L2Fmt<HB>::NONE);
        }
        if (b1 == g.b_hi - 1 && tid == g.P2 - 1) off2[(uint64_t)g.b_hi * g.P2] = obeg + excl + ((mine + 3) & ~3ULL);
        if (bend && tid == g.P2 - 1) bend[b1] = obeg + excl + ((mine + 3) & ~3ULL);             // the runs of a bucket stop short of the next bucket's
        for (uint64_t tbeg = 0; tbeg < n_items; tbeg += L2Fmt<HB>::TILE) {                     // pass B
            TileItems<N, W1> key;
            const uint32_t valid = p2_tile_load<N, W1, L1B>(g.hb1, bucket, tbeg, n_items, key);
            lds_barrier();
            scatter_tile2<HB, W1>(L, g, key, valid, l2_buf);
        }
    }
}

// ---- level 2 without the histogram pass ----
// The exact k_p2 reads every bucket twice: once to size the P2 sub-runs, once to scatter.  With a uniform hash the sizes
// are known in advance up to noise (n_b / P2 k-mers per region, sigma = sqrt of that), so the fast edition gives every
// region the same capacity -- mean + 1/16 + what the group padding can add (two items per tile on average 1.5) + 16 --
// scatters in ONE pass and reports what it really wrote (cnt2, padding included).  K-mers
// beyond a region's capacity (heavy hitters, or a very unlucky region) go to a small overflow list that the host
// inserts through the direct path; if even that list overflows, the host redoes the round with the exact kernel
// (the level-1 buffer is only read here) and stays exact for the rest of the call.
// Both in ITEMS, both multiples of 4.  p2_out_base(beg + n) - p2_out_base(beg) >= P2 * p2_region_cap(n): buckets do not overlap.
// (`al`: l2_run_align -- 4, or 12 where the items are blocked: runs start and end on block boundaries there; the 48 items of slack per run
// in p2_out_base cover the rounding of either.)
constexpr uint32_t P2_RUN_SLACK = 48;
__device__ __host__ __forceinline__ uint64_t p2_region_cap(uint64_t n_b, uint32_t P2, uint32_t tile, uint32_t al = 4) {
    const uint64_t c = ((n_b + (n_b >> 4)) >> (31 - __builtin_clz(P2))) + 2 * (n_b / tile + 1) + 16;      // (P2 is a power of two: a 64-bit division here kept its reciprocal in two spilled registers)
    return (c + al - 1) / al * al;
}
__device__ __host__ __forceinline__ uint64_t p2_out_base(uint64_t beg, uint32_t b1, uint32_t P2, uint32_t tile, uint32_t al = 4) {
    const uint64_t o = beg + (beg >> 4) + 2 * (uint64_t)P2 * (beg / tile) + (uint64_t)b1 * P2 * P2_RUN_SLACK;
    return (o + al - 1) / al * al;
}

template <int HB, bool W1, bool STAMP = false, bool L1B = false /* the level-1 buffer holds blocks of ten (kg_l1_blocks.hpp) */>
__global__ void __launch_bounds__(PART_BLOCK)
k_p2_fast(PartGeom g, const uint64_t* __restrict__ l1_off, const uint8_t* __restrict__ l1_buf, uint8_t* __restrict__ l2_buf,
          uint64_t* __restrict__ off2, uint32_t* __restrict__ cnt2, uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n,
          uint64_t ovf_cap, uint64_t seg_slots, unsigned long long* __restrict__ stamps = nullptr) {
    // STAMP (diagnostic, KATGPU_P2_STAMP): cycles of wave 0: [0] tile loads, [1] hash + rank, [2] scan, [3] staging, [4] copy-out, [5] tiles
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename P2FastLds<HB>::type Lds;
    Lds& L = *reinterpret_cast<Lds*>(lds_raw);
    constexpr int N = L2Fmt<HB>::N;
    constexpr bool BLK = l2_blocked(HB);                                        // 64-byte blocks of twelve, a quad of lanes per block (scatter_tile2_blk)
    constexpr uint32_t AL = BLK ? L2_BLOCK_ITEMS : 4;
    const uint32_t tid = threadIdx.x;
    uint64_t beg0, n0, nr0;
    l1_bucket_range(g, l1_off, seg_slots, g.b_lo, beg0, n0, nr0);
    for (uint32_t b1 = g.b_lo + blockIdx.x; b1 < g.b_hi; b1 += gridDim.x) {
        uint64_t beg, n_items, n_real;
        const uint8_t* bucket = l1_buf + l1_bucket_range(g, l1_off, seg_slots, b1, beg, n_items, n_real);
        const uint64_t cap = p2_region_cap(n_real, g.P2, L2Fmt<HB>::TILE, AL);
        const uint64_t obase = p2_out_base(beg, b1, g.P2, L2Fmt<HB>::TILE, AL) - p2_out_base(beg0, g.b_lo, g.P2, L2Fmt<HB>::TILE, AL);   // the level-2 buffer holds this pass only
        uint8_t* runs = l2_buf + l2_lo_at<HB>(obase >> 2);                      // run of sub-bucket b: groups [b * capg, (b + 1) * capg) from here (blocked: blocks [b * capg, ...): capg counts blocks)
        const uint32_t capg = (uint32_t)(cap / AL);                             // (< 2^32: host-checked through the buffer's size)
        lds_barrier();
        uint32_t carry_n = 0;                                                   // thread b: k-mers of sub-bucket b that wait for their group / block to fill
        uint32_t tid_b = tid;
        asm volatile("" : "+v"(tid_b));                                        // (per bucket: a per-lane pointer hoisted out of the bucket loop is two registers the tile loop spills)
        if (tid < g.P2) {
            if constexpr (BLK) L.cb[tid * 16 + 15] = 0; else L.cur[tid] = 0;
            off2[(uint64_t)b1 * g.P2 + tid_b] = obase + (uint64_t)tid * cap;
        }
        // Cycle stamps at the bench's shape (round 4; 32.7 K cycles per tile): a third is the wait for the tile's loads, the rest rank 20 %,
        // scan 13 % (before the DPP scan), staging 16 %, copy-out 17 %.  Issuing tile t + 1's loads earlier does not hide that wait: issued
        // before the copy-out (registers free) the stores queue behind them and the copy-out takes what the wait took, and more (180 ms per
        // step against 153); issued before the ranking they do not fit the registers at sixteen items per lane (19 spilled), and at twelve
        // (no spill) the ranking grows by what the wait shrinks (a wave sits at the issue of its loads until the memory pipeline takes
        // them) while the smaller tiles cost 8 ms in padding here and 6 in the apply; issued a piece at a time across all four phases (no
        // wave ever waits at an issue; fits the registers since the block scan stopped keeping sixteen lane masks alive) the wait is gone
        // -- 3 % -- and the copy-out takes 47 % instead of 17: the kernel's time does not change (138-142 ms against 138-150).  What a tile
        // needs is its 98 KB in and ~90 KB out (x 1.3-1.5 in partial sectors) through a memory system that 256 CUs doing the same load to
        // ~4.5 TB/s: wherever the wait is taken, it is for that.
        if constexpr (BLK && !W1) {
            // The block edition, pipelined: tile t + 1's loads are issued inside tile t's routine, right behind its ranking.  (Round 4 measured
            // three ways of doing this and kept none: with a sub-bucket's piece leaving as 20-byte groups the memory system took a tile's 98 KB in
            // and ~90 KB out at its own pace -- 22 ms per launch for the bare pattern, the kernel's own time -- wherever the wait was taken.  Whole
            // lines from a quad of lanes lower that floor to 16.6 ms (tools/ubench_l2_layout.hip): now there is something to overlap with.)
            typename std::conditional<L1B, RawTileB<N>, RawTile<N>>::type raw;
            auto issue = [&](uint64_t at) { if constexpr (L1B) p2_tile_issue_l1b<N>(bucket, at, raw); else p2_tile_issue_narrow<N>(g.hb1, bucket, at, raw); };
            issue(0);
            for (uint64_t tbeg = 0; tbeg < n_items; tbeg += L2Fmt<HB>::TILE) {
                const unsigned long long ta = STAMP ? (unsigned long long)clock64() : 0ULL;
                TileItems<N, W1> key;
                uint32_t valid;
                if constexpr (L1B) valid = p2_tile_decode_l1b<N>(tbeg, n_items, raw, key); else valid = p2_tile_decode_narrow<N>(g.hb1, tbeg, n_items, raw, key);
                if (STAMP) { __builtin_amdgcn_s_waitcnt(0); }
                lds_barrier();
                if (STAMP) { st[0] += (unsigned long long)clock64() - ta; st[5] += 1; }
                // (the last tile's prefetch reads what lies behind the bucket: the next bucket, or the level-2 buffer -- mapped, never decoded)
                scatter_tile2_blk<W1, STAMP>(L, g, b1, key, valid, runs, capg, ovf_buf, ovf_n, ovf_cap, carry_n, st,
                                             [&]() { issue(tbeg + L2Fmt<HB>::TILE); });
            }
        } else
        for (uint64_t tbeg = 0; tbeg < n_items; tbeg += L2Fmt<HB>::TILE) {
            const unsigned long long ta = STAMP ? (unsigned long long)clock64() : 0ULL;
            TileItems<N, W1> key;
            const uint32_t valid = p2_tile_load<N, W1, L1B>(g.hb1, bucket, tbeg, n_items, key);
            if (STAMP) { __builtin_amdgcn_s_waitcnt(0); }
            lds_barrier();
            if (STAMP) { st[0] += (unsigned long long)clock64() - ta; st[5] += 1; }
            if constexpr (BLK) scatter_tile2_blk<W1, STAMP>(L, g, b1, key, valid, runs, capg, ovf_buf, ovf_n, ovf_cap, carry_n, st, []() {});
            else scatter_tile2_fast<HB, W1, STAMP>(L, g, b1, key, valid, runs, capg, ovf_buf, ovf_n, ovf_cap, carry_n, st);
        }
        if constexpr (BLK) {
            const uint32_t blocks = flush_carry_blk(L, g, b1, runs, capg, ovf_buf, ovf_n, ovf_cap, carry_n);
            asm volatile("" : "+v"(tid_b));
            if (tid < g.P2) cnt2[(uint64_t)b1 * g.P2 + tid_b] = blocks * L2_BLOCK_ITEMS;
        } else {
            lds_barrier();
            if constexpr (P2FLds<HB>::CARRY) { flush_carry2<HB>(L, g, b1, runs, capg, ovf_buf, ovf_n, ovf_cap, carry_n); lds_barrier(); }
            if (tid < g.P2) cnt2[(uint64_t)b1 * g.P2 + tid] = L.cur[tid] << 2;
        }
    }
    if (STAMP && tid == 0 && stamps) for (int i = 0; i < 6; ++i) atomicAdd(&stamps[i], st[i]);
}

}  // namespace kg
#include "kg_l2_blocks.hpp"
namespace kg {

// ---- level 3: apply a region's run to the region, in LDS (KV12 tables: keys[S] u64 | counts[S] u32) ----
// A walk with one probe chain per lane inside a divergent loop (round 1's kernel) is bound by dependent LDS round trips, not by LDS
// or VALU throughput (profiles/r01_partitioned_sq_counters.txt: waves parked 67 % of their cycles, LDS array 15 % busy).
// Here a wave takes U k-mers per lane and runs NR probe rounds over all of them in straight-line code: U independent
// ds_read_b64 in flight per wave and one wait per round; a k-mer whose slot holds its key gets a NO-RETURN ds_add and is done.
// What is left after NR rounds -- k-mers that met an EMPTY slot (a new key: needs the CAS claim) or a chain longer than NR --
// goes to a small per-wave queue in LDS (key + slot + remaining probe budget), which the wave drains 64 entries at a time with
// the dependent claim/add loop: dense, and only for the minority that needs it.  Waves never meet at a barrier inside a run.
// No-return adds cannot report a 32-bit wrap, so none may happen: before a walk, counters >= 2^31 give 2^31 to the side table,
// and a walk covers fewer than 2^31 k-mers (a longer run -- one region, one round, exact level 2 only -- is walked in segments).
// Region fill and write-back move 16 bytes per lane and instruction (8- and 4-byte stores were store-issue-bound).
constexpr int AP2_QCAP = 256;                                 // straggler queue entries per wave (12 bytes each)
constexpr int AP2_LANE_PROBES = 12;                           // probes a queue entry gets from its own lane before the wave takes it over
constexpr uint64_t AP2_SEGMENT = 0x7FF00000ULL;               // k-mers per walk: < 2^31

template <int BLOCK, int KP /* 16-byte key loads per lane that cover a region */, int U, int NR, int HB /* high bytes of an item */, bool STAMP = false, bool INLINE_CLAIM = false, bool DYN = true,
          int QCAP = AP2_QCAP /* queue entries per wave: what the region leaves of the LDS */,
          bool TEST_SPILL = false /* honours spill_mod (one k-mer in spill_mod takes the spill path): instantiated for the test suite only */>
__global__ void __launch_bounds__(BLOCK)
k_p3_apply2(DevTable t, PartGeom g, const uint64_t* __restrict__ off2, const uint8_t* __restrict__ l2_buf,
            uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n,
            const uint32_t* __restrict__ cnt2, const uint64_t* __restrict__ bend, unsigned long long* __restrict__ stamps = nullptr, uint32_t spill_mod = 0,
            uint64_t seg_len = AP2_SEGMENT /* k-mers per walk; a multiple of 4 (tests shorten it) */) {
    // STAMP (diagnostic instantiation): cycle stamps of wave 0: [0] fill + sweep, [1] chunk loads + hash, [2] probe rounds, [3] queue push + drains,
    // [4] wait for the other waves, [5] write-back, [6] regions
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NW = BLOCK / 64, CP = (KP + 1) / 2;
    constexpr uint32_t CH = 64 * U;
    static_assert(U == 4, "a lane takes one group of the level-2 buffer: four items");
    const uint32_t S = g.S;                                   // S % 4 == 0 (host-checked): every region is 16-byte aligned in both arrays
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(lds_raw);
    uint32_t* rc = reinterpret_cast<uint32_t*>(lds_raw + (size_t)S * 8);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* wqk = reinterpret_cast<unsigned long long*>(lds_raw + (size_t)S * 12) + (size_t)wave * QCAP;
    uint32_t* wqs = reinterpret_cast<uint32_t*>(lds_raw + (size_t)S * 12 + (size_t)NW * QCAP * 8) + (size_t)wave * QCAP;
    uint32_t new_distinct = 0;
    u32x4 kq[KP], cq[CP];
    __shared__ unsigned long long s_next_chunk;               // chunks of the run are handed out to the waves as they come free

    const uint32_t r_hi = g.b_hi * g.P2;                       // regions [b_lo * P2, r_hi): this launch's
    auto run_end = [&](uint32_t r) -> uint64_t {
        if (bend && (r + 1) % g.P2 == 0) return bend[r / g.P2];
        return off2[r + 1];
    };
    auto next_region = [&](uint32_t from) {
        uint32_t r = from;
        while (r < r_hi && (cnt2 ? cnt2[r] == 0 : off2[r] == run_end(r))) r += gridDim.x;
        return r;
    };
    auto prefetch = [&](uint32_t r) {
        const uint64_t base = (uint64_t)r * S;
#pragma unroll
        for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; kq[u] = *reinterpret_cast<const u32x4*>(t.keys + base + (i < S ? i : 0)); }    // clamped, unconditional: stays in registers
#pragma unroll
        for (int u = 0; u < CP; ++u) { const uint32_t i = (u * BLOCK + tid) * 4; cq[u] = *reinterpret_cast<const u32x4*>(t.counts + base + (i < S ? i : 0)); }
    };

    uint32_t r = next_region(g.b_lo * g.P2 + blockIdx.x);
    if (r < r_hi) prefetch(r);
    while (r < r_hi) {
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
        const uint32_t rd1 = r >> g.l2, rd2 = r & (g.P2 - 1);

        for (uint64_t sbeg = beg; sbeg < end; sbeg += seg_len) {            // one segment, normally
            const uint64_t n_run = (end - sbeg < seg_len ? end - sbeg : seg_len);
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
                            if (--budget == 0) { spill_put(spill, spill_n, g.spill_cap, key); live = false; }     // region full: direct path later
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
                    const unsigned long long ck = lane_value((uint64_t)key, src);            // wave-uniform from here on
                    uint32_t cs = lane_value(slot, src);
                    int cb = (int)lane_value(budget, src);
#pragma unroll 1
                    for (;;) {
                        uint32_t idx = cs + lane; if (idx >= S) idx -= S;                      // S >= 64 on this path (host-checked)
                        const unsigned long long c0 = rk[idx];
                        const unsigned long long mk = __ballot(c0 == ck), me = __ballot(c0 == EMPTY);
                        if (!(mk | me)) {                                                      // 64 foreign keys
                            cb -= 64; cs = cs + 64 >= S ? cs + 64 - S : cs + 64;
                            if (cb <= 0) { if (lane == 0) spill_put(spill, spill_n, g.spill_cap, ck); break; }
                            continue;
                        }
                        const int first = __ffsll((long long)(mk | me)) - 1;
                        if (first >= cb) { if (lane == 0) spill_put(spill, spill_n, g.spill_cap, ck); break; }   // beyond the region's last unprobed slot
                        unsigned long long got = ck;                                           // what the slot holds after this step
                        if (!((mk >> first) & 1)) {                                            // EMPTY comes first: claim it
                            unsigned long long old = EMPTY;
                            if ((int)lane == first) old = atomicCAS(&rk[idx], (unsigned long long)EMPTY, ck);
                            old = lane_value((uint64_t)old, first);
                            if (old == EMPTY) { if ((int)lane == first) ++new_distinct; }
                            else got = old;
                        }
                        if (got == ck) { if ((int)lane == first) add1(idx); break; }
                        cb -= first; cs = cs + first >= S ? cs + first - S : cs + first;       // someone else's key landed there: go on from that slot
                    }
                }
            };

            const uint64_t n_chunks = (n_run + CH - 1) / CH;         // chunk c = groups [64 c, 64 c + 64) of the run
            // a lane's share of a chunk is one GROUP of the buffer (four items: 16 bytes of low words + 4 HB of high parts);
            // the next chunk's group is in flight while this one is worked on
            const uint64_t n_grp = (n_run + 3) >> 2, g0 = sbeg >> 2;      // (runs start on group boundaries and end on "no item" padding)
            unsigned long long cur[U];
            u32x4 c_lo, n_lo;
            typename HiGroup<HB>::type c_hi{}, n_hi{};
            bool c_in, n_in;                                      // the lane's group lies inside the run
            const L2Run<HB> run(l2_buf, g0);
            { const uint64_t gi = (uint64_t)wave * 64 + lane; c_in = gi < n_grp; run.load(c_in ? (uint32_t)gi : 0u, c_lo, c_hi); }
            // (static round-robin left the workgroup waiting ~12 K cycles per region for its slowest wave: the drains vary)
            auto grab = [&]() -> uint64_t {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(&s_next_chunk, 1ULL);
                return lane_value((uint64_t)v, 0);
            };
            for (uint64_t c = wave; c < n_chunks;) {
                const unsigned long long t_a = now();
                const uint64_t c_next = DYN ? grab() : c + NW;
                {                                             // next chunk: in flight behind this one (unconditional loads from a clamped index: a load inside a branch is waited for at the end of the branch)
                    const uint64_t gi = c_next * 64 + lane;
                    n_in = gi < n_grp;
                    run.load(n_in ? (uint32_t)gi : 0u, n_lo, n_hi);
                }
                uint32_t slot[U];
                bool pend[U];                                     // k-mer u still to be placed (lane masks in SGPRs)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t rem = ((uint64_t)hi_of_group<HB>(c_hi, u) << 32) | (u == 0 ? c_lo.x : u == 1 ? c_lo.y : u == 2 ? c_lo.z : c_lo.w);
                    pend[u] = c_in && rem != L2Fmt<HB>::NONE;
                    slot[u] = place_offset(rem, g.pl, S);
                    cur[u] = place_key_d(rd1, rd2, rem, g.pl);
                    if (TEST_SPILL && spill_mod && pend[u] && __umulhi((uint32_t)(mix64(cur[u]) >> 32), spill_mod) == 0) { spill_put(spill, spill_n, g.spill_cap, cur[u]); pend[u] = false; }
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
                unsigned long long pm_u[U];                           // (ballots, not a shuffle scan: see k_p3_apply_pk)
                uint32_t total = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) { pm_u[u] = __ballot(pend[u]); total += (uint32_t)__popcll(pm_u[u]); }
                if (total <= QCAP - 64) {
                    uint32_t at = q_n;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (pend[u]) {
                            const uint32_t i = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm_u[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm_u[u], 0));
                            wqk[i] = cur[u]; wqs[i] = slot[u] | ((S - NR) << 16);
                        }
                        at += (uint32_t)__popcll(pm_u[u]);
                    }
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
                            while (q_n > QCAP - 64) drain_pass(false);
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
                c_lo = n_lo; c_hi = n_hi; c_in = n_in;
                c = c_next;
                if (STAMP) { const unsigned long long t_d = now(); st[1] += t_b - t_a; st[2] += t_c - t_b; st[3] += t_d - t_c; }
            }
            if (sbeg + seg_len < end) lds_barrier();               // another segment follows: no wave may still be grabbing chunks of this one when the counter is reset
        }
        if (rn < r_hi) prefetch(rn);                           // in flight behind the write-back
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

// ---- level 3 for packed tables (kg_device.hpp "P8": one 64-bit word per slot = remainder << cbits | count, 0 = free) ----
// The same walk as k_p3_apply2 on 8-byte slots.  An item of the level-2 buffer IS what a slot stores (the remainder), so nothing is
// decoded: the home slot comes from the remainder, a hit is `word >> cbits == remainder` and ONE no-return ds_add_u64, a claim is
// one compare-and-swap 0 -> remainder | 1 (claim and first count together).  A region of 9344 slots is 73 KB instead of 110: two
// 512-thread workgroups share a CU -- while one fills or writes back its region the other walks -- and the table is swept at 16
// bytes per slot and round instead of 24.  Queue entries are one word: remainder | slot << 44 (remainders have at most 44 bits in a
// packed table); the probes an entry has left follow from its distance to the remainder's home slot.
// No-return adds cannot report a carry out of the count field, so none may happen: a walk finds every counter in 1 .. half (the
// invariant of packed tables, kg_device.hpp: table_add_pk), covers fewer k-mers than half the range (longer runs are walked in
// segments; the host passes seg_len) and hands what it left above half to the side table (keyed by the slot) before it ends.
constexpr int APK_LANE_PROBES = 12;
constexpr uint32_t APK_SLOT_SHIFT = 44;

template <int BLOCK, int KP /* 16-byte loads per lane that cover a region: two slots each */, int HB, bool INLINE_CLAIM = false, bool TEST_SPILL = false,
          bool PF = true /* the next region travels from HBM into registers behind the walk of this one (2 KP VGPRs across the walk); false: loaded when its turn comes -- for shapes with a second workgroup on the CU to cover that */,
          int NR = 3 /* probe rounds before a k-mer goes to the queue */,
          bool STAMP = false /* diagnostic (KATGPU_APPLY_STAMP): wave 0's cycles per phase, added into spill_n[8 ..]: [8] fill, [9] walk, [10] of it drains, [11] wait for the other waves + sweep, [12] write-back, [13] regions, [14] chunks */,
          int UG = 1 /* groups of the level-2 buffer a lane takes per chunk: 4 UG k-mers in flight per lane and probe round.  Two measured
                        171.9 ms per step against 168.3 with one at the bench's shape (same box, round 4): the walk is bound by VALU issue at four waves per
                        SIMD (cycle stamps: 5.3 K cycles per chunk and wave = four waves x ~350 wave instructions x 4 cycles), not by LDS round trips */>
__global__ void __launch_bounds__(BLOCK, 4)    // four waves per SIMD (128 VGPRs): two 512-thread workgroups or one of 1024 threads (a 768-thread shape at six waves spilled: 225 ms against 177)
k_p3_apply_pk(DevTable t, PartGeom g, const uint64_t* __restrict__ off2, const uint8_t* __restrict__ l2_buf,
              uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n, const uint32_t* __restrict__ cnt2, const uint64_t* __restrict__ bend,
              uint32_t qcap /* queue entries per wave: what the region leaves of the LDS; >= 72 */, uint64_t seg_len /* k-mers per walk: a multiple of 4 below half the count range */,
              uint32_t spill_mod,
              uint32_t zero_fill /* the table's first sweep (katgpu_table::zero_from): its slots hold whatever the memory held -- a region starts from zeros
                                    in LDS instead of being loaded, and EVERY region of the launch is visited and written, a run or not */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NW = BLOCK / 64, U = 4 * UG;
    constexpr uint32_t CH = 64 * U;
    const uint32_t S = g.S, cb = g.cbits;                     // S % 4 == 0 (host-checked)
    const uint64_t cmask = pk_cmask(cb), half = pk_half(cb), rem_mask = (1ULL << APK_SLOT_SHIFT) - 1;
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(lds_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* wq = rk + S + (size_t)wave * qcap;
    uint32_t new_distinct = 0;
    u32x4 kq[KP];
    __shared__ unsigned long long s_next_chunk;               // chunks of the run are handed out to the waves as they come free

    const uint32_t r_hi = g.b_hi * g.P2;                       // regions [b_lo * P2, r_hi): this launch's
    auto run_end = [&](uint32_t r) -> uint64_t {
        if (bend && (r + 1) % g.P2 == 0) return bend[r / g.P2];
        return off2[r + 1];
    };
    auto next_region = [&](uint32_t from) {
        uint32_t r = from;
        while (r < r_hi && !zero_fill && (cnt2 ? cnt2[r] == 0 : off2[r] == run_end(r))) r += gridDim.x;
        return r;
    };
    auto prefetch = [&](uint32_t r) {
        if (zero_fill) return;                                 // (nothing to load)
        const uint64_t base = (uint64_t)r * S;
#pragma unroll
        for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; kq[u] = *reinterpret_cast<const u32x4*>(t.keys + base + (i < S ? i : 0)); }    // clamped, unconditional: stays in registers
    };
    auto next_slot = [&](uint32_t s) -> uint32_t { return s + 1 == S ? 0 : s + 1; };
    // +1: the count lives in the low cbits <= 32 bits of the word and the sweep below rules a carry out of it out, so the add is a
    // 32-bit one on the word's low half (a 64-bit LDS atomic costs about twice as much)
    uint32_t* rk32 = reinterpret_cast<uint32_t*>(rk);
    auto add1 = [&](uint32_t slot) { (void)__hip_atomic_fetch_add(&rk32[2 * slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    uint32_t r = next_region(g.b_lo * g.P2 + blockIdx.x);
    if (PF && r < r_hi) prefetch(r);
    while (r < r_hi) {
        const uint64_t beg = off2[r], end = cnt2 ? beg + cnt2[r] : run_end(r);
        const uint64_t base = (uint64_t)r * S;
        const unsigned long long t_fill = now();
        // ---- fill: registers -> LDS ----
        if (!PF) prefetch(r);
        if (zero_fill) {
#pragma unroll
            for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; if (i < S) *reinterpret_cast<u32x4*>(rk + i) = u32x4{0u, 0u, 0u, 0u}; }
        } else {
#pragma unroll
            for (int u = 0; u < KP; ++u) { const uint32_t i = (u * BLOCK + tid) * 2; if (i < S) *reinterpret_cast<u32x4*>(rk + i) = kq[u]; }
        }
        const uint32_t rn = next_region(r + gridDim.x);
        // what spilling a k-mer (region full, test hook) needs: the region's digits give the k-mer back
        const uint32_t rd1 = r >> g.l2, rd2 = r & (g.P2 - 1);
        auto spill_rem = [&](uint64_t rem) { spill_put(spill, spill_n, g.spill_cap, place_key_d(rd1, rd2, rem, g.pl)); };

        for (uint64_t sbeg = beg; sbeg < end; sbeg += seg_len) {            // one segment, normally
            const uint64_t n_run = (end - sbeg < seg_len ? end - sbeg : seg_len);
            if (tid == 0) s_next_chunk = NW;                  // chunks 0 .. NW-1 are the waves' first ones
            lds_barrier();
            const unsigned long long t_walk = now();
            st[0] += t_walk - t_fill;

            // ---- the walk ----
            uint32_t q_n = 0;                                     // entries in this wave's queue (wave-uniform)
            // (Lanes that refill from the queue as they finish, a pass ending when fewer than 16 are busy: measured 141-152 ms against 142 -- the
            // drains' cost is their dependent round trips, not their idle lanes.)
            // One pass over up to 64 queue entries (taken from the tail).  Phase 1, one entry per lane: dependent probes with the
            // claim, while at least 8 lanes are busy and for at most APK_LANE_PROBES probes.  Phase 2: what is left is on a long
            // chain: the WAVE finishes such a k-mer, 64 consecutive slots per read.
            auto drain_pass = [&](bool fin /* nothing will follow: leave no entry behind */) {
                const unsigned long long t_dr = now();
                struct DrainStamp { unsigned long long& acc; unsigned long long t0; bool on; __device__ ~DrainStamp() { if (on) acc += (unsigned long long)clock64() - t0; } } drain_stamp{st[2], t_dr, STAMP};
                const uint32_t take = q_n < 64 ? q_n : 64;
                q_n -= take;
                bool live = lane < take;
                uint64_t rem = 0; uint32_t slot = 0, budget = 0;
                if (live) {
                    const unsigned long long e = wq[q_n + lane];
                    rem = e & rem_mask; slot = (uint32_t)(e >> APK_SLOT_SHIFT);
                    const uint32_t home = place_offset(rem, g.pl, S);
                    budget = S - (slot >= home ? slot - home : slot + S - home);       // probes left before the region has been walked once
                }
#pragma unroll 1
                for (int rr = 0; rr < APK_LANE_PROBES; ++rr) {
                    const int busy = __popcll(__ballot(live));
                    if (busy == 0 || (busy < 8 && (fin || rr >= 4))) break;
                    if (live) {
                        unsigned long long w = rk[slot];
                        if (w == 0) {
                            w = atomicCAS(&rk[slot], 0ULL, (unsigned long long)((rem << cb) | 1ULL));
                            if (w == 0) { ++new_distinct; live = false; }                  // claimed, counted
                        }
                        if (live) {
                            if ((w >> cb) == rem) { add1(slot); live = false; }
                            else {
                                slot = next_slot(slot);
                                if (--budget == 0) { spill_rem(rem); live = false; }       // region full: direct path later
                            }
                        }
                    }
                }
                // the wave takes over what has had its APK_LANE_PROBES (all that is left, when nothing follows); the rest goes back
                const bool lng = live && (fin || S - budget >= (uint32_t)(APK_LANE_PROBES + NR));
                {
                    const bool back = live && !lng;
                    const unsigned long long m = __ballot(back);
                    if (m) {
                        const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                        if (back) wq[at] = rem | ((unsigned long long)slot << APK_SLOT_SHIFT);
                        q_n += (uint32_t)__popcll(m);
                    }
                }
                unsigned long long todo = __ballot(lng);
#pragma unroll 1
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const unsigned long long ck = lane_value((uint64_t)rem, src);            // wave-uniform from here on
                    uint32_t cs = lane_value(slot, src);
                    int cbud = (int)lane_value(budget, src);
#pragma unroll 1
                    for (;;) {
                        uint32_t idx = cs + lane; if (idx >= S) idx -= S;                      // S >= 64 on this path (host-checked)
                        const unsigned long long w = rk[idx];
                        const unsigned long long mk = __ballot(w != 0 && (w >> cb) == ck), me = __ballot(w == 0);
                        if (!(mk | me)) {                                                      // 64 foreign k-mers
                            cbud -= 64; cs = cs + 64 >= S ? cs + 64 - S : cs + 64;
                            if (cbud <= 0) { if (lane == 0) spill_rem(ck); break; }
                            continue;
                        }
                        const int first = __ffsll((long long)(mk | me)) - 1;
                        if (first >= cbud) { if (lane == 0) spill_rem(ck); break; }            // beyond the region's last unprobed slot
                        if ((mk >> first) & 1) { if ((int)lane == first) add1(idx); break; }
                        unsigned long long old = 0;                                            // a free slot comes first: claim it
                        if ((int)lane == first) old = atomicCAS(&rk[idx], 0ULL, (unsigned long long)((ck << cb) | 1ULL));
                        old = lane_value((uint64_t)old, first);
                        if (old == 0) { if ((int)lane == first) ++new_distinct; break; }
                        if ((old >> cb) == ck) { if ((int)lane == first) add1(idx); break; }   // the same k-mer got there first
                        cbud -= first; cs = cs + first >= S ? cs + first - S : cs + first;     // someone else's k-mer landed there: go on from that slot
                    }
                }
            };

            const uint64_t n_chunks = (n_run + CH - 1) / CH;         // chunk c = groups [64 c, 64 c + 64) of the run
            const uint64_t n_grp = (n_run + 3) >> 2, g0 = sbeg >> 2;      // (runs start on group boundaries and end on "no item" padding)
            u32x4 c_lo[UG], n_lo[UG];
            typename HiGroup<HB>::type c_hi[UG], n_hi[UG];
            bool c_in[UG], n_in[UG];                              // the lane's group lies inside the run
            const L2Run<HB> run(l2_buf, g0);
#pragma unroll
            for (int gq = 0; gq < UG; ++gq) {
                const uint64_t gi = ((uint64_t)wave * UG + gq) * 64 + lane;
                c_in[gq] = gi < n_grp; c_hi[gq] = typename HiGroup<HB>::type{}; n_hi[gq] = typename HiGroup<HB>::type{};
                run.load(c_in[gq] ? (uint32_t)gi : 0u, c_lo[gq], c_hi[gq]);
            }
            auto grab = [&]() -> uint64_t {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(&s_next_chunk, 1ULL);
                return lane_value((uint64_t)v, 0);
            };
            for (uint64_t c = wave; c < n_chunks;) {
                const uint64_t c_next = grab();
#pragma unroll
                for (int gq = 0; gq < UG; ++gq) {             // next chunk: in flight behind this one (unconditional loads from a clamped index)
                    const uint64_t gi = (c_next * UG + gq) * 64 + lane;
                    n_in[gq] = gi < n_grp;
                    run.load(n_in[gq] ? (uint32_t)gi : 0u, n_lo[gq], n_hi[gq]);
                }
                uint64_t rem[U];
                uint32_t slot[U];
                bool pend[U];                                     // k-mer u still to be placed (lane masks in SGPRs)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int gq = u >> 2, q = u & 3;
                    rem[u] = ((uint64_t)hi_of_group<HB>(c_hi[gq], q) << 32) | (q == 0 ? c_lo[gq].x : q == 1 ? c_lo[gq].y : q == 2 ? c_lo[gq].z : c_lo[gq].w);
                    pend[u] = c_in[gq] && rem[u] != L2Fmt<HB>::NONE;
                    slot[u] = place_offset(rem[u], g.pl, S);
                    if (TEST_SPILL && spill_mod && pend[u] && __umulhi((uint32_t)(mix64(place_key_d(rd1, rd2, rem[u], g.pl)) >> 32), spill_mod) == 0) { spill_rem(rem[u]); pend[u] = false; }
                }
                // Probe rounds: U reads in flight, one wait; a match adds 1 and is done, a foreign k-mer moves on, a free slot is
                // claimed (inline, or through the queue).  Lanes that are done take no part in the LDS operations.
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    // every lane reads and adds whether its k-mer is still looking or not (a settled one reads its last slot and adds
                    // 0): an exec mask and a branch around each of the eight LDS operations of a round cost more issue slots than the
                    // operations, and issue is what bounds this walk (cycle stamps, round 4)
                    // (Two slots per round -- a slot and its successor in one ds_read2_b64, so that fewer k-mers go through the queue -- measured
                    // 167 ms per step against 154.5: the second compare costs every k-mer more than the drains it saves.)
                    unsigned long long seen[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) seen[u] = rk[slot[u]];
                    bool claim[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool hit = pend[u] && seen[u] != 0 && (seen[u] >> cb) == rem[u];
                        (void)__hip_atomic_fetch_add(&rk32[2 * slot[u]], hit ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        pend[u] = pend[u] && !hit;
                        claim[u] = pend[u] && seen[u] == 0;
                        slot[u] = (pend[u] && !claim[u]) ? next_slot(slot[u]) : slot[u];
                    }
                    // INLINE_CLAIM (a table's first round: nearly every k-mer is new): claims right here, U CAS in flight
                    bool any_claim = false;
#pragma unroll
                    for (int u = 0; u < U; ++u) any_claim = any_claim || claim[u];
                    if (INLINE_CLAIM && __any(any_claim)) {
                        unsigned long long got[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { got[u] = 0; if (claim[u]) got[u] = atomicCAS(&rk[slot[u]], 0ULL, (unsigned long long)((rem[u] << cb) | 1ULL)); }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (claim[u]) {
                                if (got[u] == 0) { ++new_distinct; pend[u] = false; }
                                else if ((got[u] >> cb) == rem[u]) { add1(slot[u]); pend[u] = false; }
                                else slot[u] = next_slot(slot[u]);                             // someone else's k-mer landed there
                            }
                        }
                    }
                }
                // survivors -> queue.  Normal case: one wave-wide prefix sum; a chunk with more survivors than the queue has room for
                // goes in one k-mer column at a time.
                // (ballots, not a shuffle scan: six dependent ds_bpermute -- ~200 cycles each on gfx950, tools/ubench_valu.hip -- were a
                // third of a chunk's time; a ballot and a bit count are a handful of cycles)
                unsigned long long pm_u[U];
                uint32_t total = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) { pm_u[u] = __ballot(pend[u]); total += (uint32_t)__popcll(pm_u[u]); }
                if (UG > 1) while (q_n && q_n + total > qcap) drain_pass(false);    // (twice the k-mers per chunk: make room first)
                if (q_n + total <= qcap) {
                    uint32_t at = q_n;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (pend[u]) wq[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm_u[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm_u[u], 0))] = rem[u] | ((unsigned long long)slot[u] << APK_SLOT_SHIFT);
                        at += (uint32_t)__popcll(pm_u[u]);
                    }
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
                            while (q_n > qcap - 64) drain_pass(false);
                            const uint32_t at = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            if (p) wq[at] = rem[0] | ((unsigned long long)slot[0] << APK_SLOT_SHIFT);
                            q_n += (uint32_t)__popcll(m);
                        }
                        pm >>= 1;
#pragma unroll
                        for (int u = 0; u + 1 < U; ++u) { rem[u] = rem[u + 1]; slot[u] = slot[u + 1]; }
                    }
                }
                const bool last = c_next >= n_chunks;                     // the wave's last chunk empties the queue
                while (q_n > (last ? 0u : 64u)) drain_pass(last);
#pragma unroll
                for (int gq = 0; gq < UG; ++gq) { c_lo[gq] = n_lo[gq]; c_hi[gq] = n_hi[gq]; c_in[gq] = n_in[gq]; }
                c = c_next;
                if (STAMP) st[6] += 1;
            }
            const unsigned long long t_post = now();
            st[1] += t_post - t_walk;
            // ---- back to the invariant (kg_device.hpp: table_add_pk): a slot counts 1 .. half, the rest goes to the side table.  The walk
            // found every counter there and added less than half, so none has carried; each lane looks at the slots it filled. ----
            lds_barrier();                                        // every wave's adds of this segment are in (and no wave grabs chunks any more)
            // (after the run's last segment -- the only one, normally -- the write-back below does this on its way: the same lane, the same slots)
            if (sbeg + seg_len < end) {
#pragma unroll 1
                for (int u = 0; u < KP; ++u) {
                    const uint32_t i = (u * BLOCK + tid) * 2;
                    if (i >= S) break;
                    const u64x2 ww = *reinterpret_cast<const u64x2*>(rk + i);
                    if ((ww.x & cmask) <= half && (ww.y & cmask) <= half) continue;
#pragma unroll 1
                    for (uint32_t j = 0; j < 2; ++j) {
                        const unsigned long long w = rk[i + j];
                        const uint64_t c = w & cmask;
                        if (c <= half) continue;
                        const uint64_t keep = ((c - 1) & (half - 1)) + 1;
                        rk[i + j] = w - (c - keep);
                        ovf_add(t, base + i + j, c - keep);
                    }
                }
            }
            if (STAMP) st[3] += now() - t_post;
        }
        if (PF && rn < r_hi) prefetch(rn);                     // in flight behind the write-back

        // ---- write-back: LDS -> HBM, 16 bytes per lane and store; and back to the invariant on the way ----
        const unsigned long long t_wb = now();
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            const uint32_t i = (u * BLOCK + tid) * 2;
            if (i < S) {
                u64x2 ww = *reinterpret_cast<const u64x2*>(rk + i);
                if ((ww.x & cmask) > half || (ww.y & cmask) > half) {        // (rare: a counter in the upper half of its range)
                    const uint64_t c0 = ww.x & cmask, c1 = ww.y & cmask;
                    if (c0 > half) { const uint64_t keep = ((c0 - 1) & (half - 1)) + 1; ww.x -= c0 - keep; ovf_add(t, base + i, c0 - keep); }
                    if (c1 > half) { const uint64_t keep = ((c1 - 1) & (half - 1)) + 1; ww.y -= c1 - keep; ovf_add(t, base + i + 1, c1 - keep); }
                }
                *reinterpret_cast<u64x2*>(t.keys + base + i) = ww;
            }
        }
        lds_barrier();
        if (STAMP) { st[4] += now() - t_wb; st[5] += 1; }
        r = rn;
    }
    if (STAMP && tid == 0) { for (int i = 0; i < 7; ++i) atomicAdd(&spill_n[8 + i], st[i]); }
    flush_distinct(t, new_distinct);
}

// spilled k-mers (count 1 each) through the direct path.  KV12: checked adds (table_add sees a 32-bit wrap itself): these lists
// are short, and the unchecked table_inc would oblige the host to sweep the whole table first (kg_count.hip: maybe_sweep).
static __global__ void __launch_bounds__(256)
k_insert_keys(DevTable t, const uint64_t* __restrict__ keys, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (t.cbits) table_add_pk<true>(t, keys[i], 1ULL, new_distinct);      // (a heavy hitter's spilled copies all land on one slot: no retry loops)
        else table_add(t, keys[i], 1ULL, new_distinct);
    }
    flush_distinct(t, new_distinct);
}

}  // namespace kg
