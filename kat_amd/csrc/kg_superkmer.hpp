// kg_superkmer.hpp -- the super-k-mer counter: the partitioned counter of kg_partition.hpp with RUNS of k-mers as its items.
//
// kg_partition.hpp moves every k-mer instance through two radix levels as an 8-byte item: 2.8 x the algorithmic bytes, and three
// kernels that each spend ~130 instructions per k-mer.  Consecutive k-mers of a read overlap in k-1 bases; when they also land in
// the same REGION they can travel together as one short piece of sequence.  That needs a region function that consecutive k-mers
// tend to share: the k-mer's minimizer (kg_device.hpp "minimizer regions": the 16-mer of the k-mer whose canonical form hashes
// lowest; a run of k-mers that share it is a "super-k-mer").  Tables with DevTable::mz set key their regions that way; this file
// counts into them:
//   S1  per lane 16 window starts: order values of the m-mers (recomputed per lane from the 2-bit codes in LDS), sliding minimum,
//       region digits, runs of consecutive valid k-mers with one region, cut at 8 -> ITEMS of 16 bytes: header (n, level-2 digit)
//       + 96 bits of bases (k - 1 + n <= 39 of them).  Items are counting-sorted by the level-1 digit in LDS (4-byte position
//       entries; the payload is cut from the codes at copy-out) and appended to the bucket's segment of this workgroup.
//   S2  one workgroup per level-1 bucket, exact: histogram of the level-2 digit (from the headers), scan, scatter through LDS.
//   S3  one workgroup per region: the region in LDS, each lane expands an item into its k-mers and the probe rounds of k_p3_apply2
//       run over them.
// Items are ~3.5 bytes per k-mer instead of 8 at both levels, levels 1 and 2 do their LDS work per item, and a round holds twice
// the k-mers.  What it costs: every probe outside this file computes a minimizer (kg_device.hpp: region_mz), and regions are only
// as even as the minimizers' weights (a full region parks its k-mers: table_park).
#pragma once
#include "kg_partition.hpp"

namespace kg {

constexpr int SK_MAXN = 8;                                    // k-mers per item
constexpr uint32_t SK_HDR_N = 15;                             // header: bits 0..3 n (0 = padding), bits 4..13 level-2 digit

struct S1Lds {
    uint64_t cursor[MAX_PARTS];
    uint32_t hist[MAX_PARTS];
    uint32_t off[MAX_PARTS];
    uint32_t wave_tot[16];
    uint32_t code[P1_BLOCK + 4];
    uint32_t bad[P1_BLOCK + 4];
    uint32_t pos[P1_TILE_BYTES];          // per staged item: b1 << 16 | (n - 1) << 13 | tile position
    uint16_t b2v[P1_TILE_BYTES];          // its level-2 digit                                   (68 KB in all: two workgroups per CU)
};

// sliding minimum of width W over o[0..31] for the 16 windows that start at 0..15 (W <= 17): log-step doubling, all indices static
template <int W>
__device__ __forceinline__ void sk_slide_min(uint32_t (&o)[32], uint32_t (&mn)[16]) {
    constexpr int P = W >= 16 ? 16 : W >= 8 ? 8 : W >= 4 ? 4 : W >= 2 ? 2 : 1;
#pragma unroll
    for (int s = 1; s < P; s <<= 1) {
#pragma unroll
        for (int i = 0; i + s < 32; ++i) o[i] = o[i] < o[i + s] ? o[i] : o[i + s];     // ascending i: o[i + s] is still the previous step's
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) mn[j] = o[j] < o[j + W - P] ? o[j] : o[j + W - P];
}

// What one lane knows about its 16 window starts: bit j of `valid` = a k-mer starts there; rid[j] = b1 << 16 | b2 of its region;
// `starts` = bit j: an item begins at j.  n of the item that begins at j: sk_item_len.
struct SkLane { uint32_t valid, starts; uint32_t rid[16]; };

__device__ __forceinline__ uint32_t sk_item_len(const SkLane& s, int j) {
    const uint32_t stop = ((s.starts | ~s.valid) & 0xFFFFu) >> (j + 1);
    return (uint32_t)__ffs((int)(stop | (1u << (15 - j))));              // distance to the next item start / invalid start / lane end
}

__device__ __forceinline__ void sk_lane(const uint32_t* code, const uint32_t* bad, uint32_t tid, uint32_t k, bool canonical, uint32_t P1, uint32_t P2,
                                        SkLane& L, uint32_t& ones) {
    const uint32_t m = k < MZ_M ? k : MZ_M, w = k - m + 1;
    uint32_t o[32];
    {
        uint64_t hi = ((uint64_t)code[tid] << 32) | code[tid + 1], lo = ((uint64_t)code[tid + 2] << 32) | code[tid + 3];
        const uint32_t msh = 64 - 2 * m;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            o[i] = mz_value((uint32_t)(hi >> msh), m);
            hi = (hi << 2) | (lo >> 62);
            lo <<= 2;
        }
    }
    uint32_t mn[16];
    switch (w) {                                                           // uniform
        case 1: sk_slide_min<1>(o, mn); break;   case 2: sk_slide_min<2>(o, mn); break;   case 3: sk_slide_min<3>(o, mn); break;
        case 4: sk_slide_min<4>(o, mn); break;   case 5: sk_slide_min<5>(o, mn); break;   case 6: sk_slide_min<6>(o, mn); break;
        case 7: sk_slide_min<7>(o, mn); break;   case 8: sk_slide_min<8>(o, mn); break;   case 9: sk_slide_min<9>(o, mn); break;
        case 10: sk_slide_min<10>(o, mn); break; case 11: sk_slide_min<11>(o, mn); break; case 12: sk_slide_min<12>(o, mn); break;
        case 13: sk_slide_min<13>(o, mn); break; case 14: sk_slide_min<14>(o, mn); break; case 15: sk_slide_min<15>(o, mn); break;
        case 16: sk_slide_min<16>(o, mn); break; default: sk_slide_min<17>(o, mn); break;
    }
    uint64_t mb = ((uint64_t)bad[tid] << 48) | ((uint64_t)bad[tid + 1] << 32) | ((uint64_t)bad[tid + 2] << 16);
    const uint32_t mshift = 64 - k;
    uint32_t valid = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if ((mb >> mshift) == 0) valid |= 1u << j;
        mb <<= 1;
        uint32_t b1, b2;
        mz_digits(mn[j], P1, P2, b1, b2);
        L.rid[j] = (b1 << 16) | b2;
    }
    if (tid >= P1_LANES_WITH_STARTS) valid = 0;                            // the tile's overlap lanes own no starts
    if (k == 32 && !canonical && valid) {                                  // the all-ones k-mer has no slot: tallied (the apply skips it)
        uint64_t hi = ((uint64_t)code[tid] << 32) | code[tid + 1], lo = (uint64_t)code[tid + 2] << 32;
#pragma unroll
        for (int j = 0; j < 16; ++j) { if ((valid >> j & 1) && hi == EMPTY) ++ones; hi = (hi << 2) | (lo >> 62); lo <<= 2; }
    }
    uint32_t starts = 0, len = 0, prev = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (valid >> j & 1) {
            if (len == 0 || L.rid[j] != prev || len == SK_MAXN) { starts |= 1u << j; len = 1; } else ++len;
            prev = L.rid[j];
        } else len = 0;
    }
    L.valid = valid; L.starts = starts;
}

// the 96 bits of bases that start at tile position p, from the staged codes
__device__ __forceinline__ void sk_payload(const uint32_t* code, uint32_t p, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    const uint32_t q = p >> 4, sh = 2 * (p & 15);
    const uint32_t x0 = code[q], x1 = code[q + 1], x2 = code[q + 2], x3 = code[q + 3];
    if (sh) { w1 = (x0 << sh) | (x1 >> (32 - sh)); w2 = (x1 << sh) | (x2 >> (32 - sh)); w3 = (x2 << sh) | (x3 >> (32 - sh)); }
    else { w1 = x0; w2 = x1; w3 = x2; }
}

__device__ __forceinline__ void s1_stage(S1Lds& L, const uint32_t (&wd)[4]) {
    const uint32_t tid = threadIdx.x;
    uint32_t code, bad;
    encode16(wd, code, bad);
    L.code[tid] = code;
    L.bad[tid] = bad;
    if (tid < 4) { L.code[P1_BLOCK + tid] = 0; L.bad[P1_BLOCK + tid] = 0xFFFF; }
    lds_barrier();
}

// MODE 0: count -- hist1[w * P1 + b] = items of workgroup w for bucket b (+ kmers[w] = valid k-mers of workgroup w: the round sizing
//         wants items per start and the walk wants k-mers); MODE 1: exact scatter (offs from the count + k_p1_scan);
// MODE 2: segmented scatter (as k_p1v2_scatter<true>: segment (b, w) = seg_cap items at (b * gridDim + w) * seg_cap, overflow list,
//         padding with n = 0 items).
template <int MODE>
__global__ void __launch_bounds__(P1_BLOCK, 4)                // two workgroups per CU: four waves per SIMD, 128 VGPRs
k_s1(DevTable t, PartGeom g, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
     uint32_t* __restrict__ hist1, unsigned long long* __restrict__ kmers, const uint64_t* __restrict__ offs, u32x4* __restrict__ l1_items,
     uint64_t seg_cap, u32x4* __restrict__ ovf_items, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap) {
    __shared__ __attribute__((aligned(16))) S1Lds L;
    const uint32_t tid = threadIdx.x, P = g.P1, k = t.k;
    const bool canonical = t.canonical != 0;
    uint32_t ones = 0, n_kmers = 0;
    auto seg_base = [&](uint32_t b) -> uint64_t { return ((uint64_t)b * gridDim.x + blockIdx.x) * seg_cap; };
    if (MODE == 0) { for (uint32_t b = tid; b < MAX_PARTS; b += P1_BLOCK) L.hist[b] = 0; }
    else for (uint32_t b = tid; b < P; b += P1_BLOCK) L.cursor[b] = MODE == 2 ? seg_base(b) : offs[(uint64_t)blockIdx.x * P + b];
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    uint32_t wd[4], wn[4];
    if (t0 < t1) p1_tile_load(bases, n, t0 * P1_TILE_STARTS, wd);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) p1_tile_load(bases, n, (tile + 1) * P1_TILE_STARTS, wn);
        lds_barrier();                                   // previous tile's copy-out / cursor update done
        if (MODE != 0) for (uint32_t b = tid; b < MAX_PARTS; b += P1_BLOCK) L.hist[b] = 0;
        s1_stage(L, wd);                                 // ends with a barrier
#pragma unroll
        for (int q = 0; q < 4; ++q) wd[q] = wn[q];
        SkLane sl;
        sk_lane(L.code, L.bad, tid, k, canonical, P, g.P2, sl, ones);
        if (MODE == 0) {
            n_kmers += (uint32_t)__popc(sl.valid);
#pragma unroll
            for (int j = 0; j < 16; ++j) if (sl.starts >> j & 1) atomicAdd(&L.hist[sl.rid[j] >> 16], 1u);
            continue;
        }
        uint32_t rk[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { rk[j] = 0; if (sl.starts >> j & 1) rk[j] = atomicAdd(&L.hist[sl.rid[j] >> 16], 1u); }
        lds_barrier();
        uint32_t e0, e1;
        p1_scan_pair(tid < P ? L.hist[tid] : 0, tid + P1_BLOCK < P ? L.hist[tid + P1_BLOCK] : 0, L.wave_tot, e0, e1);
        L.off[tid] = e0;
        L.off[tid + P1_BLOCK] = e1;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (sl.starts >> j & 1) {
                const uint32_t b1 = sl.rid[j] >> 16, at = L.off[b1] + rk[j];
                L.pos[at] = (b1 << 16) | ((sk_item_len(sl, j) - 1) << 13) | (tid * 16 + j);
                L.b2v[at] = (uint16_t)(sl.rid[j] & 0xFFFF);
            }
        lds_barrier();
        const uint32_t total = L.off[P - 1] + L.hist[P - 1];
        for (uint32_t idx = tid; idx < total; idx += P1_BLOCK) {
            const uint32_t e = L.pos[idx], b = e >> 16;
            u32x4 it;
            it.x = (((e >> 13) & 7) + 1) | ((uint32_t)L.b2v[idx] << 4);
            uint32_t w1, w2, w3;
            sk_payload(L.code, e & 0x1FFF, w1, w2, w3);
            it.y = w1; it.z = w2; it.w = w3;
            const uint64_t dst = L.cursor[b] + (idx - L.off[b]);
            if (MODE != 2 || dst < seg_base(b) + seg_cap) l1_items[dst] = it;
            else {                                                             // the segment is full: the overflow list
                const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                if (at < ovf_cap) ovf_items[at] = it;
            }
        }
        lds_barrier();
        for (uint32_t b = tid; b < P; b += P1_BLOCK) {
            uint64_t c = L.cursor[b] + L.hist[b];
            if (MODE == 2) { const uint64_t lim = seg_base(b) + seg_cap; c = c < lim ? c : lim; }
            L.cursor[b] = c;
        }
    }
    lds_barrier();
    if (MODE == 0) {
        for (uint32_t b = tid; b < P; b += P1_BLOCK) hist1[(uint64_t)blockIdx.x * P + b] = L.hist[b];
        for (int off = 32; off > 0; off >>= 1) n_kmers += __shfl_down(n_kmers, off, 64);
        if ((tid & 63) == 0 && n_kmers) atomicAdd(&kmers[blockIdx.x], (unsigned long long)n_kmers);
    }
    if (MODE == 2) {
        const uint32_t grp = tid >> 4, l16 = tid & 15;
        const u32x4 pad = {0, 0, 0, 0};
        for (uint32_t b = grp; b < P; b += P1_BLOCK / 16) {
            const uint64_t lim = seg_base(b) + seg_cap;
            for (uint64_t i = L.cursor[b] + l16; i < lim; i += 16) l1_items[i] = pad;
        }
    }
    if (MODE != 1) {                                                          // the exact scatter follows a count pass that tallied already
        for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
        if ((tid & 63) == 0 && ones) atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)ones);
    }
}

// ---- level 2: one workgroup per level-1 bucket, exact: histogram of the items' level-2 digit, scan, scatter through LDS ----
constexpr int S2_ITEMS = 4;                                   // items per lane and tile
constexpr int S2_TILE = PART_BLOCK * S2_ITEMS;                // 4096 items = 64 KB of staging
struct S2Lds {
    u32x4 staging[S2_TILE];
    uint64_t cursor[MAX_PARTS];
    uint32_t hist[MAX_PARTS];                                 // the bucket's total per level-2 digit
    uint32_t thist[MAX_PARTS];                                // this tile's
    uint32_t toff[MAX_PARTS];
    uint32_t wave_tot[32];
};

__device__ __forceinline__ u32x4 s2_load(const u32x4* __restrict__ items, uint64_t i, uint64_t beg, uint64_t end) {
    u32x4 v = items[i < end ? i : beg];                       // unconditional load from a clamped index
    if (i >= end) v.x = 0;
    return v;
}

__global__ void __launch_bounds__(PART_BLOCK)
k_s2(PartGeom g, const uint64_t* __restrict__ l1_off, uint64_t seg_slots, const u32x4* __restrict__ l1_items, u32x4* __restrict__ l2_items,
     uint64_t* __restrict__ off2, uint32_t* __restrict__ cnt2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    S2Lds& L = *reinterpret_cast<S2Lds*>(lds_raw);
    const uint32_t tid = threadIdx.x, P = g.P2;
    for (uint32_t b1 = blockIdx.x; b1 < g.P1; b1 += gridDim.x) {
        uint64_t beg, end;
        l1_bucket_range(l1_off, seg_slots, b1, beg, end);
        lds_barrier();
        if (tid < MAX_PARTS) L.hist[tid] = 0;
        lds_barrier();
        for (uint64_t i0 = beg; i0 < end; i0 += (uint64_t)S2_TILE) {            // pass A
            u32x4 v[S2_ITEMS];
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j) v[j] = s2_load(l1_items, i0 + (uint64_t)j * PART_BLOCK + tid, beg, end);
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j) if (v[j].x & SK_HDR_N) atomicAdd(&L.hist[(v[j].x >> 4) & 1023], 1u);
        }
        lds_barrier();
        uint32_t total;
        const uint32_t mine = tid < P ? L.hist[tid] : 0;
        const uint32_t excl = block_exclusive_scan(mine, L.wave_tot, &total);
        if (tid < P) {
            L.cursor[tid] = beg + excl;                        // the level-2 buffer mirrors the level-1 layout: a bucket's runs fit its extent
            off2[(uint64_t)b1 * P + tid] = beg + excl;
            cnt2[(uint64_t)b1 * P + tid] = mine;
        }
        for (uint64_t tbeg = beg; tbeg < end; tbeg += (uint64_t)S2_TILE) {     // pass B
            lds_barrier();
            if (tid < MAX_PARTS) L.thist[tid] = 0;
            u32x4 v[S2_ITEMS];
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j) v[j] = s2_load(l1_items, tbeg + (uint64_t)j * PART_BLOCK + tid, beg, end);
            lds_barrier();
            uint32_t rk[S2_ITEMS];
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j) { rk[j] = 0; if (v[j].x & SK_HDR_N) rk[j] = atomicAdd(&L.thist[(v[j].x >> 4) & 1023], 1u); }
            lds_barrier();
            uint32_t ttotal;
            const uint32_t tm = tid < P ? L.thist[tid] : 0;
            const uint32_t te = block_exclusive_scan(tm, L.wave_tot, &ttotal);
            if (tid < MAX_PARTS) L.toff[tid] = te;
            lds_barrier();
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j) if (v[j].x & SK_HDR_N) L.staging[L.toff[(v[j].x >> 4) & 1023] + rk[j]] = v[j];
            lds_barrier();
            for (uint32_t idx = tid; idx < ttotal; idx += PART_BLOCK) {
                const u32x4 it = L.staging[idx];
                const uint32_t b = (it.x >> 4) & 1023;
                l2_items[L.cursor[b] + (idx - L.toff[b])] = it;
            }
            lds_barrier();
            if (tid < P) L.cursor[tid] += L.thist[tid];
        }
    }
}


// ---- level 3: the apply of k_p3_apply2 (kg_partition.hpp) over ITEMS: a lane takes one item, expands it into its (up to 8) k-mers,
// and the probe rounds, the per-wave queues and the wave-cooperative finish run as they do there ----
constexpr uint64_t S3_SEGMENT = 0x0FF00000ULL;               // items per walk: < 2^31 k-mers

template <int U>
__device__ __forceinline__ void sk_expand(u32x4 it, uint32_t k, bool canonical, unsigned long long (&cur)[U]) {
    const uint32_t n = it.x & SK_HDR_N;
    uint64_t hi = ((uint64_t)it.y << 32) | it.z;
    uint32_t lo = it.w;
    const uint32_t ksh = 64 - 2 * k;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint64_t fwd = hi >> ksh;
        uint64_t key = fwd;
        if (canonical) { const uint64_t rc = kmer_revcomp(fwd, k); key = rc < fwd ? rc : fwd; }
        cur[j] = (uint32_t)j < n ? key : EMPTY;                // (the all-ones k-mer -- k = 32, not canonical -- reads as padding: level 1 tallied it)
        hi = (hi << 2) | (lo >> 30);
        lo <<= 2;
    }
}

template <int BLOCK, int KP /* 16-byte key loads per lane that cover a region */, int U, int NR, bool STAMP = false, bool INLINE_CLAIM = false, bool DYN = true>
__global__ void __launch_bounds__(BLOCK)
k_s3_apply(DevTable t, PartGeom g, const uint64_t* __restrict__ off2, const u32x4* __restrict__ l2_items,
            uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_n,
            const uint32_t* __restrict__ cnt2, const uint64_t* __restrict__ bend, unsigned long long* __restrict__ stamps = nullptr) {
    // STAMP (diagnostic instantiation): cycle stamps of wave 0: [0] fill + sweep, [1] chunk loads + hash, [2] probe rounds, [3] queue push + drains,
    // [4] wait for the other waves, [5] write-back, [6] regions
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? (unsigned long long)clock64() : 0ULL; };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NW = BLOCK / 64, CP = (KP + 1) / 2;
    static_assert(U == SK_MAXN, "one item per lane: U k-mers");
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

        for (uint64_t sbeg = beg; sbeg < end; sbeg += S3_SEGMENT) {        // one segment, normally
            const uint64_t n_run = (end - sbeg < S3_SEGMENT ? end - sbeg : S3_SEGMENT);      // items
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

            const uint64_t n_chunks = (n_run + 63) / 64;                      // a chunk = 64 items, one per lane
            u32x4 item, item_n;
            { const uint64_t i = (uint64_t)wave * 64 + lane; item = l2_items[sbeg + (i < n_run ? i : 0)]; if (i >= n_run) item.x = 0; }
            // (static round-robin left the workgroup waiting ~12 K cycles per region for its slowest wave: the drains vary)
            auto grab = [&]() -> uint64_t {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(&s_next_chunk, 1ULL);
                return __shfl(v, 0, 64);
            };
            for (uint64_t c = wave; c < n_chunks;) {
                const unsigned long long t_a = now();
                const uint64_t c_next = DYN ? grab() : c + NW;
                { const uint64_t i = c_next * 64 + lane; item_n = l2_items[sbeg + (i < n_run ? i : 0)]; if (i >= n_run) item_n.x = 0; }   // next chunk: in flight behind this one
                unsigned long long cur[U];
                sk_expand<U>(item, t.k, t.canonical != 0, cur);
                uint32_t slot[U];
                bool pend[U];                                     // k-mer u still to be placed (lane masks in SGPRs)
#pragma unroll
                for (int u = 0; u < U; ++u) { slot[u] = offset_of_hash(mix64(cur[u]), S); pend[u] = cur[u] != EMPTY; }
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
                item = item_n;
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


// items that found no room in their level-1 segment (a few), through the direct path: every k-mer of the item, count 1 each
__global__ void __launch_bounds__(256)
k_insert_items(DevTable t, const u32x4* __restrict__ items, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long cur[SK_MAXN];
        sk_expand<SK_MAXN>(items[i], t.k, t.canonical != 0, cur);
#pragma unroll 1
        for (int j = 0; j < SK_MAXN; ++j) if (cur[j] != EMPTY) table_add(t, cur[j], 1ULL, new_distinct);
    }
    flush_distinct(t, new_distinct);
}

}  // namespace kg
