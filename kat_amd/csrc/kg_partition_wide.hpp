// kg_partition_wide.hpp -- the partitioned counter for wide tables (33 <= k <= 63; kg_device.hpp "wide keys").
//
// The same three steps as kg_partition.hpp -- radix-partition a round's k-mers by the high digit of their region, split every bucket
// by the low digit, apply every region's run to the region in LDS -- so that a wide k-mer costs streaming traffic instead of the three
// random accesses and the global atomic of k_count_w (11 G k-mers/s whatever the table).  What differs from the one-word counter:
//   * the table's hash (keyw_hash) is not one to one, so an item is the k-mer itself at both levels: its two 63-bit halves, 16 bytes;
//   * a slot is 20 bytes in three arrays (keys, keys_b, counts): a region of 6144 slots fills 120 KB of LDS, one workgroup per CU;
//   * both levels are the EXACT edition (histogram, scan, scatter): no segment or run capacities, no overflow lists.  A workgroup
//     owns its share of every bucket (level 1: offs[w][b] from k_p1_scan; level 2: a bucket is one workgroup's), so an item's place
//     is one returning LDS add on the share's cursor -- no ranking pass, no staging; neighbours of a run are written by the lanes of
//     one tile within microseconds of each other and meet in L2.
// The apply walks a region's run with table_add_w's protocol (claim the first half with a CAS, then the second) on the LDS copy;
// a region without a free slot spills the k-mer to a list the host inserts through the direct kernel after a regrow.
// Replaces the same reference code as k_count_w: mer_iterator + multi-word mer_dna (mer_iterator.hpp:59-89, mer_dna.hpp:235-258)
// and hash_counter::add (hash_counter.hpp:90-113).
#pragma once
#include "kg_partition.hpp"
#include "kg_wide.hpp"

namespace kg {

constexpr int W1_BLOCK = 512;
constexpr int W1_TILE_BYTES = W1_BLOCK * BASES_PER_LANE;                  // 8192 bytes staged per tile
constexpr int W1_TILE_STARTS = W1_TILE_BYTES - WIDE_OVERLAP;              // 8128 window starts per tile (a multiple of 16)
constexpr int W1_LANES_WITH_STARTS = W1_TILE_STARTS / BASES_PER_LANE;     // 508
constexpr uint32_t WIDE_AP_MAX_SLOTS = 7168;                              // 20 B per slot: 140 KB of the 160 KB
constexpr int W3_BLOCK = 1024;
static_assert(BASES_PER_LANE == 16 && W1_TILE_STARTS % 16 == 0, "tiles start on 16-byte boundaries");

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));              // an item: {a, b} of KeyW, one 16-byte access

// Stage one tile (512 lanes x 16 bytes) as 2-bit codes + validity flags.  Ends with a barrier.
__device__ __forceinline__ void w1_stage(uint32_t* s_code, uint32_t* s_bad, const uint8_t* __restrict__ bases, uint64_t n, uint64_t tile) {
    uint32_t w[4];
    p1_tile_load(bases, n, tile * W1_TILE_STARTS, w);                      // (bytes past n read as 'N')
    uint32_t code, bad;
    encode16(w, code, bad);
    __syncthreads();                                                      // the previous tile's windows have been read
    s_code[threadIdx.x] = code;
    s_bad[threadIdx.x] = bad;
    if (threadIdx.x < 4) { s_code[W1_BLOCK + threadIdx.x] = 0; s_bad[W1_BLOCK + threadIdx.x] = 0xFFFF; }
    __syncthreads();
}

// The 16 windows of lane t (starts 16t .. 16t+15 of the tile): f(key) for every valid one, key canonical if the table is.
// The 160-bit register window of k_count_w.
template <class F>
__device__ __forceinline__ void w1_windows(const uint32_t* s_code, const uint32_t* s_bad, uint32_t k, bool canonical, F f) {
    const uint32_t t = threadIdx.x;
    uint64_t hi = ((uint64_t)s_code[t] << 32) | s_code[t + 1];
    uint64_t lo = ((uint64_t)s_code[t + 2] << 32) | s_code[t + 3];
    uint64_t nx = (uint64_t)s_code[t + 4] << 32;
    uint64_t m = ((uint64_t)s_bad[t] << 48) | ((uint64_t)s_bad[t + 1] << 32) | ((uint64_t)s_bad[t + 2] << 16) | s_bad[t + 3];
    uint64_t mn = (uint64_t)s_bad[t + 4] << 48;
    const uint32_t s = 128 - 2 * k, mshift = 64 - k;                      // s: 2 .. 62, mshift: 1 .. 31
#pragma unroll 2
    for (int j = 0; j < BASES_PER_LANE; ++j) {
        if ((m >> mshift) == 0) {
            const uint64_t fhi = hi >> s, flo = (lo >> s) | (hi << (64 - s));
            KeyW key = keyw_from_words(fhi, flo);
            if (canonical) {
                uint64_t rhi, rlo;
                revcomp_words(fhi, flo, k, rhi, rlo);
                if (rhi < fhi || (rhi == fhi && rlo < flo)) key = keyw_from_words(rhi, rlo);
            }
            f(key);
        }
        hi = (hi << 2) | (lo >> 62);
        lo = (lo << 2) | (nx >> 62);
        nx <<= 2;
        m = (m << 1) | (mn >> 63);
        mn <<= 1;
    }
}

// ---- level 1 ----
// SCATTER = false: hist1[w * P1 + b] = k-mers of workgroup w's tiles whose level-1 digit is b (then k_p1_scan: offs, l1_off).
// SCATTER = true: the same tiles again; workgroup w's k-mers of bucket b go to l1_buf[offs[w * P1 + b] ...), in any order.
template <bool SCATTER>
__global__ void __launch_bounds__(W1_BLOCK)
k_w1(DevTable t, uint32_t P1, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
     uint32_t* __restrict__ hist1, const uint64_t* __restrict__ offs, u64x2* __restrict__ l1_buf) {
    __shared__ uint32_t s_code[W1_BLOCK + 4];
    __shared__ uint32_t s_bad[W1_BLOCK + 4];
    __shared__ unsigned long long s_cur[MAX_PARTS];                       // histogram / next free position of this workgroup's share of bucket b
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < MAX_PARTS; b += W1_BLOCK) s_cur[b] = SCATTER && b < P1 ? offs[(uint64_t)blockIdx.x * P1 + b] : 0ULL;
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        w1_stage(s_code, s_bad, bases, n, tile);                           // (its first barrier also covers the initialisation of s_cur)
        if (tid < W1_LANES_WITH_STARTS)
            w1_windows(s_code, s_bad, t.k, t.canonical != 0, [&](KeyW key) {
                const uint32_t b = digit1_of_hash(keyw_hash(key), P1);
                const unsigned long long at = atomicAdd(&s_cur[b], 1ULL);
                if (SCATTER) { u64x2 it; it.x = key.a; it.y = key.b; l1_buf[at] = it; }
            });
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t b = tid; b < P1; b += W1_BLOCK) hist1[(uint64_t)blockIdx.x * P1 + b] = (uint32_t)s_cur[b];
    }
}

// ---- level 2 ----
// One workgroup per level-1 bucket b1 (grid-stride): its run l1_buf[l1_off[b1], l1_off[b1 + 1]) is read twice -- histogram of the
// level-2 digit, then scatter -- and lands in l2_buf over the SAME extent, sorted by region: off2[b1 * P2 + d2] = where region
// (b1, d2)'s run starts; a region's run ends where the next one starts (off2[P1 * P2] = the item count).
static __global__ void __launch_bounds__(PART_BLOCK)
k_w2(uint32_t P1, uint32_t P2, const uint64_t* __restrict__ l1_off, const u64x2* __restrict__ l1_buf, u64x2* __restrict__ l2_buf,
     uint64_t* __restrict__ off2) {
    __shared__ unsigned long long s_cnt[MAX_PARTS];
    __shared__ uint64_t s_scan[PART_BLOCK / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t b1 = blockIdx.x; b1 < P1; b1 += gridDim.x) {
        const uint64_t beg = l1_off[b1], end = l1_off[b1 + 1];
        __syncthreads();                                                  // the previous bucket's cursors are no longer in use
        if (tid < MAX_PARTS) s_cnt[tid] = 0;
        __syncthreads();
        for (uint64_t i = beg + tid; i < end; i += PART_BLOCK) {
            const u64x2 it = l1_buf[i];
            atomicAdd(&s_cnt[digit2_of_hash(keyw_hash(KeyW{it.x, it.y}), P2)], 1ULL);
        }
        __syncthreads();
        const uint64_t mine = tid < P2 ? s_cnt[tid] : 0;
        const uint64_t excl = block_exclusive_scan64(mine, s_scan);       // (barriers inside: every lane has read its count before the cursors are written)
        if (tid < P2) { s_cnt[tid] = beg + excl; off2[(uint64_t)b1 * P2 + tid] = beg + excl; }
        if (b1 == P1 - 1 && tid == 0) off2[(uint64_t)P1 * P2] = end;
        __syncthreads();
        for (uint64_t i = beg + tid; i < end; i += PART_BLOCK) {
            const u64x2 it = l1_buf[i];
            const unsigned long long at = atomicAdd(&s_cnt[digit2_of_hash(keyw_hash(KeyW{it.x, it.y}), P2)], 1ULL);
            l2_buf[at] = it;
        }
    }
}

// ---- level 3: a region's run applied to the region in LDS ----
// LDS: a[S] | b[S] | counts[S].  The walk is table_add_w on the LDS copy: the first half claims a free slot with a CAS, whoever finds
// its first half in a slot installs (or meets) the second half the same way, a match adds one.  A 32-bit wrap of the slot counter
// is seen by the lane whose add returned 2^32 - 1 and goes to the side table (keyed by the slot, as for every wide table).
// A k-mer that finds no slot in its region -- the region is full -- goes to the spill list.
static __global__ void __launch_bounds__(W3_BLOCK)
k_w3_apply(DevTable t, const uint64_t* __restrict__ off2, const u64x2* __restrict__ l2_buf, u64x2* __restrict__ spill, unsigned long long* __restrict__ spill_n,
           uint32_t spill_mod /* tests: one k-mer in spill_mod takes the spill path; 0 = none */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t S = t.region_slots, tid = threadIdx.x;
    unsigned long long* la = reinterpret_cast<unsigned long long*>(lds_raw);
    unsigned long long* lb = la + S;
    uint32_t* lc = reinterpret_cast<uint32_t*>(lb + S);
    uint32_t new_distinct = 0;
    for (uint32_t r = blockIdx.x; r < t.n_regions; r += gridDim.x) {
        const uint64_t beg = off2[r], end = off2[r + 1];
        if (beg == end) continue;                                         // (uniform over the workgroup)
        const uint64_t base = (uint64_t)r * S;
        __syncthreads();                                                  // the previous region has been written back
        for (uint32_t i = tid; i < S; i += W3_BLOCK) { la[i] = t.keys[base + i]; lb[i] = t.keys_b[base + i]; lc[i] = t.counts[base + i]; }
        __syncthreads();
        for (uint64_t i = beg + tid; i < end; i += W3_BLOCK) {
            const u64x2 it = l2_buf[i];
            const KeyW key{it.x, it.y};
            const uint64_t h = keyw_hash(key);
            uint32_t s = offset_of_hash(h, S);
            bool placed = false;
            if (!(spill_mod && (uint32_t)(h >> 7) % spill_mod == 0)) {
                for (uint32_t probe = 0; probe < S; ++probe) {
                    unsigned long long a = la[s];
                    if (a == EMPTY) {
                        a = atomicCAS(&la[s], (unsigned long long)EMPTY, (unsigned long long)key.a);
                        if (a == EMPTY) a = key.a;
                    }
                    if (a == key.a) {
                        unsigned long long b = lb[s];
                        if (b == EMPTY) {
                            b = atomicCAS(&lb[s], (unsigned long long)EMPTY, (unsigned long long)key.b);
                            if (b == EMPTY) { ++new_distinct; b = key.b; }
                        }
                        if (b == key.b) {
                            if (atomicAdd(&lc[s], 1u) == 0xFFFFFFFFu) ovf_add(t, base + s, 1ULL << 32);
                            placed = true;
                            break;
                        }
                    }
                    s = s + 1 == S ? 0 : s + 1;
                }
            }
            if (!placed) spill[atomicAdd(spill_n, 1ULL)] = it;
        }
        __syncthreads();
        for (uint32_t i = tid; i < S; i += W3_BLOCK) { t.keys[base + i] = la[i]; t.keys_b[base + i] = lb[i]; t.counts[base + i] = lc[i]; }
    }
    flush_distinct(t, new_distinct);
}

// spilled k-mers (one occurrence each) through the direct path
static __global__ void __launch_bounds__(256)
k_insert_keys_w(DevTable t, const u64x2* __restrict__ items, uint64_t n) {
    uint32_t new_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64x2 it = items[i];
        table_add_w(t, KeyW{it.x, it.y}, 1, new_distinct);
    }
    flush_distinct(t, new_distinct);
}

}  // namespace kg
