// kg_l2_blocks.hpp -- level 2 of the bench's shape (5-byte items out of level 1's blocks of ten), by level 1's block edition's recipe
// (kg_l1_blocks.hpp): every finished block of a tile has a pool image of its own in LDS, a sub-bucket's waiting items are copied in front of
// its first one, what is left over goes straight into the -- now free -- waiting image; ONE placing pass, branch-free; and a tile's blocks
// leave between the NEXT tile's rankings, a quad of lanes per 64-byte line, so that nothing of the workgroup ever waits for a store.
//
// Why (cycle stamps of k_p2_fast's block edition, 26 K cycles per 16 K-item tile): the vector-memory counter is in order, so the wait for
// a tile's loads -- requested before the tile before's copy-out, as they have to be -- was a wait for that copy-out's ~1400 lines to be
// acknowledged (12 %); the copy-out itself parks every wave behind a CU's ~10 bytes a clock of stores (15 %); and the three-way placing
// with its second pass for the left-overs (30 %) kept a tile's items in registers across all of it.
//
// LDS: 1024 waiting images (64 KB) + 1252 pool images (78 KB) + the per-sub-bucket words: 14 items per lane, a 14 336-item tile (1195
// blocks when every item is a k-mer).  The output format is k_p2_fast's: runs of 64-byte blocks of twelve, low words 0 .. 11, high bytes
// at 48 .. 59, a run's capacity fixed, what does not fit to the overflow list (the host's exact edition behind it).
#pragma once

namespace kg {

constexpr int L2X_N = 14;
constexpr uint32_t L2X_TILE = L2X_N * PART_BLOCK;        // 14336
constexpr uint32_t L2X_POOL = 1252;                    // (all the LDS there is: a tile of k-mers only makes 1194.7 blocks, +- 13 with what waits -- 4.4 sigma; beyond: the overflow list)

struct P2XLds {
    uint32_t hist[MAX_PARTS + 64];      // next rank in each sub-bucket's line-up (the waiting items first) + one dump counter per lane of a wave
    uint2 gb[MAX_PARTS];                // per sub-bucket and tile: x = ranks below this are placed in the pool | ranks from this on are left over << 16; y = image of its first block
    uint32_t where[L2X_POOL];           // pool image s leaves for block where[s] of the bucket's runs
    uint32_t wave_tot[32];
    __attribute__((aligned(64))) uint32_t img[(MAX_PARTS + L2X_POOL) * 16];      // image B < 1024: sub-bucket B's waiting items (dword 15: blocks its run holds | items that wait << 28); 1024 + s: pool image s
};
static_assert(sizeof(P2XLds) <= 160 * 1024 - 256 && offsetof(P2XLds, img) % 64 == 0, "LDS");
static_assert(PART_BLOCK == MAX_PARTS, "thread b is sub-bucket b");

template <bool STAMP = false /* diagnostic (KATGPU_P2_STAMP): wave 0's cycles per phase, summed over the workgroups into stamps[0 .. 9] */>
__global__ void __launch_bounds__(PART_BLOCK)
k_p2x_fast(PartGeom g, const uint64_t* __restrict__ l1_off, const uint8_t* __restrict__ l1_buf, uint8_t* __restrict__ l2_buf,
           uint64_t* __restrict__ off2, uint32_t* __restrict__ cnt2, uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n,
           uint64_t ovf_cap, uint64_t seg_slots, unsigned long long* __restrict__ stamps) {
    constexpr int N = L2X_N;
    constexpr uint32_t PB = MAX_PARTS, TILE_CAP = L2Fmt<1>::TILE /* the capacity formulas' tile: the host's */;
    extern __shared__ __attribute__((aligned(64))) unsigned char lds_raw[];
    P2XLds& L = *reinterpret_cast<P2XLds*>(lds_raw);
    unsigned char* const img8 = reinterpret_cast<unsigned char*>(L.img);
    const uint32_t tid = threadIdx.x, P = g.P2;
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;   // STAMP: [0] wait for the tile + decode, [1] its barrier, [2] ranking (+ the blocks out), [3] barrier, [4] per sub-bucket, [5] barrier, [6] placing, [9] tiles
    auto stamp = [&](int i) { if (STAMP) { const unsigned long long x = (unsigned long long)clock64(); st[i] += x - tq; tq = x; } };
    uint64_t beg0, n0, nr0;
    l1_bucket_range(g, l1_off, seg_slots, g.b_lo, beg0, n0, nr0);
    uint32_t tid_o = tid;
    for (uint32_t b1 = g.b_lo + blockIdx.x; b1 < g.b_hi; b1 += gridDim.x) {
        uint64_t beg, n_items, n_real;
        const uint8_t* bucket = l1_buf + l1_bucket_range(g, l1_off, seg_slots, b1, beg, n_items, n_real);
        const uint64_t cap = p2_region_cap(n_real, P, TILE_CAP, L2_BLOCK_ITEMS);
        const uint64_t obase = p2_out_base(beg, b1, P, TILE_CAP, L2_BLOCK_ITEMS) - p2_out_base(beg0, g.b_lo, P, TILE_CAP, L2_BLOCK_ITEMS);   // the level-2 buffer holds this pass only
        uint8_t* const runs = l2_buf + l2_lo_at<1>(obase >> 2);                 // run of sub-bucket b: blocks [b * capb, (b + 1) * capb) from here
        const uint32_t capb = (uint32_t)(cap / L2_BLOCK_ITEMS);                 // (< 2^32: host-checked through the buffer's size)
        lds_barrier();                                                          // the bucket before is through with the carve
        asm volatile("" : "+v"(tid_o));
        L.hist[tid_o] = 0; L.gb[tid_o] = uint2{0u, 0u};
        if (tid < 64) L.hist[PB + tid_o] = 0;
        L.img[tid_o * 16 + 15] = 0;
        if (tid < P) off2[(uint64_t)b1 * P + tid_o] = obase + (uint64_t)tid * cap;
        uint32_t n_out = 0;                                                     // pool images the tile before filled
        // ---- blocks out (of the tile before): a quad of lanes per block, sixteen bytes each, in steps the ranking takes between its items ----
        constexpr int OUT_STEPS = (L2X_POOL + PART_BLOCK / 4 - 1) / (PART_BLOCK / 4);
        static_assert(OUT_STEPS <= N / 2, "a step per pair of the ranking's items");
        struct OutStep { u32x4 v; uint32_t where; };
        auto out_read = [&](int it, OutStep& o) {
            asm volatile("" : "+v"(tid_o));
            const uint32_t s = min((tid_o >> 2) + (uint32_t)it * (PART_BLOCK / 4), L2X_POOL - 1);
            o.where = L.where[s];
            o.v = *reinterpret_cast<const u32x4*>(&L.img[(PB + s) * 16 + (tid_o & 3) * 4]);
        };
        auto out_store = [&](int it, const OutStep& o) {
            asm volatile("" : "+v"(tid_o));
            if ((tid_o >> 2) + (uint32_t)it * (PART_BLOCK / 4) < n_out)
                *reinterpret_cast<u32x4*>(runs + (((uint64_t)o.where << 6) | ((tid_o & 3) * 16))) = o.v;
        };
        RawTileB<N> raw;
        p2_tile_issue_l1b<N>(bucket, 0, raw);
        for (uint64_t tbeg = 0; tbeg < n_items; tbeg += L2X_TILE) {
            if (STAMP) tq = (unsigned long long)clock64();
            TileItems<N, false> key;
            const uint32_t valid = p2_tile_decode_l1b<N>(tbeg, n_items, raw, key);
            {                                                                   // (the tile's items have arrived -- waited for HERE, with nothing younger than a tile in flight)
#pragma unroll
                for (int j = 0; j < N; ++j) asm volatile("" : "+v"(key.lo[j]));
#pragma unroll
                for (int j = 0; j < N / 2; ++j) asm volatile("" : "+v"(key.hi[j]));
            }
            // (the last tile's request reads what lies behind the bucket: the next bucket, or the level-2 buffer -- mapped, never decoded)
            p2_tile_issue_l1b<N>(bucket, tbeg + L2X_TILE, raw);
            stamp(0);
            lds_barrier();                                                      // the tile before has been placed; the counters hold what waits
            stamp(1);
            // ---- ranking: digit and rank behind the sub-bucket's waiting items (straight-line: a slot without a k-mer ranks in a dump counter) ----
            uint32_t br[N];                                                     // digit << 16 | rank
            {
                uint32_t rk[N];
                const uint32_t dump = PB + (tid & 63);
                OutStep o;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (j % 2 == 0 && j / 2 < OUT_STEPS) out_read(j / 2, o);    // the tile before leaves: a step's reads before one item, its store behind the next
                    const uint32_t b = place_digit2_of(key.r1(j), g.pl) & (PB - 1);
                    rk[j] = atomicAdd(&L.hist[(valid >> j & 1) ? b : dump], 1u);
                    br[j] = b << 16;
                    if (j >= 2) br[j - 2] |= rk[j - 2];
                    if (j % 2 == 1 && j / 2 < OUT_STEPS) out_store(j / 2, o);
                }
#pragma unroll
                for (int j = N - 2; j < N; ++j) br[j] |= rk[j];
            }
            stamp(2);
            lds_barrier();
            stamp(3);
            // ---- per sub-bucket: how many blocks complete, what the run and the pool have room for; the waiting items move in front of the first ----
            asm volatile("" : "+v"(tid_o));
            {
                uint32_t want = 0, avail = 0, nblk = 0, cur = 0, cn_old = 0;
                if (tid < P) {
                    avail = L.hist[tid_o];
                    const uint32_t w15 = L.img[tid_o * 16 + 15];
                    cur = w15 & 0x0FFFFFFFu; cn_old = w15 >> 28;
                    nblk = __umulhi(avail, 0xAAAAAAABu) >> 3;                   // avail / 12
                    want = min(nblk, capb > cur ? capb - cur : 0u);
                }
                const uint32_t lane = tid_o & 63, wave = tid_o >> 6;
                const uint32_t inc = wave_inclusive_scan(want);
                if (lane == 63) L.wave_tot[wave] = inc;
                lds_barrier();
                const uint32_t sc = wave_inclusive_scan(lane < PART_BLOCK / 64 ? L.wave_tot[lane] : 0u);
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
                const uint32_t excl = (wv ? lane_value(sc, (int)wv - 1) : 0u) + inc - want;
                { const uint32_t tot = lane_value(sc, PART_BLOCK / 64 - 1); n_out = tot < L2X_POOL ? tot : L2X_POOL; }
                if (tid < P) {
                    const uint32_t nst = excl < L2X_POOL ? min(want, L2X_POOL - excl) : 0u;     // blocks that leave
                    // Placed: ranks below 12 nst.  Left over (into the waiting image, free once its items have moved): ranks from 12 nblk on.  What lies
                    // between has no room and takes the overflow list -- and when NO block can leave although one is complete (a full run, an
                    // exhausted pool), all of the tile's k-mers of the sub-bucket do, and what waited still waits.
                    const bool stuck = nblk != 0 && nst == 0;
                    L.gb[tid_o] = uint2{(L2_BLOCK_ITEMS * nst) | ((stuck ? 0xFFFFu : L2_BLOCK_ITEMS * nblk) << 16), PB + excl};
                    if (nst) {
                        const u32x4* src = reinterpret_cast<const u32x4*>(&L.img[tid_o * 16]);
                        u32x4* dst = reinterpret_cast<u32x4*>(&L.img[(PB + excl) * 16]);
                        const u32x4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
                        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
                        const uint32_t first = tid_o * capb + cur;
                        for (uint32_t q = 0; q < nst; ++q) L.where[excl + q] = first + q;
                    }
                    const uint32_t cn = stuck ? cn_old : avail - L2_BLOCK_ITEMS * nblk;
                    L.img[tid_o * 16 + 15] = (cur + nst) | (cn << 28);
                    L.hist[tid_o] = cn;                                         // the next tile's k-mers rank behind what waits
                } else L.hist[tid_o] = 0;
                if (tid < 64) L.hist[PB + tid_o] = 0;                           // (the dump counters: a rank is kept in sixteen bits)
            }
            stamp(4);
            lds_barrier();
            stamp(5);
            // ---- placing: branch-free but for the two writes.  Rank v = 12 q + s below gb.x's low half: slot s of pool image gb.y + q; from
            // its high half on (q = nblk there): slot s of the sub-bucket's waiting image. ----
            {
                uint32_t lostm = 0;
                constexpr int SB = 2;
#pragma unroll
                for (int j0 = 0; j0 < N; j0 += SB) {
                    uint2 G[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) G[u] = L.gb[br[j0 + u] >> 16];        // (a slot without a k-mer has a digit all the same: some sub-bucket's words, not written)
#pragma unroll
                    for (int u = 0; u < SB; ++u) asm volatile("" : "+v"(G[u].x), "+v"(G[u].y));
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int j = j0 + u;
                        const bool ok = (valid >> j & 1) != 0;
                        const uint32_t v = br[j] & 0xFFFFu, lf = G[u].x >> 16;
                        const uint32_t q = __umulhi(v, 0xAAAAAAABu) >> 3;
                        uint32_t s = v - L2_BLOCK_ITEMS * q, B_p = G[u].y + q;
                        asm volatile("" : "+v"(s), "+v"(B_p));                 // (both candidates computed: the choice is a select, not a branch per item)
                        const bool placed = v < (G[u].x & 0xFFFFu), later = v >= lf;
                        const uint32_t B = later ? br[j] >> 16 : B_p;
                        const uint64_t rem = key.r1(j) & g.pl.mr;
                        if (ok && (placed || later)) { L.img[B * 16 + s] = (uint32_t)rem; img8[B * 64 + 48 + s] = (uint8_t)(rem >> 32); }
                        lostm |= (ok && !placed && !later) ? 1u << j : 0u;
                    }
                }
                if (__any(lostm != 0)) {                                        // (rare: a full run, a tile with more whole blocks than the pool holds)
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (lostm >> j & 1) {
                            const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                            if (at < ovf_cap) ovf_buf[at] = place_key_d(b1, br[j] >> 16, key.r1(j) & g.pl.mr, g.pl);
                        }
                }
            }
            stamp(6);
            if (STAMP) st[9] += 1;
        }
        // ---- the bucket's end: the last tile's blocks; then what waits leaves as padded blocks ----
        lds_barrier();
#pragma unroll
        for (int it = 0; it < OUT_STEPS; ++it) { OutStep o; out_read(it, o); out_store(it, o); }
        lds_barrier();
        asm volatile("" : "+v"(tid_o));
        uint32_t blocks = 0;
        {
            const uint32_t w15 = L.img[tid_o * 16 + 15], cur = w15 & 0x0FFFFFFFu, cn = tid < P ? w15 >> 28 : 0u;
            const bool room = cur < capb;
            if (cn && room) for (uint32_t q = cn; q < L2_BLOCK_ITEMS; ++q) { L.img[tid_o * 16 + q] = 0xFFFFFFFFu; img8[tid_o * 64 + 48 + q] = 0xFF; }
            if (cn && !room)                                                    // (a full run: its waiting items take the overflow list)
                for (uint32_t q = 0; q < cn; ++q) {
                    const unsigned long long at = atomicAdd(ovf_n, 1ULL);
                    if (at < ovf_cap) ovf_buf[at] = place_key_d(b1, tid, ((uint64_t)img8[tid_o * 64 + 48 + q] << 32) | L.img[tid_o * 16 + q], g.pl);
                }
            L.gb[tid_o].x = cn && room ? 1u : 0u;
            blocks = cur + (cn && room ? 1u : 0u);
        }
        lds_barrier();
        for (uint32_t qi = tid_o; qi < 4 * P; qi += PART_BLOCK) {
            const uint32_t b = qi >> 2, w = qi & 3;
            const uint32_t cur = L.img[b * 16 + 15] & 0x0FFFFFFFu;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&L.img[b * 16 + 4 * w]);
            if (L.gb[b].x) *reinterpret_cast<u32x4*>(runs + (((uint64_t)b * capb + cur) << 6) + 16 * w) = v;
        }
        asm volatile("" : "+v"(tid_o));
        if (tid < P) cnt2[(uint64_t)b1 * P + tid_o] = blocks * L2_BLOCK_ITEMS;
    }
    if (STAMP && tid == 0 && stamps) for (int i = 0; i < 10; ++i) atomicAdd(&stamps[i], st[i]);
}

}  // namespace kg
