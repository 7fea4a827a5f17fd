// kg_l1_blocks.hpp -- level 1, block edition (round 6): whole 64-byte lines from a quad of lanes, items that never leave their registers,
// stores that have a whole tile to drain.
//
// What bounds level 1's group edition (tools/ubench_l1_tlb.hip, profiles/r06_ubench_l1_store_shapes.txt): its STORE PATTERN alone -- every
// workgroup appending ~16 six-byte items per tile to its segment of each of 512 buckets as 24-byte groups, two store instructions per group,
// every piece ending in partial lines -- costs 38.8-44.3 ms per round, the whole kernel's time.  The same items as 64-byte blocks stored by a
// QUAD of lanes in one instruction -- sixteen bytes each, a whole line per request -- 16.2 ms.  (Address translation is not it: laid out
// [workgroup][bucket] the pattern misses no UTCL1 entry -- 1.7 K misses against 3.1 G -- and runs no faster.)
//
// So: 6-byte items (HB1 = 2: n1 = 41 .. 47 bits, every k = 27 table of 64 .. 512 level-1 digits) travel in BLOCKS OF TEN = five pair records
// {low word a, low word b, high half a | high half b << 16} + 4 unused bytes, 64-byte aligned: level 2 loads a pair with one 12-byte
// instruction and needs no permute.
//
// Two things the earlier block editions (round 5's; this round's first, two 512-thread workgroups per CU) taught:
//  * The vector-memory counter is IN ORDER.  A tile's blocks were stored at the tile's end, behind the request for the next tile's bytes; the
//    wait for those bytes at the next tile's start was therefore a wait for every store of the tile to be ACKNOWLEDGED -- once per tile, with
//    nothing else of the workgroup to run meanwhile ("the store burst that nothing overlaps" of round 5).  Now a tile's finished blocks stay in
//    LDS and leave at the START of the next tile's work: by the time anything waits on the counter again they are a tile old.
//    (Measured on the 512-thread kernel, same call: 156.3 -> 146.7 ms per step.)
//  * Holding a tile's finished blocks AND the buckets' waiting items takes 52 + 32 KB per 8 K-base tile: more than half a CU's LDS, so two
//    workgroups per CU had to let a bucket's first block leave from the waiting image itself and park what the tile left over in registers
//    across the tile (40 of them, a second placing pass, three-way branches per item: placing cost more instructions than finding the k-mers).
//    ONE 1024-thread workgroup per CU (a 16 K-base tile) pays the 32 KB of waiting images once: every finished block has a pool image of its
//    own, the bucket's waiting items are copied in front of its first one (one lane, 64 bytes), and what is left over goes straight into the --
//    now free -- waiting image.  One placing pass, branch-free: rank v of bucket b lies in slot v - 10 q of pool image base[b] + q, or, from
//    rank 10 nblk on, in slot v - 10 nblk of image b.  Five barriers per 16 K bases where the 512-thread kernel had eight per 8 K.
//
// Everything that cannot be placed -- a segment that is full, more whole blocks in a tile than the pool holds (a skewed input) -- goes to the
// overflow list as a k-mer; if that list overflows the host redoes the round with the exact edition, as for the group edition.
#pragma once

namespace kg {

constexpr uint32_t L1B_ITEMS = 10, L1B_BYTES = 64;
constexpr int L1B_THREADS = 1024;
constexpr int L1B_TILE_BYTES = L1B_THREADS * PART_ITEMS;              // 16384
constexpr int L1B_TILE_STARTS = L1B_TILE_BYTES - CHUNK_OVERLAP;       // 16352
constexpr int L1B_LANES_WITH_STARTS = L1B_TILE_STARTS / PART_ITEMS;   // 1022
constexpr uint32_t L1B_PB = 512;                        // buckets, at most: thread b is bucket b
constexpr uint32_t L1B_POOL = 1712;                     // finished-block images: a tile of 16352 k-mers makes 1635 in the steady state

struct P1BLds {
    uint32_t hist[L1B_PB + 64];         // next rank in each bucket's line-up (the waiting items first) + one dump counter per lane of a wave
    uint2 gb[L1B_PB + 64];              // per bucket and tile: x = ranks below this are placed in the pool | ranks from this on are left over << 16 (what lies between: no room, the overflow list); y = image of the bucket's first block
    uint32_t where[L1B_POOL];           // pool image s leaves for block where[s] of the level-1 buffer, counted in 64-byte blocks from this workgroup's segment of bucket 0
    uint32_t wave_tot[L1B_THREADS / 64];
    uint32_t code[L1B_THREADS + 2];
    uint16_t bad[L1B_THREADS + 2];
    __attribute__((aligned(16))) uint32_t img[(L1B_PB + L1B_POOL) * 16];      // image B: dwords 16 B .. 16 B + 14.  B < 512: bucket B's waiting items (dword 15: blocks its segment holds | items that wait << 28); 512 + s: pool image s
};
static_assert(sizeof(P1BLds) <= 160 * 1024 - 1280, "one workgroup per CU");

// dword (low word) and 16-bit index (high half) of slot s of the image that starts at dword B
__device__ __forceinline__ uint32_t l1b_lo_at(uint32_t B, uint32_t s) { return B + 3 * (s >> 1) + (s & 1); }
__device__ __forceinline__ uint32_t l1b_hi_at(uint32_t B, uint32_t s) { return 2 * (B + 3 * (s >> 1) + 2) + (s & 1); }
// ... of item i of a bucket in the level-1 buffer (the exact edition and the slow paths): byte offsets
__device__ __host__ __forceinline__ uint64_t l1b_item_lo(uint64_t i) { const uint64_t blk = i / L1B_ITEMS; const uint32_t s = (uint32_t)(i - blk * L1B_ITEMS); return blk * L1B_BYTES + 12 * (s >> 1) + 4 * (s & 1); }
__device__ __host__ __forceinline__ uint64_t l1b_item_hi(uint64_t i) { const uint64_t blk = i / L1B_ITEMS; const uint32_t s = (uint32_t)(i - blk * L1B_ITEMS); return blk * L1B_BYTES + 12 * (s >> 1) + 8 + 2 * (s & 1); }

template <bool STAMP = false /* diagnostic (KATGPU_L1B_STAMP): wave 0's cycles per phase, summed over the workgroups into stamps[0 .. 9] */>
__global__ void __launch_bounds__(L1B_THREADS)          // four waves per SIMD: one workgroup per CU, 128 registers
k_p1b_scatter(DevTable t, PartGeom g, const uint8_t* __restrict__ bases, uint64_t n, uint64_t n_tiles, uint64_t tiles_per_wg,
              uint8_t* __restrict__ l1_buf, uint32_t segb /* blocks per segment */, uint32_t bucket_stride /* bytes from one bucket's first block to the next's; < 2^32 */,
              uint64_t* __restrict__ ovf_buf, unsigned long long* __restrict__ ovf_n, uint64_t ovf_cap, unsigned long long* __restrict__ stamps) {
    constexpr uint32_t PB = L1B_PB;
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;   // STAMP: [0] codes, [1] blocks out, [3] sweep, [5] per bucket, [7] placing, [2] [4] [6] [8] the barriers behind them, [9] tiles
    auto stamp = [&](int i) { if (STAMP) { const unsigned long long x = (unsigned long long)clock64(); st[i] += x - tq; tq = x; } };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // (158 KB: beyond what a static carve may take)
    P1BLds& L = *reinterpret_cast<P1BLds*>(lds_raw);
    const uint32_t tid = threadIdx.x, P = g.P1, k = t.k;            // P <= PB (host-checked), 17 <= k <= 31 (lean_applies)
    uint16_t* const img16 = reinterpret_cast<uint16_t*>(L.img);
    uint8_t* const seg0 = l1_buf + (uint64_t)blockIdx.x * segb * L1B_BYTES;     // this workgroup's segment of bucket 0 (uniform)
    const LeanGeom lg = lean_geom(k, t.canonical != 0, g.pl.n1);
    auto overflow = [&](uint32_t b, uint32_t lo, uint32_t hi) {
        const unsigned long long at = atomicAdd(ovf_n, 1ULL);
        if (at < ovf_cap) ovf_buf[at] = place_key_r1(b, ((uint64_t)hi << 32) | lo, g.pl);
    };
    if (tid < PB + 64) { L.hist[tid] = 0; L.gb[tid] = uint2{0u, 0u}; }
    if (tid < PB) L.img[tid * 16 + 15] = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, n_tiles);
    uint32_t n_out = 0;                                              // pool images the tile before filled
    uint32_t tid_o = tid;
    // ---- blocks out (of the tile before): a quad of lanes per block, sixteen bytes each, one store instruction per 64-byte line.  In SEVEN steps
    // (1712 pool images, 256 quads), each in two halves -- LDS reads, then the store -- that the sweep below takes between its k-mers: a CU's
    // stores leave at ~10 bytes a clock whoever waits for them (cycle stamps: the seven steps back to back took 8.9 K of a tile's 31 K cycles,
    // every wave of the workgroup parked behind them), so they leave while the sweep computes. ----
    constexpr int OUT_STEPS = (L1B_POOL + L1B_THREADS / 4 - 1) / (L1B_THREADS / 4);
    static_assert(OUT_STEPS <= PART_ITEMS / 2, "a step per pair of the sweep's k-mers");
    const uint32_t stride_b = bucket_stride >> 6;                      // (blocks: the host keeps a bucket's stride a multiple of 64 bytes)
    struct OutStep { u32x4 v; uint32_t where; };
    auto out_read = [&](int it, OutStep& o) {
        asm volatile("" : "+v"(tid_o));                              // (per use: what the compiler would derive from the lane's number once for the whole kernel is a register for the whole kernel)
        const uint32_t s = min((tid_o >> 2) + (uint32_t)it * (L1B_THREADS / 4), L1B_POOL - 1);
        o.where = L.where[s];
        o.v = *reinterpret_cast<const u32x4*>(&L.img[(PB + s) * 16 + (tid_o & 3) * 4]);
    };
    auto out_store = [&](int it, const OutStep& o) {
        asm volatile("" : "+v"(tid_o));
        if ((tid_o >> 2) + (uint32_t)it * (L1B_THREADS / 4) < n_out)
            *reinterpret_cast<u32x4*>(seg0 + (((uint64_t)o.where << 6) | ((tid_o & 3) * 16))) = o.v;
    };
    auto issue = [&](uint64_t tile_i) -> u32x4 {                     // (p1_tile_issue for this kernel's tile; from the opaque lane number: nothing of the address is kept)
        const uint64_t off = tile_i * L1B_TILE_STARTS + (uint64_t)tid_o * PART_ITEMS;
        return *reinterpret_cast<const u32x4*>(bases + (off + PART_ITEMS <= n ? off : 0));
    };
    u32x4 raw = issue(t0 < t1 ? t0 : 0);
    for (uint64_t tile = t0; tile < t1; ++tile) {
        if (STAMP) tq = (unsigned long long)clock64();
        asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));      // the tile's bytes have arrived -- waited for HERE, with nothing younger than a tile in flight, not behind the stores below
        {
            uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w}, code, bad;
            if (tile * L1B_TILE_STARTS + L1B_TILE_BYTES > n) p1_tile_fix(bases, n, tile * L1B_TILE_STARTS, raw, w4);      // (uniform: the stream's last tile)
            encode16(w4, code, bad);
            L.code[tid] = code;                          // (the tile before is through with its codes: its sweep was their only reader)
            L.bad[tid] = (uint16_t)bad;
            if (tid < 2) { L.code[L1B_THREADS + tid] = 0; L.bad[L1B_THREADS + tid] = 0xFFFF; }
        }
        stamp(0);
        asm volatile("" : "+v"(tid_o));
        raw = issue(tile + 1 < t1 ? tile + 1 : tile);    // the next tile: in flight behind all of this one
        stamp(1);
        lds_barrier();                                   // the codes are there; the pool has been read
        stamp(2);
        // ---- sweep 1: every window's k-mer, its bucket, its rank behind the bucket's waiting items (straight-line: a window without a k-mer
        // ranks in a dump counter; three LDS round trips in flight per lane) -- and the item stays where it is: two registers ----
        uint32_t lo[PART_ITEMS], hs[PART_ITEMS], rk2[PART_ITEMS / 2];  // low word; high half | bucket << 16; ranks, two to a register (a rank is below 2^15)
        const uint32_t v16 = tid < L1B_LANES_WITH_STARTS ? lean_valid16(L.bad[tid], L.bad[tid + 1], L.bad[tid + 2], k) : 0u;
        {
            LeanWin w{L.code[tid], L.code[tid + 1], L.code[tid + 2], 0, 0};
            uint32_t f_hi, f_lo;
            lean_fwd(w, lg, f_hi, f_lo);
            { const uint64_t rc0 = kmer_revcomp(((uint64_t)f_hi << 32) | f_lo, k); w.rc_hi = (uint32_t)(rc0 >> 32); w.rc_lo = (uint32_t)rc0; }
            const uint32_t dump = PB + (tid & 63);
            OutStep o;
#pragma unroll
            for (int j = 0; j < PART_ITEMS; ++j) {
                if (j % 2 == 0 && j / 2 < OUT_STEPS) out_read(j / 2, o);          // the tile before leaves: a step's reads before one k-mer, its store behind it
                if (j) { lean_step(w); lean_fwd(w, lg, f_hi, f_lo); lean_rc_roll(w, lg, f_lo); }
                uint32_t key_hi, key_lo;
                bool took_rc;
                const uint32_t b = lean_digit1(w, lg, g.pl, f_hi, f_lo, key_hi, key_lo, took_rc);
                const bool ok = (v16 & (0x8000u >> j)) != 0;
                const uint32_t sel = ok ? b : dump;
                const uint32_t r = atomicAdd(&L.hist[sel], 1u);
                if (j & 1) rk2[j >> 1] |= r << 16; else rk2[j >> 1] = r;
                lo[j] = key_lo;
                hs[j] = (key_hi & lg.mask_lhi) | (sel << 16);
                if (j % 2 == 1 && j / 2 < OUT_STEPS) out_store(j / 2, o);
            }
        }
        stamp(3);
        lds_barrier();
        stamp(4);
        // ---- per bucket: how many blocks complete, what the segment and the pool have room for; the waiting items move in front of the first ----
        asm volatile("" : "+v"(tid_o));
        {
            uint32_t want = 0, avail = 0, nblk = 0, cur = 0, cn_old = 0;
            if (tid < P) {
                avail = L.hist[tid_o];
                const uint32_t w15 = L.img[tid_o * 16 + 15];
                cur = w15 & 0x0FFFFFFFu; cn_old = w15 >> 28;
                nblk = __umulhi(avail, 0xCCCCCCCDu) >> 3;             // avail / 10
                want = min(nblk, segb > cur ? segb - cur : 0u);
            }
            const uint32_t lane = tid_o & 63, wave = tid_o >> 6;
            const uint32_t inc = wave_inclusive_scan(want);
            if (lane == 63) L.wave_tot[wave] = inc;
            lds_barrier();
            const uint32_t sc = wave_inclusive_scan(lane < L1B_THREADS / 64 ? L.wave_tot[lane] : 0u);
            const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
            const uint32_t excl = (wv ? lane_value(sc, (int)wv - 1) : 0u) + inc - want;
            { const uint32_t tot = lane_value(sc, L1B_THREADS / 64 - 1); n_out = tot < L1B_POOL ? tot : L1B_POOL; }
            if (tid < P) {
                const uint32_t nst = excl < L1B_POOL ? min(want, L1B_POOL - excl) : 0u;      // blocks that leave
                // Placed: ranks below 10 nst.  Left over (into the waiting image, which is free once its items have moved): ranks from 10 nblk on.
                // What lies between has no room and takes the overflow list -- and when NO block can leave although one is complete (a full segment, an
                // exhausted pool), all of the tile's k-mers do, and what waited still waits.
                const bool stuck = nblk != 0 && nst == 0;
                L.gb[tid_o] = uint2{(L1B_ITEMS * nst) | ((stuck ? 0xFFFFu : L1B_ITEMS * nblk) << 16), PB + excl};
                if (nst) {
                    const u32x4* src = reinterpret_cast<const u32x4*>(&L.img[tid_o * 16]);
                    u32x4* dst = reinterpret_cast<u32x4*>(&L.img[(PB + excl) * 16]);
                    const u32x4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
                    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
                    const uint32_t first = tid_o * stride_b + cur;
                    for (uint32_t q = 0; q < nst; ++q) L.where[excl + q] = first + q;
                }
                const uint32_t cn = stuck ? cn_old : avail - L1B_ITEMS * nblk;
                L.img[tid_o * 16 + 15] = (cur + nst) | (cn << 28);
                L.hist[tid_o] = cn;                       // the next tile's k-mers rank behind what waits
            } else if (tid < PB + 64) L.hist[tid_o] = 0;  // (buckets the table does not have, the dump counters: a rank is kept in sixteen bits)
        }
        stamp(5);
        lds_barrier();
        stamp(6);
        // ---- placing: branch-free but for the two writes.  Rank v = 10 q + s below gb.x's low half: slot s of pool image gb.y + q; from its
        // high half on: slot v - that of the bucket's waiting image. ----
        {
            uint32_t lostm = 0;
            constexpr int SB = 4;                                                    // four buckets' words in flight
#pragma unroll
            for (int j0 = 0; j0 < PART_ITEMS; j0 += SB) {
                uint2 G[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) G[u] = L.gb[hs[j0 + u] >> 16];            // (a window without a k-mer: a dump counter's entry, {0, 0}: "left over", not written)
#pragma unroll
                for (int u = 0; u < SB; ++u) asm volatile("" : "+v"(G[u].x), "+v"(G[u].y));
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int j = j0 + u;
                    const bool ok = (v16 & (0x8000u >> j)) != 0;
                    const uint32_t v = (rk2[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, lf = G[u].x >> 16;
                    const uint32_t q = __umulhi(v, 0xCCCCCCCDu) >> 3;
                    uint32_t s = v - L1B_ITEMS * q, B_p = G[u].y + q;              // (what is left over has q = nblk: its slot in the waiting image is v - 10 q as well)
                    asm volatile("" : "+v"(s), "+v"(B_p));                         // (both candidates computed: the choice is a select, not a branch per item)
                    const bool placed = v < (G[u].x & 0xFFFFu), later = v >= lf;
                    const uint32_t B = later ? hs[j] >> 16 : B_p, odd = s & 1u;
                    // slot s of image B, in bytes: low word at 64 B + 12 (s >> 1) + 4 (s & 1) = 64 B + 6 s - 2 odd, high half 8 - 2 odd behind it
                    const uint32_t lo_at = (B << 6) + 6u * s - 2u * odd;
                    if (ok && (placed || later)) {
                        *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(L.img) + lo_at) = lo[j];
                        *reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(L.img) + lo_at + 8u - 2u * odd) = (uint16_t)hs[j];
                    }
                    lostm |= (ok && !placed && !later) ? 1u << j : 0u;
                }
            }
            if (__any(lostm != 0)) {                                     // (rare: a full segment, a tile with more whole blocks than the pool holds)
#pragma unroll
                for (int j = 0; j < PART_ITEMS; ++j) if (lostm >> j & 1) overflow(hs[j] >> 16, lo[j], hs[j] & 0xFFFFu);      // (unrolled: a loop would index the register arrays)
            }
        }
        stamp(7);
        lds_barrier();                                   // images and pool settled: the pool leaves with the next tile's start
        stamp(8);
        if (STAMP) st[9] += 1;
    }
    if (STAMP && tid == 0 && stamps) for (int i = 0; i < 10; ++i) atomicAdd(&stamps[i], st[i]);
    // ---- the end of the stream: the last tile's blocks; then the waiting items leave as padded blocks, the segments' rest is "no item" ----
#pragma unroll
    for (int it = 0; it < OUT_STEPS; ++it) { OutStep o; out_read(it, o); out_store(it, o); }
    lds_barrier();
    if (tid < P) {
        const uint32_t w15 = L.img[tid * 16 + 15], cur = w15 & 0x0FFFFFFFu, cn = w15 >> 28;
        L.img[tid * 16 + 15] = cur;
        const bool room = cur < segb;
        if (cn && !room) {                                                   // (a full segment: its waiting items take the overflow list)
            for (uint32_t s = 0; s < cn; ++s) overflow(tid, L.img[l1b_lo_at(tid * 16, s)], img16[l1b_hi_at(tid * 16, s)]);
        }
        if (cn && room) for (uint32_t s = cn; s < L1B_ITEMS; ++s) { L.img[l1b_lo_at(tid * 16, s)] = 0xFFFFFFFFu; img16[l1b_hi_at(tid * 16, s)] = 0xFFFFu; }
        L.gb[tid].x = cn && room ? 1u : 0u;
    }
    lds_barrier();
    for (uint32_t qi = tid; qi < 4 * P; qi += L1B_THREADS) {
        const uint32_t b = qi >> 2, w = qi & 3;
        const uint32_t cur = L.img[b * 16 + 15];
        u32x4 v = *reinterpret_cast<const u32x4*>(&L.img[b * 16 + 4 * w]);
        if (w == 3) v.w = 0xFFFFFFFFu;
        if (L.gb[b].x) *reinterpret_cast<u32x4*>(seg0 + ((uint64_t)b * bucket_stride + ((uint64_t)cur << 6) + 16 * w)) = v;
    }
    lds_barrier();
    {
        const uint32_t wave = tid >> 6, lane = tid & 63;
        for (uint32_t b = wave; b < P; b += L1B_THREADS / 64) {
            const uint32_t from = L.img[b * 16 + 15] + L.gb[b].x;
            uint8_t* seg = seg0 + (uint64_t)b * bucket_stride;
            for (uint32_t q = from * 4 + lane; q < segb * 4; q += 64) *reinterpret_cast<u32x4*>(seg + (uint64_t)q * 16) = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        }
    }
}

}  // namespace kg
