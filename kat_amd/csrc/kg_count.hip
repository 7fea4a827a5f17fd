// kg_count.hip -- counting: the direct kernel's launches and overflow guard, the partitioned counter's host loop (rounds, passes,
// fall backs, growth beside the arena), the host feeder (pinned staging -> device rings -> count_resident) and the katgpu_count*
// entry points that replace InputHandler::count (lib/src/input_handler.cc:180-202).
#include "kg_host.hpp"
#include "kg_ingest.hpp"
#include "kg_kernels.hpp"
#include "kg_partition.hpp"
#include "kg_partition_wide.hpp"
#include "kg_wide.hpp"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

// ------------------------------------------------------------------ counting --------------------------

// test hooks (tests/test_gpu_parity.py): shrink the sweep threshold / the launch size so small inputs exercise them
static const uint64_t g_test_sweep_thr = hook("KATGPU_TEST_SWEEP_THR") ? strtoull(hook("KATGPU_TEST_SWEEP_THR"), nullptr, 10) : 0;
static const uint64_t g_test_max_starts = hook("KATGPU_TEST_MAX_STARTS") ? strtoull(hook("KATGPU_TEST_MAX_STARTS"), nullptr, 10) : 0;

// k_count adds with no-return atomics and cannot see a 32-bit wrap; make one impossible.  Invariant: every counter
// <= count_bound + unchecked_adds.  When the next launch could break "<= 2^32-1", k_sweep moves multiples of thr out of
// the large counters into the side table and reports the new maximum.
static int maybe_sweep(katgpu_table* t, uint64_t next_starts) {
    katgpu_ctx* c = t->ctx;
    if (t->dev().cbits) return KATGPU_OK;                          // packed tables take the checked add (kg_device.hpp: table_inc)
    const uint64_t limit = g_test_sweep_thr ? 2 * g_test_sweep_thr - 1 : 0xFFFFFFFFULL;
    if (t->count_bound + t->unchecked_adds + next_starts <= limit) return KATGPU_OK;
    const uint32_t thr = g_test_sweep_thr ? (uint32_t)g_test_sweep_thr : 0x80000000u;
    unsigned long long* scratch = (unsigned long long*)&t->dev().ctrs[CTR_SCRATCH];
    HIPCHK(c, hipMemsetAsync(scratch, 0, sizeof(uint64_t), c->stream));
    {
        ScopedTimer tm(c, KATGPU_K_REGROW, t->dev().cap);
        hipLaunchKernelGGL(k_sweep, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, t->dev(), thr, scratch);
    }
    uint64_t mx = 0;
    HIPCHK(c, hipMemcpyAsync(&mx, scratch, sizeof mx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t->count_bound = mx;
    t->unchecked_adds = 0;
    return KATGPU_OK;
}

static int launch_count(katgpu_table* t, const uint8_t* dev_bases, size_t n) {
    katgpu_ctx* c = t->ctx;
    if (n < t->dev().k) return KATGPU_OK;
    if (t->dev().keys_b) {                                         // wide k-mers: checked adds, nothing to sweep
        const uint64_t n_chunks = (n + WIDE_CHUNK_STARTS - 1) / WIDE_CHUNK_STARTS;
        const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * 4);
        ScopedTimer tm(c, KATGPU_K_COUNT, n);
        if ((reinterpret_cast<uintptr_t>(dev_bases) & 15) == 0)
            hipLaunchKernelGGL(k_count_w<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), dev_bases, (uint64_t)n, n_chunks);
        else
            hipLaunchKernelGGL(k_count_w<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), dev_bases, (uint64_t)n, n_chunks);
        HIPCHK(c, hipGetLastError());
        return KATGPU_OK;
    }
    int src = maybe_sweep(t, n);
    if (src) return src;
    t->unchecked_adds += n;
    const uint64_t n_chunks = (n + CHUNK_STARTS - 1) / CHUNK_STARTS;
    // exactly the resident set: a larger grid leaves a second, thinly populated wave of blocks (measured 12.7 G k-mers/s at
    // 8 blocks/CU requested vs 15.5 at the 6 that were actually resident)
    const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * c->count_blocks_per_cu);
    ScopedTimer tm(c, KATGPU_K_COUNT, n);
    if ((reinterpret_cast<uintptr_t>(dev_bases) & 15) == 0)
        hipLaunchKernelGGL(k_count<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), dev_bases, (uint64_t)n, n_chunks);
    else
        hipLaunchKernelGGL(k_count<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), dev_bases, (uint64_t)n, n_chunks);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}


// ------------------------------------------------------------------ partitioned counter (kg_partition.hpp) ----

static const uint64_t g_part_min_starts = getenv("KATGPU_PART_MIN_STARTS") ? strtoull(getenv("KATGPU_PART_MIN_STARTS"), nullptr, 10) : (32ULL << 20);
static const uint64_t g_test_round_items = hook("KATGPU_TEST_ROUND_ITEMS") ? strtoull(hook("KATGPU_TEST_ROUND_ITEMS"), nullptr, 10) : 0;
// share of the free HBM the partition arena may take (multi-GPU runs may lower it; bench.py sets 0.75 there)
static const double g_arena_fraction = getenv("KATGPU_ARENA_FRACTION") ? std::min(0.95, std::max(0.05, atof(getenv("KATGPU_ARENA_FRACTION")))) : 0.85;
static const uint32_t g_p1_wgs = hook("KATGPU_P1_WGS") ? std::max<uint32_t>(1, (uint32_t)strtoul(hook("KATGPU_P1_WGS"), nullptr, 10)) : 3;   // level-1 workgroups per CU
static const bool g_apply_noinline = hook("KATGPU_APPLY_NOINLINE") != nullptr;   // A/B: no inline claims in a table's first round
static const uint32_t g_apply_block = hook("KATGPU_APPLY_BLOCK") ? (uint32_t)strtoul(hook("KATGPU_APPLY_BLOCK"), nullptr, 10) : 0;   // 0: by region size
// level 2 without its histogram pass (kg_partition.hpp: k_p2_fast): 0 = never, 1 = when the mean run is long enough for the
// capacity slack to cover the noise, 2 = always (tests).  KATGPU_TEST_P2_OVF_CAP shrinks the overflow list (tests: forces the
// fall back to the exact kernel).
// level 1 without its counting pass (kg_partition.hpp: k_p1v2_scatter<true>, one fixed-capacity segment per workgroup and bucket):
// 0 = never, 1 = for rounds of at least 64 M k-mers (the default), 2 = always (tests)
static const uint32_t g_test_l1_cpb = hook("KATGPU_TEST_L1_CPB") ? (uint32_t)strtoul(hook("KATGPU_TEST_L1_CPB"), nullptr, 10) : 0;   // tests: segment capacity (forces overflow)
static const bool g_l1_lean = hook_u64("KATGPU_L1_LEAN", 1) != 0;   // A/B: 0 = level 1's ranking sweep in its 64-bit form (kg_l1_lean.hpp is the 32-bit one)
static const uint32_t g_l1_fast = hook("KATGPU_L1_FAST") ? (uint32_t)strtoul(hook("KATGPU_L1_FAST"), nullptr, 10) : 1;
static const bool g_l1_blocks = hook_u64("KATGPU_L1_BLOCKS", 1) != 0;   // A/B: 0 = the group edition of the segmented level 1 where the block edition (kg_l1_blocks.hpp) would run
static const uint32_t g_p2_fast = hook("KATGPU_P2_FAST") ? (uint32_t)strtoul(hook("KATGPU_P2_FAST"), nullptr, 10) : 1;
static const uint64_t g_test_p2_ovf_cap = hook("KATGPU_TEST_P2_OVF_CAP") ? strtoull(hook("KATGPU_TEST_P2_OVF_CAP"), nullptr, 10) : 0;
static const uint32_t g_test_spill_mod = hook("KATGPU_TEST_SPILL_MOD") ? (uint32_t)strtoul(hook("KATGPU_TEST_SPILL_MOD"), nullptr, 10) : 0;
static const uint64_t g_test_ap_seg = hook_u64("KATGPU_TEST_AP_SEG", 0) & ~3ULL;   // tests: k-mers per walk segment of the apply kernels (several segments per run)
static const uint32_t g_apply_nr = (uint32_t)hook_u64("KATGPU_APPLY_NR", 2);            // A/B: probe rounds of the packed apply at the bench's shape (1, 2 or 3)
static const uint32_t g_apply_min_q = (uint32_t)hook_u64("KATGPU_APPLY_MIN_Q", 72);     // A/B: queue entries per wave the SECOND workgroup of a CU must leave (>= 72)
static const uint32_t g_apply_per_cu = (uint32_t)hook_u64("KATGPU_APPLY_PER_CU", 0);   // A/B: packed apply workgroups per CU (0: as many as the LDS holds)
static const bool g_l1b_stamp = hook_u64("KATGPU_L1B_STAMP", 0) != 0;   // diagnostic: level 1's block edition with cycle stamps (printed per round)
static const bool g_p2x = hook_u64("KATGPU_P2X", 1) != 0;   // A/B: 0 = k_p2_fast's block edition where kg_l2_blocks.hpp's kernel would run
static const bool g_p2x_stamp = hook_u64("KATGPU_P2X_STAMP", 0) != 0;   // diagnostic: that kernel with cycle stamps (printed per pass)
static const bool g_p2_stamp = hook_u64("KATGPU_P2_STAMP", 0) != 0;   // diagnostic: the bench-shape one-pass level 2 with cycle stamps (printed per pass)
static const bool g_apply_stamp = hook_u64("KATGPU_APPLY_STAMP", 0) != 0;              // diagnostic: the bench-shape apply with cycle stamps (printed per pass)

static const uint32_t g_test_hb = hook("KATGPU_TEST_HB") ? (uint32_t)strtoul(hook("KATGPU_TEST_HB"), nullptr, 10) : 0;   // A/B: wider level-2 items than needed (1, 2, 4)
static bool part_geometry(const DevTable& d, PartGeom* g) {
    g->R = d.n_regions; g->S = d.region_slots; g->P1 = d.p1; g->P2 = d.p2; g->l2 = d.l2;
    g->b_lo = 0; g->b_hi = d.p1;
    g->pl = place_make(d.k, d.p1, d.n1, d.l2);
    g->hb = std::max(l2_hi_bytes(g->pl.rb), g_test_hb);
    g->hb1 = l2_hi_bytes(g->pl.n1);
    g->cbits = d.cbits;
    if (d.cbits && g->hb > 2) g->hb = 2;                       // (a packed table's remainder has at most 44 bits)
    // the apply kernels hold a region of whole 16-byte lines, at least a wave's worth of slots, in LDS
    return d.k <= 32 && g->pl.rb <= 63 /* all-ones is "no item" */ && g->P1 <= MAX_PARTS && g->P2 <= MAX_PARTS && g->S % 4 == 0 && g->S >= 64 && g->S <= AP2_MAX_SLOTS;
}
// bytes of partition arena per k-mer of a round: level-1 buffer (8 B + the segment slack 1/24), level-2 buffer (4 + hb B, that
// slack again + the run slack 1/16 + the group padding's allowance 2 * 1024 / tile), overflow list (8 B / 32)
static double l2_items_per_l1_item(uint32_t hb) { return 1 + 1.0 / 16 + 2.0 * MAX_PARTS / l2_tile_items(hb); }
// buckets per pass of level 2 + apply: a CU-full when the buckets are a whole number of those (alloc_dev_table sees to it), else all
static const uint32_t g_test_pass_buckets = hook("KATGPU_TEST_PASS_BUCKETS") ? (uint32_t)strtoul(hook("KATGPU_TEST_PASS_BUCKETS"), nullptr, 10) : 0;   // tests: passes of this many buckets
static uint32_t pass_buckets(uint32_t p1, uint32_t n_cu) {
    if (g_test_pass_buckets) return std::max<uint32_t>(1, std::min(p1, g_test_pass_buckets));
    return n_cu && p1 > n_cu && p1 % n_cu == 0 ? n_cu : p1;
}
// ... of which the level-2 buffer holds one pass = 1 / passes of a round
static double arena_bytes_per_item(uint32_t hb, uint32_t passes) { return 8.0 * (1 + 1.0 / 24) + l2_bytes_per_item(hb) * (1 + 1.0 / 24) * l2_items_per_l1_item(hb) / passes + 0.25 + 0.02; }

static const bool g_test_grow_nomem = hook("KATGPU_TEST_GROW_NOMEM") != nullptr;   // tests: table growth "fails" while the arena is busy

// Growth while a partition call holds the arena.  First with the arena protected; when the device cannot hold the old
// table, the new one and the arena at once, `stash` (spilled keys that live in the arena, may be null) is parked in host
// memory, the arena is given up, the growth retried and the keys re-inserted from the host.  *arena_lost tells the
// caller that its carve of the arena is gone.
typedef std::vector<std::pair<const uint64_t*, uint64_t>> KeyLists;
static int grow_beside_arena(katgpu_table* t, uint64_t incoming, uint64_t min_cap, const KeyLists& stash, bool* arena_lost) {
    uint64_t n_stash = 0;
    for (auto& l : stash) n_stash += l.second;
    katgpu_ctx* c = t->ctx;
    auto grow = [&]() -> int {
        if (min_cap > t->dev().cap) {
            if (t->disable_grow) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
            uint64_t nc = t->dev().cap; while (nc < min_cap) nc *= 2;
            return regrow(t, nc);
        }
        return ensure_room(t, incoming);
    };
    *arena_lost = false;
    int rc = g_test_grow_nomem ? KATGPU_ERR_NOMEM : grow();
    if (rc != KATGPU_ERR_NOMEM) return rc;
    (void)hipGetLastError();
    std::vector<uint64_t> host;
    if (n_stash) {
        try { host.resize(n_stash); } catch (...) { return fail(c, KATGPU_ERR_NOMEM, "no host memory to park %llu spilled k-mers", (unsigned long long)n_stash); }
        uint64_t at = 0;
        for (auto& l : stash) { HIPCHK(c, hipMemcpy(host.data() + at, l.first, l.second * 8, hipMemcpyDeviceToHost)); at += l.second; }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    release_arena(c);
    *arena_lost = true;
    if (g_trace) fprintf(stderr, "[katgpu] growth beside the partition arena failed: arena released, %llu keys parked on the host\n", (unsigned long long)n_stash);
    rc = grow();
    if (rc) return rc;
    if (n_stash) {
        const size_t chunk = std::min<size_t>(n_stash, (size_t)32 << 20);
        uint64_t* d = nullptr;
        HIPCHK(c, pool_alloc(c, (void**)&d, chunk * 8));
        for (size_t i = 0; i < n_stash && rc == KATGPU_OK; i += chunk) {
            const size_t m = std::min(chunk, (size_t)n_stash - i);
            if (hipMemcpyAsync(d, host.data() + i, m * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(c, KATGPU_ERR_DEVICE, "spill upload"); break; }
            ScopedTimer tm(c, KATGPU_K_COUNT, m);
            hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(c, m, 256, 6)), dim3(256), 0, c->stream, t->dev(), (const uint64_t*)d, (uint64_t)m);
            if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "spill insert");
        }
        pool_release(c, d);
    }
    return rc;
}

#define KG_FOR_HB_APPLY(M) M(0) M(1) M(2) M(4)
// the dynamic-LDS ceiling of a kernel is raised once per device (c->lds_attr holds the kernels done)
static int ensure_lds_attr(katgpu_ctx* c, const void* fn, size_t bytes) {
    if (c->lds_attr.count(fn)) return KATGPU_OK;
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    c->lds_attr.insert(fn);
    return KATGPU_OK;
}
#define KG_LDS_ATTR(K, BYTES) do { int rc__ = ensure_lds_attr(c, reinterpret_cast<const void*>(K), (BYTES)); if (rc__) return rc__; } while (0)
constexpr size_t LDS_BYTES = 160 * 1024, LDS_GRANULE = 1280;     // gfx950: 160 KB per CU, allocated in 320-dword granules

// Level 3 of a pass: one workgroup per region of buckets [g.b_lo, g.b_hi) -- region into LDS, its run applied, region written back
// (kg_partition.hpp: k_p3_apply_pk for packed tables, k_p3_apply2 for KV12).  A table's first round claims its new k-mers inside
// the probe rounds (INLINE_CLAIM); the test suite's spill hook has its own instantiations.
static int launch_apply(katgpu_table* t, const PartGeom& g, const uint64_t* off2, const uint8_t* l2_buf, uint64_t* spill_buf, unsigned long long* spill_n,
                        const uint32_t* run_len, const uint64_t* bucket_end) {
    katgpu_ctx* c = t->ctx;
    const uint32_t n_cu = (uint32_t)c->n_cu, regions = (g.b_hi - g.b_lo) * g.P2;
    const bool fresh = t->distinct == 0 && !g_apply_noinline;
    const bool hooked = g_test_spill_mod != 0;
    if (g.cbits) {
        // 512-thread workgroups, as many per CU as the LDS holds next to their queues (two at the bench's 9344-slot regions, four for
        // small regions); one of 1024 threads when a region leaves no room for a second
        const size_t region_b = (size_t)g.S * 8;
        auto room = [&](uint32_t wgs) -> long { return (long)(LDS_BYTES / wgs / LDS_GRANULE * LDS_GRANULE) - 64 - (long)region_b; };   // bytes left for the queues
        uint32_t per_cu = 4;
        // room for the queues of eight waves: 96 entries each for a third and fourth workgroup, 72 for the second -- a second workgroup is worth
        // short queues (config 5's 9656-slot regions leave exactly 72: apply 103.5 -> 98.2 ms per step against one 1024-thread workgroup)
        auto min_q = [&](uint32_t wgs) -> long { return 8L * 8 * (wgs == 2 ? std::max<uint32_t>(g_apply_min_q, 72) : 96); };
        while (per_cu > 1 && room(per_cu) < min_q(per_cu)) --per_cu;
        if (g_apply_per_cu) per_cu = std::min(per_cu, g_apply_per_cu);
        uint32_t blk = per_cu == 1 ? 1024 : 512;
        if (g_apply_block == 512 || g_apply_block == 1024) blk = g_apply_block;
        if (hooked) { blk = 1024; per_cu = 1; }
        per_cu = std::min<uint32_t>(per_cu, 2048 / blk);
        const uint32_t nw = blk / 64;
        while (per_cu > 1 && room(per_cu) < (long)(nw * 72 * 8)) --per_cu;
        const uint32_t qcap = (uint32_t)std::min<long>(256, room(per_cu) / (long)(nw * 8));
        if (qcap < 72) return fail(c, KATGPU_ERR_DEVICE, "a region of %u packed slots leaves no room for the apply kernel's queues", g.S);
        const size_t lds = region_b + (size_t)nw * qcap * 8;
        const uint64_t seg_cap = (pk_half(g.cbits) - 1) & ~3ULL;                              // a walk adds less than half the count range
        const uint64_t seg_len = g_test_ap_seg ? std::min<uint64_t>(g_test_ap_seg, seg_cap) : std::min<uint64_t>(AP2_SEGMENT, seg_cap);
        const dim3 grid(std::min<uint32_t>(regions, n_cu * per_cu));
        // A table whose slots have not been cleared yet (katgpu_table::zero_from): this pass is their first sweep when its regions are the
        // next in line -- the kernel starts every region of the pass from zeros and writes every one back; else they are cleared now.
        const uint64_t r_lo = (uint64_t)g.b_lo * g.P2, r_hi = (uint64_t)g.b_hi * g.P2;
        uint32_t zero_fill = 0;
        if (t->zero_from != ~0ULL) { if (t->zero_from == r_lo) zero_fill = 1; else t->zero_rest(); }
#define KG_APK(B, KP, HB, INL, HK, PF, NR) do { KG_LDS_ATTR((k_p3_apply_pk<B, KP, HB, INL, HK, PF, NR>), LDS_BYTES - 256); \
            hipLaunchKernelGGL((k_p3_apply_pk<B, KP, HB, INL, HK, PF, NR>), grid, dim3(B), lds, c->stream, t->dv, g, off2, l2_buf, spill_buf, spill_n, run_len, bucket_end, \
                               qcap, seg_len, g_test_spill_mod, zero_fill); } while (0)
        // probe rounds before the queue: 2 (measured at the bench size, same box: 163 ms per step against 177 with 3 and 198 with inline
        // claims in every round); a table's first round claims inline and keeps 3
#define KG_APK_SHAPE(HB) case HB: \
            if (hooked) KG_APK(1024, 5, HB, false, true, true, 3); \
            else if (blk == 1024) { if (fresh) KG_APK(1024, 5, HB, true, false, true, 3); else KG_APK(1024, 5, HB, false, false, true, 2); } \
            else if (g.S <= 4096) { if (fresh) KG_APK(512, 4, HB, true, false, true, 3); else KG_APK(512, 4, HB, false, false, true, 2); } \
            else if (HB == 1 && g_apply_nr == 1 && !fresh) KG_APK(512, 10, 1, false, false, false, 1); \
            else if (HB == 1 && g_apply_nr == 3 && !fresh) KG_APK(512, 10, 1, false, false, false, 3); \
            else if (HB == 1 && g_apply_stamp && !fresh) { KG_LDS_ATTR((k_p3_apply_pk<512, 10, 1, false, false, false, 2, true>), LDS_BYTES - 256); \
                hipLaunchKernelGGL((k_p3_apply_pk<512, 10, 1, false, false, false, 2, true>), grid, dim3(512), lds, c->stream, t->dv, g, off2, l2_buf, spill_buf, spill_n, run_len, bucket_end, qcap, seg_len, g_test_spill_mod, zero_fill); } \
            else { if (fresh) KG_APK(512, 10, HB, true, false, false, 3); else KG_APK(512, 10, HB, false, false, false, 2); } \
            break;
        switch (g.hb) { KG_APK_SHAPE(0) KG_APK_SHAPE(1) KG_APK_SHAPE(2) default: return fail(c, KATGPU_ERR_DEVICE, "packed apply: item width %u", g.hb); }
#undef KG_APK_SHAPE
#undef KG_APK
        HIPCHK(c, hipGetLastError());
        if (zero_fill) t->zero_from = r_hi >= t->dv.n_regions ? ~0ULL : r_hi;      // (regions [r_lo, r_hi) have had their first sweep)
        return KATGPU_OK;
    }
    // KV12: as many workgroups per CU as the regions' LDS footprint (and the 2048-thread limit) admits
    const uint32_t blk = hooked ? 1024 : g_apply_block ? g_apply_block : (g.S <= 4096 ? 512 : 1024);
    const bool big = g.S > 8192 || hooked;
    const size_t lds = (size_t)g.S * 12 + (size_t)(blk / 64) * (big ? AP2_QCAP_BIG : AP2_QCAP) * 12;
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(LDS_BYTES / (lds + 512), 2048 / blk));
    const dim3 grid(std::min<uint32_t>(regions, n_cu * per_cu));
    const uint64_t seg_len = g_test_ap_seg ? std::min<uint64_t>(g_test_ap_seg, AP2_SEGMENT) : AP2_SEGMENT;
#define KG_AP2(B, KP, HB, INL, QC, HK) do { KG_LDS_ATTR((k_p3_apply2<B, KP, 4, 3, HB, false, INL, true, QC, HK>), LDS_BYTES - 256); \
        hipLaunchKernelGGL((k_p3_apply2<B, KP, 4, 3, HB, false, INL, true, QC, HK>), grid, dim3(B), lds, c->stream, t->dev(), g, off2, l2_buf, spill_buf, spill_n, run_len, bucket_end, \
                           (unsigned long long*)nullptr, g_test_spill_mod, seg_len); } while (0)
#define KG_AP2_SHAPE(HB) case HB: \
        if (hooked) KG_AP2(1024, 5, HB, false, AP2_QCAP_BIG, true); \
        else if (blk == 512) { if (g.S <= 2048) { if (fresh) KG_AP2(512, 2, HB, true, AP2_QCAP, false); else KG_AP2(512, 2, HB, false, AP2_QCAP, false); } \
                               else { if (fresh) KG_AP2(512, 4, HB, true, AP2_QCAP, false); else KG_AP2(512, 4, HB, false, AP2_QCAP, false); } } \
        else if (big) { if (fresh) KG_AP2(1024, 5, HB, true, AP2_QCAP_BIG, false); else KG_AP2(1024, 5, HB, false, AP2_QCAP_BIG, false); } \
        else { if (fresh) KG_AP2(1024, 4, HB, true, AP2_QCAP, false); else KG_AP2(1024, 4, HB, false, AP2_QCAP, false); } \
        break;
    switch (g.hb) { KG_FOR_HB_APPLY(KG_AP2_SHAPE) }
#undef KG_AP2_SHAPE
#undef KG_AP2
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}

// Count a resident, 16-byte aligned base stream through partition rounds.  *done = number of window starts consumed
// (all of them unless the geometry stops fitting, in which case the caller finishes with the direct kernel).
static int count_partitioned(katgpu_table* t, const uint8_t* dev_bases, size_t n, size_t* done) {
    katgpu_ctx* c = t->ctx;
    const uint32_t k = t->dv.k;
    const size_t n_starts = n - k + 1;
    *done = 0;
    c->arena_borrowed = false;                    // a borrowed arena is only promised until the next count call
    if (n < 64) return KATGPU_OK;                 // (the tile loader reads whole 16-byte pieces: direct path)
    // A table hopelessly small for this input (KAT's default -H against a whole run) would spill nearly every k-mer of the
    // first round: give it room for 1/16 of the starts first -- cheap while it is still small, and before the arena exists.
    if (!g_test_round_items && !t->disable_grow && t->dv.cap < n_starts / 16) {
        uint64_t nc = t->dv.cap; while (nc < n_starts / 16) nc *= 2;
        int grc = regrow(t, nc);
        if (grc) return grc;
    }
    const uint32_t W = (uint32_t)c->n_cu * std::min<uint32_t>(g_p1_wgs, 4);                     // level-1 workgroups (rows of hist1 / offs)
    const uint32_t W2 = (uint32_t)c->n_cu;                                                      // level-2 / apply: one per CU
    const size_t tile_starts = P1_TILE_STARTS;
#define KG_FOR_HB(M) M(0) M(1) M(2) M(4)
    if (!c->part_attr_set) {
#define KG_ATTR_HB(HB) KG_LDS_ATTR((k_p2<HB, false>), sizeof(P2Lds<HB>)); KG_LDS_ATTR((k_p2_fast<HB, false>), sizeof(P2FastLds<HB>::type)); \
                       KG_LDS_ATTR((k_p2<HB, true>), sizeof(P2Lds<HB>)); KG_LDS_ATTR((k_p2_fast<HB, true>), sizeof(P2FastLds<HB>::type));
        KG_FOR_HB(KG_ATTR_HB)
#undef KG_ATTR_HB
        c->part_attr_set = true;
    }
    // ---- arena: [hist1 | offs | l1_off | off2 | cnt2 | bend | spill_n, ovf_n | L1 buffer | L2 buffer | overflow list] ----
    // L1 buffer: a round's k-mers + 1/24 + 64 per workgroup and bucket (segment slack of k_p1v2_scatter<true>);
    // L2 buffer: items of 4 + hb bytes in groups of four (kg_partition.hpp "the level-2 buffer"): the L1 count + 1/16 + 16 per region
    // (capacity slack of k_p2_fast) + two items per tile and region (group padding); overflow list: 1/32.
    // 14.8 bytes per k-mer of a round at hb = 1 (k = 27 at the bench size), 18.5 at hb = 4 -- with one pass; the level-2 buffer
    // holds one PASS of level 2 + apply (a CU-full of buckets, see the rounds below): 11.7 bytes with two passes.
    PartGeom g0;
    if (!part_geometry(t->dv, &g0)) return KATGPU_OK;                              // direct path
    const uint32_t hb0 = g0.hb;                                                  // a table that grows has more regions: never more remainder bits
    const uint32_t passes0 = std::max<uint32_t>(1, g0.P1 / pass_buckets(g0.P1, (uint32_t)c->n_cu));   // (rounded down: the buffer never too small)
    const double per_item = arena_bytes_per_item(hb0, passes0);
    constexpr size_t SEG_PAD = 64;
    const size_t fixed_l1 = (size_t)W * MAX_PARTS * SEG_PAD;
    const size_t fixed_l2 = (size_t)((double)fixed_l1 * l2_items_per_l1_item(hb0) / passes0) + (size_t)MAX_PARTS * MAX_PARTS * P2_RUN_SLACK + 8192;
    const size_t small_bytes = align_up((size_t)W * MAX_PARTS * 4, 256) + align_up((size_t)W * MAX_PARTS * 8, 256) +   /* W <= 4 * CUs */
                               align_up((MAX_PARTS + 1) * 8, 256) + align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256) +
                               align_up((size_t)MAX_PARTS * MAX_PARTS * 4, 256) + align_up((size_t)MAX_PARTS * 8, 256) + align_up((size_t)MAX_PARTS * 4, 256) + 256 +
                               (fixed_l1 + fixed_l2 + 4096) * 8 + 4096;
    size_t want_items = n_starts;
    if (g_test_round_items) want_items = std::min<size_t>(want_items, g_test_round_items);
    size_t want_bytes = small_bytes + (size_t)((per_item + 0.5) * (double)want_items);
    if (c->arena_limit) want_bytes = std::min(want_bytes, std::max(c->arena_limit, small_bytes + 18 * ((size_t)64 << 20)));      // (the file feeders: more rounds, less to allocate)
    if (c->arena_bytes < want_bytes) {                                           // the arena could be more useful than it is
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        free_b += c->arena_bytes;
        size_t bytes = std::min<size_t>(want_bytes, (size_t)(g_arena_fraction * (double)free_b));
        // re-allocate only for a substantially larger arena (fewer rounds): a fresh hipMalloc of this size is not free
        if (bytes > c->arena_bytes + c->arena_bytes / 2 || c->arena_bytes < small_bytes + 18 * std::min<size_t>(want_items, (size_t)64 << 20)) {
            if (c->arena) { HIPCHK(c, hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
            if (!g_test_round_items && bytes < small_bytes + 18 * ((size_t)1 << 20)) return KATGPU_OK;   // no room for a useful round: direct path
            const double t_ar = now_ms();
            if (hipMalloc((void**)&c->arena, bytes) != hipSuccess) { (void)hipGetLastError(); c->arena = nullptr; return KATGPU_OK; }   // direct path
            c->arena_bytes = bytes;
            if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] partition arena of %.1f GB: %.0f ms\n", since_load(), bytes / 1e9, now_ms() - t_ar);
        }
    }
    struct Busy { katgpu_ctx* c; explicit Busy(katgpu_ctx* c_) : c(c_) { c->arena_busy = true; } ~Busy() { c->arena_busy = false; } } busy(c);
    uint8_t* a = c->arena;
    uint32_t* hist1 = (uint32_t*)a;               a += align_up((size_t)W * MAX_PARTS * 4, 256);
    uint64_t* offs = (uint64_t*)a;                a += align_up((size_t)W * MAX_PARTS * 8, 256);
    uint64_t* l1_off = (uint64_t*)a;              a += align_up((MAX_PARTS + 1) * 8, 256);
    uint64_t* off2 = (uint64_t*)a;                a += align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256);
    uint32_t* cnt2 = (uint32_t*)a;                a += align_up((size_t)MAX_PARTS * MAX_PARTS * 4, 256);
    uint64_t* bend = (uint64_t*)a;                a += align_up((size_t)MAX_PARTS * 8, 256);
    a += align_up((size_t)MAX_PARTS * 4, 256);
    unsigned long long* spill_n = (unsigned long long*)a;
    unsigned long long* ovf_n = spill_n + 1;      a += 256;
    if (g_apply_stamp) HIPCHK(c, hipMemset(spill_n + 8, 0, 7 * sizeof(unsigned long long)));
    const size_t round_items = std::min<size_t>(want_items, (size_t)((double)(c->arena_bytes - small_bytes) / per_item));
    const size_t l1_items = (round_items + round_items / 24 + fixed_l1 + 15) & ~(size_t)15;      // (a multiple of 16: the level-2 buffer starts on a 128-byte boundary)
    const size_t l2_items = ((size_t)((double)l1_items * l2_items_per_l1_item(hb0) / passes0) + (size_t)MAX_PARTS * MAX_PARTS * P2_RUN_SLACK + 4096 + 11) / 12 * 12;
    uint8_t* l1_buf = a;                                                            // level-1 items, groups of 4, 8 bytes of room per item (kg_partition.hpp "the level-1 buffer")
    uint8_t* l2_buf = l1_buf + l1_items * 8;                                        // level-2 items, groups of 4 (5-byte items: 64-byte blocks of 12)
    uint64_t* ovf_buf = (uint64_t*)(l2_buf + align_up(l2_buffer_bytes(hb0, l2_items), 16));
    const uint64_t ovf_cap = g_test_p2_ovf_cap ? g_test_p2_ovf_cap : round_items / 32 + 1024;
    bool p2_fast_ok = g_p2_fast != 0, l1_fast_ok = g_l1_fast != 0;
    if (!g_test_round_items && round_items < ((size_t)1 << 20) && round_items < n_starts) return KATGPU_OK;

    // Rounds are sized in ITEMS (valid k-mers), not window starts: a cheap pre-count of a prefix measures items/starts
    // (0.82 for 150 bp reads at k=27) so that the buffers are filled and the table is swept as few times as possible.
    double items_per_start = 1.0;
    size_t pos = 0;
    bool ratio_known = false;
    while (pos < n_starts) {
        int rc = refresh_counters(t);
        if (rc) return rc;
        if ((double)t->distinct > 0.6 * (double)t->dv.cap) {
            bool lost = false;
            rc = grow_beside_arena(t, 0, t->dv.cap * 2, KeyLists(), &lost);
            if (rc) return rc;
            if (lost) break;                                                      // the caller re-enters with a fresh arena
        }
        PartGeom g;
        if (!part_geometry(t->dv, &g)) break;                                      // table too large for two levels: direct path
        if (g.hb > hb0) break;                                                    // (cannot happen: see hb0) the level-2 carve would not hold these items
        // (the segmented level 1 sizes its segments from this ratio, so it wants it even when one round takes everything)
        if (!ratio_known && !g_test_round_items && (n_starts - pos > round_items || (l1_fast_ok && n_starts - pos >= ((size_t)64 << 20)))) {
            const size_t probe_m = std::min<size_t>(n_starts - pos, (size_t)64 << 20) / tile_starts * tile_starts;
            const uint64_t pt = probe_m / tile_starts, ptw = (pt + W - 1) / W;
            hipLaunchKernelGGL(k_p1v2_count, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->dv, g, dev_bases + pos, (uint64_t)(probe_m + k - 1), pt, ptw, hist1);
            hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            uint64_t probe_items = 0;
            HIPCHK(c, hipMemcpyAsync(&probe_items, &l1_off[g.P1], sizeof probe_items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // (the probe's all-ones tally must not count twice: the real count pass over the same prefix follows)
            if (t->dv.k == 32 && !t->dv.canonical) HIPCHK(c, hipMemcpyAsync(&t->dv.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            items_per_start = std::max(0.05, (double)probe_items / (double)probe_m);
            ratio_known = true;
        }
        size_t m = n_starts - pos;                                                 // items <= starts: this always fits
        if (m > round_items) m = std::min(m, (size_t)((double)round_items / items_per_start * 0.98));
        if (m < n_starts - pos) {
            const size_t rounds_left = (n_starts - pos + m - 1) / m;               // balance the remaining rounds
            m = (n_starts - pos + rounds_left - 1) / rounds_left;
            m += tile_starts - m % tile_starts;                                    // whole tiles, keeps the next round 16-byte aligned
            m = std::min(m, n_starts - pos);
        }
        const size_t nb = m + k - 1;
        const uint8_t* p = dev_bases + pos;
        t->count_bound = 0xFFFFFFFFULL;          // the apply kernel chains its own carries; a later direct launch sweeps first
        const uint64_t n_tiles = (m + tile_starts - 1) / tile_starts;
        const uint64_t tiles_per_wg = (n_tiles + W - 1) / W;
        // Level 1.  Segmented edition (one pass, fixed-capacity segments) when the round is big enough for its fixed costs; the
        // exact edition (count + scan + scatter) otherwise, and for the rest of the call once a segmented round overflowed.
        const uint64_t est_items = (uint64_t)((double)m * items_per_start);
        const bool lean = g_l1_lean && lean_applies(k, g.pl.n1);
        const bool pb512 = g.P1 <= 512;
        // The segmented edition's BLOCK form (kg_l1_blocks.hpp): 6-byte items in 64-byte blocks of ten, one 1024-thread workgroup per CU, 16 K-base tiles.
        const bool l1b = g_l1_blocks && lean && pb512 && g.hb1 == 2;
        const uint32_t Ws = l1b ? (uint32_t)c->n_cu : W;                            // workgroups of the segmented edition (<= W: the small arrays hold them)
        const uint64_t n_tiles_s = l1b ? (m + L1B_TILE_STARTS - 1) / L1B_TILE_STARTS : n_tiles;
        const uint64_t tiles_per_wg_s = (n_tiles_s + Ws - 1) / Ws;
        uint64_t seg_cap = est_items / ((uint64_t)Ws * g.P1);
        seg_cap += seg_cap / 24 + SEG_PAD;
        if (g_test_l1_cpb) seg_cap = std::min<uint64_t>(seg_cap, g_test_l1_cpb);
        if (l1b) seg_cap = (seg_cap + L1B_ITEMS - 1) / L1B_ITEMS * L1B_ITEMS;       // whole blocks: every slot may hold a k-mer
        const uint64_t cap_plain = seg_cap;                                         // k-mers a segment is expected to take at most (what the spill list must hold: 8 bytes each)
        // groups (kg_partition.hpp: k_p1v2_scatter): a bucket's k-mers of a tile are padded to whole groups, 1.5 items per tile and bucket on average
        if (!l1b && g.hb1 != 4 && !g_test_l1_cpb) seg_cap += std::min<uint64_t>(3 * seg_cap, 2 * tiles_per_wg_s);
        if (!l1b) seg_cap = (seg_cap + 3) & ~3ULL;                                  // whole groups
        const uint64_t stride64 = l1b ? align_up(8 * (uint64_t)Ws * cap_plain + 32, 64)      // (blocks start on 64-byte boundaries; 6.4 bytes per item lie inside the 8)
                                      : std::max<uint64_t>(8 * (uint64_t)Ws * cap_plain, (uint64_t)(4 + g.hb1) * Ws * seg_cap) + 32;   // bytes of a bucket (l1_bucket_base's + 32)
        const bool seg = l1_fast_ok && (g_l1_fast == 2 || (ratio_known && est_items >= ((uint64_t)64 << 20))) && stride64 * g.P1 <= (uint64_t)l1_items * 8 &&
                         seg_cap < (1u << 24) && stride64 <= 0xFFFFFFFFULL /* the kernel's segment arithmetic: 24 x 8 and 32 x 32 -> 64 bits */;
        const uint64_t seg_slots = seg ? (uint64_t)Ws * seg_cap : 0;               // items of one bucket
        const uint32_t bucket_stride = (uint32_t)stride64;
        g.l1_stride = seg ? stride64 : 0;
        g.l1_real = seg ? (uint64_t)Ws * cap_plain : 0;
        const bool l1_blocked = seg && l1b;                                         // what level 2 reads this round: blocks of ten, or groups
        uint64_t items = 0;
        unsigned long long ovf_l1 = 0;
        HIPCHK(c, hipMemsetAsync(spill_n, 0, 2 * sizeof(unsigned long long), c->stream));          // spill_n, ovf_n
        if (seg) {
            items = est_items;                                                    // the exact number is not needed (and not known)
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, items);
#define KG_L1S(LEAN, PB) hipLaunchKernelGGL((k_p1v2_scatter<true, LEAN, PB>), dim3(Ws), dim3(P1_BLOCK), 0, c->stream, t->dv, g, p, (uint64_t)nb, n_tiles_s, tiles_per_wg_s, \
                                               (const uint64_t*)nullptr, (const uint64_t*)nullptr, l1_buf, (uint32_t)seg_cap, bucket_stride, (uint32_t)cap_plain, ovf_buf, ovf_n, ovf_cap)
                if (l1b) {
                    if (g_l1b_stamp) {                                             // diagnostic: wave 0's cycles per phase
                        unsigned long long* stamps = spill_n + 16;
                        HIPCHK(c, hipMemsetAsync(stamps, 0, 10 * sizeof(unsigned long long), c->stream));
                        KG_LDS_ATTR(k_p1b_scatter<true>, sizeof(P1BLds));
                        hipLaunchKernelGGL(k_p1b_scatter<true>, dim3(Ws), dim3(L1B_THREADS), sizeof(P1BLds), c->stream, t->dv, g, p, (uint64_t)nb, n_tiles_s, tiles_per_wg_s,
                                           l1_buf, (uint32_t)(seg_cap / L1B_ITEMS), bucket_stride, ovf_buf, ovf_n, ovf_cap, stamps);
                        unsigned long long st[10];
                        HIPCHK(c, hipMemcpyAsync(st, stamps, sizeof st, hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                        double tot = 0;
                        for (int i = 0; i < 9; ++i) tot += (double)st[i];
                        if (st[9]) fprintf(stderr, "[katgpu] level-1 (blocks) stamps (wave 0 of every workgroup, cycles summed): codes %.1f %%  blocks out %.1f %% (+ barrier %.1f %%)  sweep %.1f %% (+ %.1f %%)  per bucket %.1f %% (+ %.1f %%)  placing %.1f %% (+ %.1f %%); %llu tiles, %.0f cycles per tile\n",
                                           100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot, 100 * st[3] / tot, 100 * st[4] / tot, 100 * st[5] / tot, 100 * st[6] / tot, 100 * st[7] / tot, 100 * st[8] / tot, st[9], tot / st[9]);
                    } else {
                        KG_LDS_ATTR(k_p1b_scatter<false>, sizeof(P1BLds));
                        hipLaunchKernelGGL(k_p1b_scatter<false>, dim3(Ws), dim3(L1B_THREADS), sizeof(P1BLds), c->stream, t->dv, g, p, (uint64_t)nb, n_tiles_s, tiles_per_wg_s,
                                           l1_buf, (uint32_t)(seg_cap / L1B_ITEMS), bucket_stride, ovf_buf, ovf_n, ovf_cap, (unsigned long long*)nullptr);
                    }
                } else
                if (lean) { if (pb512) KG_L1S(true, 512); else KG_L1S(true, MAX_PARTS); }
                else { if (pb512) KG_L1S(false, 512); else KG_L1S(false, MAX_PARTS); }
#undef KG_L1S
            }
            HIPCHK(c, hipMemcpyAsync(&ovf_l1, ovf_n, sizeof ovf_l1, hipMemcpyDeviceToHost, c->stream));      // read at the next synchronisation
            if (g_trace) fprintf(stderr, "[katgpu] partition round (segmented level 1%s): %zu starts, ~%llu items, %llu k-mers per segment (arena %.1f GB)\n", l1b ? ", blocks of ten" : "", m, (unsigned long long)items, (unsigned long long)seg_cap, c->arena_bytes / 1e9);
        } else {
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1, m);
                hipLaunchKernelGGL(k_p1v2_count, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->dv, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, hist1);
                hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            }
            HIPCHK(c, hipMemcpyAsync(&items, &l1_off[g.P1], sizeof items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (g_trace) fprintf(stderr, "[katgpu] partition round: %zu starts -> %llu items (buffer %zu items, arena %.1f GB, ratio %.3f)\n", m, (unsigned long long)items, round_items, c->arena_bytes / 1e9, items_per_start);
            if (items > round_items) {                      // denser than the prefix suggested: redo this round smaller
                if (t->dv.k == 32 && !t->dv.canonical) HIPCHK(c, hipMemcpyAsync(&t->dv.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                items_per_start = std::min(1.0, (double)items / (double)m * 1.02);
                continue;
            }
            if (items) {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, items);
                if (lean)
                    hipLaunchKernelGGL((k_p1v2_scatter<false, true>), dim3(W), dim3(P1_BLOCK), 0, c->stream, t->dv, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (const uint64_t*)offs, (const uint64_t*)l1_off, l1_buf,
                                       0u, 0u, 0u, (uint64_t*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
                else
                    hipLaunchKernelGGL((k_p1v2_scatter<false, false>), dim3(W), dim3(P1_BLOCK), 0, c->stream, t->dv, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (const uint64_t*)offs, (const uint64_t*)l1_off, l1_buf,
                                       0u, 0u, 0u, (uint64_t*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
            }
        }
        if (items) {
            // Level 2 + apply, in passes over sets of buckets: the level-2 buffer holds one pass (arena sizing above), a pass is a whole
            // number of CU-fulls of buckets where the geometry allows (alloc_dev_table).  Where bucket b starts in the level-1 buffer:
            std::vector<uint64_t> h_l1_off;
            if (!seg) {
                h_l1_off.resize(g.P1 + 1);
                HIPCHK(c, hipMemcpyAsync(h_l1_off.data(), l1_off, (g.P1 + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            auto lbeg = [&](uint32_t b) -> uint64_t { return seg ? (uint64_t)b * g.l1_real : h_l1_off[b]; };      // k-mers before bucket b (segmented: their bound)
            const uint32_t tile2 = l2_tile_items(g.hb);
            auto pass_extent = [&](uint32_t b_lo, uint32_t b_hi) -> uint64_t {       // bound of what level 2 writes for these buckets, in items (either edition)
                const uint64_t nn = lbeg(b_hi) - lbeg(b_lo);
                return nn + nn / 16 + 2ULL * g.P2 * (nn / tile2 + 1) + (uint64_t)(b_hi - b_lo) * g.P2 * P2_RUN_SLACK + 64;
            };
            if (seg) HIPCHK(c, hipStreamSynchronize(c->stream));                   // ovf_l1 has arrived
            uint32_t step = pass_buckets(g.P1, (uint32_t)c->n_cu);
            auto fits = [&](uint32_t st) { for (uint32_t b = 0; b < g.P1; b += st) if (pass_extent(b, std::min(g.P1, b + st)) > l2_items) return false; return true; };
            while (step > 1 && !fits(step)) step = (step + 1) / 2;
            if (!fits(step)) {                                                      // (a single bucket beyond the buffer: direct path)
                // this round's level 1 (either edition) has tallied the all-ones key of [pos, pos + m), which the direct kernel will count again
                if (seg || (t->dv.k == 32 && !t->dv.canonical)) HIPCHK(c, hipMemcpyAsync(&t->dv.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                break;
            }
            // level 2: one pass over the bucket when the runs are predictable (k_p2_fast), else -- or when its overflow list did not
            // hold -- the exact two-pass kernel
            const bool try_fast0 = p2_fast_ok && (g_p2_fast == 2 || items / g.R >= 1024);
            std::vector<std::pair<const uint64_t*, uint64_t>> lists;              // spilled k-mers: in the parts of the level-1 buffer that are dead
            unsigned long long ovf_total = ovf_l1;                                 // entries of the overflow list so far (level 1's, then every pass's)
            bool redo_round = false;
            if (g_trace && g.P1 > step) fprintf(stderr, "[katgpu]   level 2 + apply in %u passes of %u buckets (level-2 buffer: %zu items)\n", (g.P1 + step - 1) / step, step, l2_items);
            for (uint32_t b_lo = 0; b_lo < g.P1 && !redo_round; b_lo += step) {
                g.b_lo = b_lo; g.b_hi = std::min(g.P1, b_lo + step);
                const uint64_t pass_items = std::max<uint64_t>(1, (uint64_t)((double)items * (g.b_hi - g.b_lo) / g.P1));
                // this pass's part of the level-1 buffer, dead once its level 2 is through: the pass's spill list (room for 8 bytes per k-mer)
                auto l1_at = [&](uint32_t b) -> uint64_t { return seg ? (uint64_t)b * g.l1_stride : l1_bucket_base(lbeg(b), b); };
                uint64_t* spill_buf = (uint64_t*)(l1_buf + l1_at(b_lo));
                g.spill_cap = (l1_at(g.b_hi) - l1_at(b_lo)) / 8;
                const uint32_t* run_len = nullptr;
                unsigned long long overflowed = ovf_total;
                const bool try_fast = try_fast0 && p2_fast_ok;
                const uint32_t grid_l2 = std::min<uint32_t>(g.b_hi - g.b_lo, W2);
                HIPCHK(c, hipMemsetAsync(spill_n, 0, sizeof(unsigned long long), c->stream));
                if (try_fast) {
                    ScopedTimer tm(c, KATGPU_K_PART_L2, pass_items);
#define KG_P2F_B(HB) case HB: KG_LDS_ATTR((k_p2_fast<HB, false, false, true>), sizeof(P2FastLds<HB>::type)); \
                            hipLaunchKernelGGL((k_p2_fast<HB, false, false, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2FastLds<HB>::type), c->stream, g, l1_off, l1_buf, l2_buf, \
                                               off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, (unsigned long long*)nullptr); break;
#define KG_P2F(HB) case HB: if (g.hb1 == 4) hipLaunchKernelGGL((k_p2_fast<HB, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2FastLds<HB>::type), c->stream, g, l1_off, l1_buf, l2_buf, \
                                               off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, (unsigned long long*)nullptr); \
                            else hipLaunchKernelGGL((k_p2_fast<HB, false>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2FastLds<HB>::type), c->stream, g, l1_off, l1_buf, l2_buf, \
                                               off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, (unsigned long long*)nullptr); break;
                    if (g.hb == 1 && g.hb1 != 4 && g_p2_stamp) {                       // diagnostic: the bench's shape with cycle stamps
                        unsigned long long* stamps = spill_n + 16;
                        HIPCHK(c, hipMemsetAsync(stamps, 0, 6 * sizeof(unsigned long long), c->stream));
                        KG_LDS_ATTR((k_p2_fast<1, false, true>), sizeof(P2FastLds<1>::type));
                        KG_LDS_ATTR((k_p2_fast<1, false, true, true>), sizeof(P2FastLds<1>::type));
                        if (l1_blocked) hipLaunchKernelGGL((k_p2_fast<1, false, true, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2FastLds<1>::type), c->stream, g, l1_off, l1_buf, l2_buf, off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, stamps);
                        else
                        hipLaunchKernelGGL((k_p2_fast<1, false, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2FastLds<1>::type), c->stream, g, l1_off, l1_buf, l2_buf, off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, stamps);
                        unsigned long long st[6];
                        HIPCHK(c, hipMemcpyAsync(st, stamps, sizeof st, hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                        const double tot = (double)(st[0] + st[1] + st[2] + st[3] + st[4]);
                        if (st[5]) fprintf(stderr, "[katgpu] level-2 stamps (lane 0 of every workgroup, cycles summed): wait for the tile %.0f %%  digit + rank %.0f %%  scan %.0f %%  staging %.0f %%  copy-out %.0f %%; %llu tiles, %.0f cycles per tile\n",
                                           100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot, 100 * st[3] / tot, 100 * st[4] / tot, st[5], tot / st[5]);
                    } else if (l1_blocked && g.hb == 1 && g_p2x) {                   // the bench's shape: kg_l2_blocks.hpp
                        if (g_p2x_stamp) {
                            unsigned long long* stamps = spill_n + 16;
                            HIPCHK(c, hipMemsetAsync(stamps, 0, 10 * sizeof(unsigned long long), c->stream));
                            KG_LDS_ATTR(k_p2x_fast<true>, sizeof(P2XLds));
                            hipLaunchKernelGGL(k_p2x_fast<true>, dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2XLds), c->stream, g, l1_off, l1_buf, l2_buf, off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, stamps);
                            unsigned long long st[10];
                            HIPCHK(c, hipMemcpyAsync(st, stamps, sizeof st, hipMemcpyDeviceToHost, c->stream));
                            HIPCHK(c, hipStreamSynchronize(c->stream));
                            double tot = 0;
                            for (int i = 0; i < 9; ++i) tot += (double)st[i];
                            if (st[9]) fprintf(stderr, "[katgpu] level-2 (blocks in, blocks out) stamps (wave 0 of every workgroup, cycles summed): wait for the tile %.1f %% (+ barrier %.1f %%)  ranking + blocks out %.1f %% (+ %.1f %%)  per sub-bucket %.1f %% (+ %.1f %%)  placing %.1f %%; %llu tiles, %.0f cycles per tile\n",
                                               100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot, 100 * st[3] / tot, 100 * st[4] / tot, 100 * st[5] / tot, 100 * st[6] / tot, st[9], tot / st[9]);
                        } else {
                            KG_LDS_ATTR(k_p2x_fast<false>, sizeof(P2XLds));
                            hipLaunchKernelGGL(k_p2x_fast<false>, dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2XLds), c->stream, g, l1_off, l1_buf, l2_buf, off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, (unsigned long long*)nullptr);
                        }
                    } else if (l1_blocked) {
                        switch (g.hb) { KG_P2F_B(0) KG_P2F_B(1) KG_P2F_B(2) default: return fail(c, KATGPU_ERR_DEVICE, "level 2 from blocked level-1 items: item width %u", g.hb); }
                    } else
                    switch (g.hb) { KG_FOR_HB(KG_P2F) }
#undef KG_P2F
#undef KG_P2F_B
                }
                if (try_fast || (seg && b_lo == 0)) {
                    HIPCHK(c, hipMemcpyAsync(&overflowed, ovf_n, sizeof overflowed, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (seg && ovf_l1 > ovf_cap) {               // the level-1 buffer itself is incomplete: this round again, exactly
                        if (g_trace) fprintf(stderr, "[katgpu] segmented level 1: %llu k-mers beyond their segments (list holds %llu): exact level 1 from here on\n", ovf_l1, (unsigned long long)ovf_cap);
                        l1_fast_ok = false;
                        HIPCHK(c, hipMemcpyAsync(&t->dv.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));   // the scatter tallied the all-ones key
                        redo_round = true;                       // (only ever in the first pass: nothing has been applied yet)
                        break;
                    }
                    if (try_fast && overflowed <= ovf_cap) run_len = cnt2;
                    else if (try_fast) {
                        if (g_trace) fprintf(stderr, "[katgpu] k_p2_fast: %llu k-mers beyond their runs (list holds %llu): exact level 2 from here on\n", overflowed, (unsigned long long)ovf_cap);
                        p2_fast_ok = false;
                        overflowed = ovf_total;                  // what was on the list before this pass is still there and still valid
                        HIPCHK(c, hipMemcpyAsync(ovf_n, &ovf_total, sizeof ovf_total, hipMemcpyHostToDevice, c->stream));
                    }
                }
                if (!run_len) {
                    ScopedTimer tm(c, KATGPU_K_PART_L2, pass_items);
#define KG_P2(HB) case HB: if (g.hb1 == 4) hipLaunchKernelGGL((k_p2<HB, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<HB>), c->stream, g, l1_off, l1_buf, l2_buf, off2, seg_slots, bend); \
                           else hipLaunchKernelGGL((k_p2<HB, false>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<HB>), c->stream, g, l1_off, l1_buf, l2_buf, off2, seg_slots, bend); break;
#define KG_P2_B(HB) case HB: KG_LDS_ATTR((k_p2<HB, false, true>), sizeof(P2Lds<HB>)); \
                           hipLaunchKernelGGL((k_p2<HB, false, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<HB>), c->stream, g, l1_off, l1_buf, l2_buf, off2, seg_slots, bend); break;
                    if (l1_blocked) { switch (g.hb) { KG_P2_B(0) KG_P2_B(1) KG_P2_B(2) default: return fail(c, KATGPU_ERR_DEVICE, "level 2 from blocked level-1 items: item width %u", g.hb); } }
                    else
                    switch (g.hb) { KG_FOR_HB(KG_P2) }
#undef KG_P2
#undef KG_P2_B
                }
                ovf_total = overflowed;
                const uint64_t* bucket_end = !run_len ? bend : nullptr;             // exact level 2: a bucket's runs stop short of the next bucket's
                {
                    ScopedTimer tm(c, KATGPU_K_PART_APPLY, pass_items);
                    rc = launch_apply(t, g, off2, l2_buf, spill_buf, spill_n, run_len, bucket_end);
                    if (rc) return rc;
                }
                HIPCHK(c, hipGetLastError());
                unsigned long long spilled = 0;
                HIPCHK(c, hipMemcpyAsync(&spilled, spill_n, sizeof spilled, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (g_apply_stamp) {
                    unsigned long long st[7];
                    HIPCHK(c, hipMemcpy(st, spill_n + 8, sizeof st, hipMemcpyDeviceToHost));
                    HIPCHK(c, hipMemset(spill_n + 8, 0, sizeof st));
                    const double tot = (double)(st[0] + st[1] + st[3] + st[4]);
                    if (st[5]) fprintf(stderr, "[katgpu] apply stamps (wave 0 of every workgroup, cycles summed): fill %.3g (%.0f %%)  walk %.3g (%.0f %%; drains %.0f %% of the walk)  barrier + sweep %.3g (%.0f %%)  write-back %.3g (%.0f %%); %llu regions, %.1f chunks per wave and region, %.0f cycles per region\n",
                                       (double)st[0], 100 * st[0] / tot, (double)st[1], 100 * st[1] / tot, 100.0 * st[2] / std::max(1.0, (double)st[1]), (double)st[3], 100 * st[3] / tot, (double)st[4], 100 * st[4] / tot, st[5], (double)st[6] / st[5], tot / st[5]);
                }
                if (spilled > g.spill_cap)       // (more k-mers without a slot than the pass's segments were sized for: a table far too small, met by a 5-sigma round)
                    return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full: %llu k-mers of a partition pass found no slot (the list holds %llu); raise the size hint", spilled, (unsigned long long)g.spill_cap);
                if (spilled) lists.push_back({spill_buf, spilled});
            }
            if (redo_round) continue;
            if (ovf_total) lists.push_back({ovf_buf, ovf_total});                  // what level 1 / level 2 could not place
            if (!lists.empty()) {                      // regions that ran out of slots, runs beyond their capacity: make room, then the direct path
                uint64_t total = 0;
                for (auto& l : lists) total += l.second;
                bool lost = false;
                rc = grow_beside_arena(t, total, 0, lists, &lost);
                if (rc) return rc;
                if (lost) { pos += m; break; }         // the lists went in from the host; the caller re-enters for the rest
                for (auto& l : lists) {
                    ScopedTimer tm(c, KATGPU_K_COUNT, l.second);
                    hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(c, l.second, 256, 6)), dim3(256), 0, c->stream, t->dev(), l.first, (uint64_t)l.second);
                }
            }
        }
        pos += m;
    }
    *done = pos;
    return refresh_counters(t);
}

// The same for wide tables (kg_partition_wide.hpp): 16-byte items, exact level 1 and level 2, one pass of level 2 + apply per round.
// Arena: [hist1 | offs | l1_off | off2 | spill_n | level-1 buffer | level-2 buffer], 32 bytes per k-mer of a round; the spill list
// of a round lies in its level-1 buffer, which is dead by then.
static bool wide_part_geometry(const DevTable& d) {
    return d.keys_b && d.p1 <= (uint32_t)MAX_PARTS && d.p2 <= (uint32_t)MAX_PARTS && d.region_slots % 4 == 0 && d.region_slots >= 64 && d.region_slots <= WIDE_AP_MAX_SLOTS &&
           (uint64_t)d.p1 * d.p2 == d.n_regions;
}
static int count_partitioned_w(katgpu_table* t, const uint8_t* dev_bases, size_t n, size_t* done) {
    katgpu_ctx* c = t->ctx;
    const uint32_t k = t->dev().k;
    const size_t n_starts = n - k + 1;
    *done = 0;
    c->arena_borrowed = false;
    if (n < 4096 || !wide_part_geometry(t->dev())) return KATGPU_OK;               // direct path
    if (!g_test_round_items && !t->disable_grow && t->dev().cap < n_starts / 16) {     // (as count_partitioned: room for 1/16 of the starts first)
        uint64_t nc = t->dev().cap; while (nc < n_starts / 16) nc *= 2;
        int grc = regrow(t, nc);
        if (grc) return grc;
        if (!wide_part_geometry(t->dev())) return KATGPU_OK;
    }
    const uint32_t W = (uint32_t)c->n_cu * 3;                                  // level-1 workgroups (512 threads, 12 KB of LDS)
    const size_t tile_starts = W1_TILE_STARTS;
    const size_t small_bytes = align_up((size_t)W * MAX_PARTS * 4, 256) + align_up((size_t)W * MAX_PARTS * 8, 256) + align_up((MAX_PARTS + 1) * 8, 256) +
                               align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256) + 256 + 4096;
    size_t want_items = n_starts;
    if (g_test_round_items) want_items = std::min<size_t>(want_items, g_test_round_items);
    size_t want_bytes = small_bytes + 32 * (want_items + 64);
    if (c->arena_limit) want_bytes = std::min(want_bytes, std::max(c->arena_limit, small_bytes + 32 * ((size_t)64 << 20)));
    if (c->arena_bytes < want_bytes) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        free_b += c->arena_bytes;
        const size_t bytes = std::min<size_t>(want_bytes, (size_t)(g_arena_fraction * (double)free_b));
        if (bytes > c->arena_bytes + c->arena_bytes / 2 || c->arena_bytes < small_bytes + 32 * std::min<size_t>(want_items + 64, (size_t)64 << 20)) {
            if (c->arena) { HIPCHK(c, hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
            if (!g_test_round_items && bytes < small_bytes + 32 * ((size_t)1 << 20)) return KATGPU_OK;
            if (hipMalloc((void**)&c->arena, bytes) != hipSuccess) { (void)hipGetLastError(); c->arena = nullptr; return KATGPU_OK; }
            c->arena_bytes = bytes;
            if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] partition arena of %.1f GB (wide k-mers)\n", since_load(), bytes / 1e9);
        }
    }
    struct Busy { katgpu_ctx* c; explicit Busy(katgpu_ctx* c_) : c(c_) { c->arena_busy = true; } ~Busy() { c->arena_busy = false; } } busy(c);
    uint8_t* a = c->arena;
    uint32_t* hist1 = (uint32_t*)a;               a += align_up((size_t)W * MAX_PARTS * 4, 256);
    uint64_t* offs = (uint64_t*)a;                a += align_up((size_t)W * MAX_PARTS * 8, 256);
    uint64_t* l1_off = (uint64_t*)a;              a += align_up((MAX_PARTS + 1) * 8, 256);
    uint64_t* off2 = (uint64_t*)a;                a += align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256);
    unsigned long long* spill_n = (unsigned long long*)a; a += 256;
    const size_t round_items = std::min<size_t>(want_items, (c->arena_bytes - small_bytes) / 32);
    u64x2* l1_buf = (u64x2*)a;
    u64x2* l2_buf = l1_buf + round_items;
    if (!g_test_round_items && round_items < ((size_t)1 << 20) && round_items < n_starts) return KATGPU_OK;
    if (round_items < tile_starts && round_items < n_starts) return KATGPU_OK;

    size_t pos = 0;
    while (pos < n_starts) {
        int rc = refresh_counters(t);
        if (rc) return rc;
        if ((double)t->distinct > 0.6 * (double)t->dev().cap) {
            bool lost = false;
            rc = grow_beside_arena(t, 0, t->dev().cap * 2, KeyLists(), &lost);
            if (rc) return rc;
            if (lost) break;                                                      // the caller re-enters with a fresh arena
        }
        if (!wide_part_geometry(t->dev())) break;
        const DevTable d = t->dev();
        size_t m = std::min(n_starts - pos, round_items);                          // items <= starts: a round always fits its buffers
        if (m < n_starts - pos) {
            const size_t rounds_left = (n_starts - pos + m - 1) / m;               // balance the remaining rounds
            m = std::min(m, (n_starts - pos + rounds_left - 1) / rounds_left + tile_starts);
            m -= m % tile_starts;                                                  // whole tiles: the next round starts 16-byte aligned
            if (!m) break;
        }
        const size_t nb = m + k - 1;
        const uint8_t* p = dev_bases + pos;
        const uint64_t n_tiles = (m + tile_starts - 1) / tile_starts;
        const uint64_t tiles_per_wg = (n_tiles + W - 1) / W;
        PartGeom g{};
        g.P1 = d.p1;                                                               // (all k_p1_scan looks at)
        HIPCHK(c, hipMemsetAsync(spill_n, 0, sizeof(unsigned long long), c->stream));
        {
            ScopedTimer tm(c, KATGPU_K_PART_L1, m);
            hipLaunchKernelGGL(k_w1<false>, dim3(W), dim3(W1_BLOCK), 0, c->stream, d, d.p1, p, (uint64_t)nb, n_tiles, tiles_per_wg, hist1, (const uint64_t*)nullptr, (u64x2*)nullptr);
            hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
        }
        uint64_t items = 0;
        HIPCHK(c, hipMemcpyAsync(&items, &l1_off[d.p1], sizeof items, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (g_trace) fprintf(stderr, "[katgpu] partition round (wide k-mers): %zu starts -> %llu items (buffer %zu items, arena %.1f GB)\n", m, (unsigned long long)items, round_items, c->arena_bytes / 1e9);
        if (items > round_items) return fail(c, KATGPU_ERR_DEVICE, "wide partition round: %llu items from %zu starts", (unsigned long long)items, m);
        if (items) {
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, items);
                hipLaunchKernelGGL(k_w1<true>, dim3(W), dim3(W1_BLOCK), 0, c->stream, d, d.p1, p, (uint64_t)nb, n_tiles, tiles_per_wg, (uint32_t*)nullptr, (const uint64_t*)offs, l1_buf);
            }
            {
                ScopedTimer tm(c, KATGPU_K_PART_L2, items);
                hipLaunchKernelGGL(k_w2, dim3(std::min<uint32_t>(d.p1, (uint32_t)c->n_cu)), dim3(PART_BLOCK), 0, c->stream, d.p1, d.p2, (const uint64_t*)l1_off, (const u64x2*)l1_buf, l2_buf, off2);
            }
            {
                const size_t lds = (size_t)d.region_slots * 20;
                KG_LDS_ATTR(k_w3_apply, LDS_BYTES - 256);
                ScopedTimer tm(c, KATGPU_K_PART_APPLY, items);
                hipLaunchKernelGGL(k_w3_apply, dim3(std::min<uint32_t>(d.n_regions, (uint32_t)c->n_cu)), dim3(W3_BLOCK), lds, c->stream, d, (const uint64_t*)off2, (const u64x2*)l2_buf,
                                   l1_buf /* the spill list: the level-1 buffer is dead */, spill_n, g_test_spill_mod);
            }
            HIPCHK(c, hipGetLastError());
            unsigned long long spilled = 0;
            HIPCHK(c, hipMemcpyAsync(&spilled, spill_n, sizeof spilled, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (spilled) {                              // regions that ran out of slots: make room, then the direct path
                if (g_trace) fprintf(stderr, "[katgpu]   %llu k-mers spilled by full regions\n", spilled);
                rc = g_test_grow_nomem ? KATGPU_ERR_NOMEM : ensure_room(t, spilled);
                if (rc == KATGPU_ERR_NOMEM) {           // no room for the larger table beside the arena: park the list on the host, give the arena up
                    (void)hipGetLastError();
                    std::vector<uint64_t> host;
                    try { host.resize((size_t)spilled * 2); } catch (...) { return fail(c, KATGPU_ERR_NOMEM, "no host memory to park %llu spilled k-mers", spilled); }
                    HIPCHK(c, hipMemcpy(host.data(), l1_buf, (size_t)spilled * 16, hipMemcpyDeviceToHost));
                    release_arena(c);
                    rc = ensure_room(t, spilled);
                    if (rc) return rc;
                    const size_t chunk = std::min<size_t>(spilled, (size_t)16 << 20);
                    u64x2* dbuf = nullptr;
                    HIPCHK(c, pool_alloc(c, (void**)&dbuf, chunk * 16));
                    for (size_t i = 0; i < spilled && rc == KATGPU_OK; i += chunk) {
                        const size_t mm = std::min(chunk, (size_t)spilled - i);
                        if (hipMemcpyAsync(dbuf, host.data() + 2 * i, mm * 16, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(c, KATGPU_ERR_DEVICE, "spill upload"); break; }
                        ScopedTimer tm(c, KATGPU_K_COUNT, mm);
                        hipLaunchKernelGGL(k_insert_keys_w, dim3(grid_for(c, mm, 256, 6)), dim3(256), 0, c->stream, t->dev(), (const u64x2*)dbuf, (uint64_t)mm);
                        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "spill insert");
                    }
                    pool_release(c, dbuf);
                    if (rc) return rc;
                    pos += m;
                    break;                              // the caller re-enters with a fresh arena
                }
                if (rc) return rc;
                ScopedTimer tm(c, KATGPU_K_COUNT, spilled);
                hipLaunchKernelGGL(k_insert_keys_w, dim3(grid_for(c, spilled, 256, 6)), dim3(256), 0, c->stream, t->dev(), (const u64x2*)l1_buf, (uint64_t)spilled);
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
        }
        pos += m;
    }
    *done = pos;
    return refresh_counters(t);
}

// Count a resident base stream.  The stream is cut into sub-batches so that "distinct + sub-batch starts" stays under
// the load limit (the table can then never fill in the middle of a launch); consecutive sub-batches overlap by k-1.
int count_resident(katgpu_table* t, const uint8_t* dev_bases, size_t n) {
    const uint32_t k = t->dv.k;                              // (dv, not dev(): a table whose slots wait for their first sweep is left to the partitioned counter)
    if (n < k) return KATGPU_OK;
    size_t pos = 0;
    const size_t n_starts = n - k + 1;
    // Large, aligned inputs go through the partitioned counter (no global atomic per k-mer); whatever it leaves (nothing,
    // normally) and everything small goes through the direct kernel below.
    while (!t->dv.keys_b && n_starts - pos >= std::max<uint64_t>(g_part_min_starts, 1) && (reinterpret_cast<uintptr_t>(dev_bases + pos) & 15) == 0) {
        size_t done = 0;                        // returns early (done < remaining) when a table growth cost it the arena
        int prc = count_partitioned(t, dev_bases + pos, n - pos, &done);
        if (prc) return prc;
        if (!done) break;
        pos += done;
    }
    while (t->dv.keys_b && n_starts - pos >= std::max<uint64_t>(g_part_min_starts, 1) && (reinterpret_cast<uintptr_t>(dev_bases + pos) & 15) == 0) {
        size_t done = 0;                        // wide tables: kg_partition_wide.hpp
        int prc = count_partitioned_w(t, dev_bases + pos, n - pos, &done);
        if (prc) return prc;
        if (!done) break;
        pos += done;
    }
    while (pos < n_starts) {
        int rc = refresh_counters(t);
        if (rc) return rc;
        // largest batch that provably fits; if even a minimal one does not, grow first
        uint64_t room = (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) > t->distinct ? (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) - t->distinct : 0;
        uint64_t want = std::min<uint64_t>(n_starts - pos, (uint64_t)CHUNK_STARTS * 65536);   // <= 266 M starts per launch
        if (g_test_max_starts) want = std::min<uint64_t>(want, g_test_max_starts);
        // As the table fills, launches shrink to the remaining room (each adds far fewer distinct k-mers than window
        // starts on real coverage, so the room shrinks slowly); only when the room is down to 1/64 of the table do we grow.
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 64, CHUNK_STARTS))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 2, CHUNK_STARTS)));
            if (rc) return rc;
            continue;
        }
        uint64_t starts = std::min(want, room);
        if (starts < n_starts - pos) starts -= starts % 16;            // keep the next sub-batch 16-byte aligned
        if (starts == 0) starts = std::min<uint64_t>(16, n_starts - pos);
        rc = launch_count(t, dev_bases + pos, (size_t)(starts + k - 1));
        if (rc) return rc;
        pos += starts;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_count_bases_device(katgpu_table* t, const uint8_t* dev_bases, size_t n) {
    if (!t || (!dev_bases && n)) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    return count_resident(t, dev_bases, n);
}

// KATGPU_RING_MB: size of each of the two device rings the host feeder fills (default 1024)
static const size_t g_ring_bytes = (getenv("KATGPU_RING_MB") ? std::max<size_t>(1, strtoull(getenv("KATGPU_RING_MB"), nullptr, 10)) : 1024) << 20;

// want: bytes of stream the caller expects (0: unknown): small inputs get small rings (two 1 GiB rings for a 100-base call, or for
// the CLI on the reference's 1000-read test files, would be most of the call's time and could fail on a full device)
static int ensure_staging(katgpu_ctx* c, size_t want = 0) {
    if (!c->stage_bytes) {
        const size_t bytes = (size_t)64 << 20;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipHostMalloc((void**)&c->pinned[i], bytes, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&c->pin_free[i], hipEventDisableTiming));
        }
        c->stage_bytes = bytes;
    }
    size_t ring_want = g_ring_bytes;
    if (want) ring_want = std::min(g_ring_bytes, std::max<size_t>((size_t)16 << 20, align_up(want + 4096, (size_t)16 << 20)));
    if (c->ring_bytes && c->ring_bytes < ring_want) {              // grown for a bigger input
        for (int i = 0; i < 2; ++i) { hipFree(c->ring[i]); c->ring[i] = nullptr; }
        c->ring_bytes = 0;
    }
    for (; !c->ring_bytes; ring_want /= 2) {                       // halve on failure: a smaller ring is only more count calls
        if (ring_want < ((size_t)1 << 20)) return fail(c, KATGPU_ERR_NOMEM, "no device memory for the staging rings");
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) {
            hipError_t e = hipMalloc((void**)&c->ring[i], ring_want);
            if (e != hipSuccess && c->arena && !c->arena_borrowed && !c->arena_busy) {       // the cached arena holds most of the free HBM: give it back
                (void)hipGetLastError();
                hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0;
                e = hipMalloc((void**)&c->ring[i], ring_want);
            }
            if (e != hipSuccess) { (void)hipGetLastError(); for (int j = 0; j < i; ++j) { hipFree(c->ring[j]); c->ring[j] = nullptr; } ok = false; }
        }
        if (ok) c->ring_bytes = ring_want;
    }
    return KATGPU_OK;
}

// Host base stream -> table.  The stream is copied through two pinned buffers (64 MiB each) into one of two DEVICE RINGS; a full
// ring is a resident stretch of the stream and goes to count_resident -- the partitioned counter for anything of size, exactly
// what a caller with device-resident input gets -- on a worker thread, while the feeder (and the parser team behind it) fills
// the other ring.  A ring starts with the previous ring's last k-1 bytes, so windows across the cut are counted once.
struct HostFeeder {
    katgpu_table* t; katgpu_ctx* c;
    int cur = 0; size_t fill = 0; bool pin_used[2] = {false, false};
    int ring_cur = 0; size_t ring_fill = 0;
    uint8_t tail[64]; uint32_t tail_n = 0;          // last k-1 bytes of the stream so far
    static constexpr size_t HEAD = 64;               // carry area (k - 1 <= 62 bytes) in front of a ring's payload: keeps it 16-byte aligned
    // worker
    std::thread worker;
    std::mutex mu; std::condition_variable cv;
    struct Job { int ring; size_t n; };
    std::deque<Job> jobs;
    bool ring_busy[2] = {false, false};
    bool stop = false;
    int worker_rc = KATGPU_OK; std::string worker_err;

    explicit HostFeeder(katgpu_table* t_) : t(t_), c(t_->ctx) {}
    ~HostFeeder() { shutdown(); }

    void run() {
        hipSetDevice(c->device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                j = jobs.front(); jobs.pop_front();
            }
            int rc = worker_rc ? worker_rc : count_resident(t, c->ring[j.ring], j.n);     // (after an error the queued rings are dropped, not counted)
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc && !worker_rc) { worker_rc = rc; worker_err = c->err; }
                ring_busy[j.ring] = false;
            }
            cv.notify_all();
        }
    }
    void shutdown() {
        if (!worker.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        worker.join();
    }
    int begin(size_t want = 0) {
        int rc = ensure_staging(c, want); if (rc) return rc;
        tail_n = 0;
        worker = std::thread([this] { run(); });
        return open_ring();
    }
    int open_ring() {                                 // ring_cur is free: seed its head with the carry
        uint8_t head[HEAD];
        memset(head, 'N', HEAD);
        memcpy(head + HEAD - tail_n, tail, tail_n);
        HIPCHK(c, hipMemcpyAsync(c->ring[ring_cur], head, HEAD, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));          // `head` is on this stack
        ring_fill = HEAD;
        return KATGPU_OK;
    }
    int submit_ring() {                               // hand the current ring to the worker, move on to the other one
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));          // every copy into it has landed
        const int other = ring_cur ^ 1;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (ring_fill > HEAD) { ring_busy[ring_cur] = true; jobs.push_back({ring_cur, ring_fill}); }
            cv.notify_all();
            cv.wait(lk, [&] { return !ring_busy[other]; });
            if (worker_rc) return fail(c, worker_rc, "%s", worker_err.c_str());
        }
        if (ring_fill > HEAD) ring_cur = other;
        return open_ring();
    }
    int push(const uint8_t* p, size_t n) {
        while (n) {
            if (fill == 0 && pin_used[cur]) { hipError_t e = hipEventSynchronize(c->pin_free[cur]); if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e)); }
            const size_t room = c->stage_bytes - fill, take = std::min(room, n);
            memcpy(c->pinned[cur] + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == c->stage_bytes) { int rc = flush(); if (rc) return rc; }
        }
        return KATGPU_OK;
    }
    int flush() {                                     // pinned[cur][0, fill) -> the current ring (asynchronously)
        if (!fill) return KATGPU_OK;
        const uint32_t want = t->dv.k - 1;
        size_t off = 0;
        while (off < fill) {
            if (ring_fill == c->ring_bytes) { int rc = submit_ring(); if (rc) return rc; }      // its head = `tail`, the k-1 bytes before `off`
            const size_t take = std::min(fill - off, c->ring_bytes - ring_fill);
            const uint8_t* src = c->pinned[cur] + off;
            HIPCHK(c, hipMemcpyAsync(c->ring[ring_cur] + ring_fill, src, take, hipMemcpyHostToDevice, c->copy_stream));
            ring_fill += take; off += take;
            // the last k-1 bytes of the stream that is in the rings so far
            if (take >= want) { memcpy(tail, src + take - want, want); tail_n = want; }
            else {
                uint8_t tmp[128]; const uint32_t keep = (uint32_t)std::min<size_t>(tail_n, want - take);
                memcpy(tmp, tail + tail_n - keep, keep); memcpy(tmp + keep, src, take);
                tail_n = keep + (uint32_t)take; memcpy(tail, tmp, tail_n);
            }
        }
        HIPCHK(c, hipEventRecord(c->pin_free[cur], c->copy_stream));
        pin_used[cur] = true;
        cur ^= 1; fill = 0;
        return KATGPU_OK;
    }
    int end_of_file() {             // files of a group never join (mer_overlap_sequence_parser.hpp:151-155: have_seam = false)
        static const uint8_t sep = 'N';
        return push(&sep, 1);
    }
    int finish() {
        int rc = flush(); if (rc) { shutdown(); return rc; }
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        {
            std::unique_lock<std::mutex> lk(mu);
            if (ring_fill > HEAD) { ring_busy[ring_cur] = true; jobs.push_back({ring_cur, ring_fill}); }
            cv.notify_all();
            cv.wait(lk, [&] { return !ring_busy[0] && !ring_busy[1] && jobs.empty(); });
        }
        shutdown();
        if (worker_rc) return fail(c, worker_rc, "%s", worker_err.c_str());
        t->carry_n = 0;
        return refresh_counters(t);
    }
};

extern "C" int katgpu_count_bases_host(katgpu_table* t, const uint8_t* bases, size_t n) {
    if (!t || (!bases && n)) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    t->carry_n = 0;
    HostFeeder f(t);
    int rc = f.begin(n); if (rc) return rc;
    rc = f.push(bases, n); if (rc) return rc;
    return f.finish();
}

// rank / world: this process's share of the group in a multi-GPU run.  Plain FASTQ files big enough for the device scan are cut
// between the ranks batch by batch (kg_scan.hip); every other file goes whole to rank (index mod world).
static int count_files_impl(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p, int rank, int world) {
    if (!t || !paths || world < 1 || rank < 0 || rank >= world) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    t->carry_n = 0;
    // Large plain FASTQ / FASTA files: raw bytes to the device, the record scan there (kg_scan.hip).  Files of a group never join and
    // the table is a multiset, so the order in which the group's files are counted is free.
    std::vector<const char*> rest;
    std::vector<uint16_t> rest_trim;
    size_t rest_bytes = 0;
    size_t whole = 0;                                              // files that are dealt whole, counted over the group
    for (size_t i = 0; i < n_paths; ++i) {
        const uint32_t trim = trim5p ? trim5p[i] : 0;
        bool took = false;
        uint8_t first = 0;
        if (world > 1 && device_scan_applies(paths[i], trim, nullptr, &first) && first == '@') {
            // a FASTQ file the ranks share out batch by batch: every rank takes this branch (the test is a property of the file), and a
            // rank that cannot -- no room for its batch buffers -- fails the run: falling back to the whole-file dealing on ONE rank
            // would count the file's batches twice or not at all, and put that rank's file counter out of step with the others'
            int rc = count_file_device_scan(t, paths[i], trim, &took, rank, world);
            if (rc) return rc;
            if (!took) return fail(c, KATGPU_ERR_NOMEM, "rank %d of %d: no device memory for the batch buffers of %s (a file the ranks share out cannot fall back to the streaming reader on one of them)", rank, world, paths[i]);
            continue;
        }
        if (world > 1 && (int)(whole++ % (size_t)world) != rank) continue;       // another rank's file
        int rc = count_file_device_scan(t, paths[i], trim, &took);
        if (rc) return rc;
        if (!took) { rest.push_back(paths[i]); rest_trim.push_back((uint16_t)trim); rest_bytes += (size_t)kg::file_size_or_zero(paths[i]); }
    }
    if (rest.empty()) { int wrc = table_wait(t); return wrc ? wrc : refresh_counters(t); }
    { int wrc = table_wait(t); if (wrc) return wrc; }                // (the streaming feeder does not overlap the allocation)
    HostFeeder f(t);
    int rc = f.begin(rest_bytes * 4); if (rc) return rc;           // (gzip inflates: be generous)
    // the group's other files -> one base stream (kg_ingest.hpp: thread team for large plain files, concurrent readers for gzip & co.)
    std::string err;
    rc = kg::stream_group(rest.data(), rest.size(), trim5p ? rest_trim.data() : nullptr, t->dv.k, [&](const uint8_t* p, size_t n) { return f.push(p, n); }, &err);
    if (rc) return err.empty() ? rc : fail(c, rc, "%s", err.c_str());
    return f.finish();
}

extern "C" int katgpu_count_files(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p) {
    return count_files_impl(t, paths, n_paths, trim5p, 0, 1);
}
extern "C" int katgpu_count_files_sharded(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p, int rank, int world) {
    return count_files_impl(t, paths, n_paths, trim5p, rank, world);
}

extern "C" int katgpu_count(katgpu_ctx* c, const char* const* paths, size_t n_paths, uint32_t k, int canonical,
                            const uint16_t* trim5p, uint64_t size_hint, int disable_grow, katgpu_table** out) {
    if (!c || !out || !paths) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (size_hint == 0) {        // every input byte starts at most one new k-mer
        uint64_t bytes = 0;
        for (size_t i = 0; i < n_paths; ++i) bytes += kg::file_size_or_zero(paths[i]);
        size_hint = std::max<uint64_t>(1u << 20, bytes);
    }
    // Big plain files: the table (and the feeders' arena) are allocated on a thread of their own while the device scan already reads
    // and parses into its accumulation buffers -- tens of GB of hipMalloc take about as long as the first GBs of the files take to arrive.
    uint64_t scan_bytes = 0;
    uint8_t scan_first = 0;
    for (size_t i = 0; i < n_paths; ++i) { uint64_t sz = 0; uint8_t fb = 0; if (device_scan_applies(paths[i], trim5p ? trim5p[i] : 0, &sz, &fb)) { scan_bytes += sz; if (!scan_first) scan_first = fb; } }
    katgpu_table* t = nullptr;
    int rc;
    if (scan_bytes >= ((uint64_t)4 << 30) && k >= 1 && k <= KATGPU_MAX_K && !getenv("KATGPU_SYNC_ALLOC")) {
        HIPCHK(c, hipSetDevice(c->device));
        t = new katgpu_table();
        t->ctx = c; t->disable_grow = disable_grow;
        t->dv.k = k; t->dv.canonical = canonical ? 1 : 0;                    // what the feeders' own threads look at before the slots exist
        const uint64_t cap = std::max<uint64_t>(size_hint, 1024);
        const size_t arena_bytes = scan_arena_bytes(scan_first);
        c->scan_waiting.store(1, std::memory_order_release);                  // the scan buffers go first (katgpu_ctx::scan_waiting; lowered by the first feeder's setup)
        c->big_alloc_running.store(1, std::memory_order_release);
        t->alloc_thread = std::thread([c, t, k, canonical, cap, arena_bytes]() {
            struct Done { katgpu_ctx* c; ~Done() { c->big_alloc_running.store(0, std::memory_order_release); } } done{c};
            hipSetDevice(c->device);
            alloc_turn(c, false, 3000.0);
            DevTable d{};
            bool lazy = false;
            const int arc = alloc_dev_table(c, k, canonical, cap, &d, 0, 0, &lazy);
            if (arc) { t->alloc_rc = arc; t->alloc_err = c->err; return; }
            t->dv = d;
            if (lazy) t->zero_from = 0;
            if (!c->arena && hipMalloc((void**)&c->arena, arena_bytes) == hipSuccess) c->arena_bytes = arena_bytes; else (void)hipGetLastError();
            if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] table and arena allocated (beside the feeders)\n", since_load());
        });
        rc = KATGPU_OK;
    } else
        rc = katgpu_table_create(c, k, canonical, size_hint, disable_grow, &t);
    if (rc) return rc;
    rc = katgpu_count_files(t, paths, n_paths, trim5p);
    c->scan_waiting.store(0, std::memory_order_release);          // (a run whose files did not take the device scan after all)
    if (rc) { katgpu_table_free(t); return rc; }
    *out = t;
    return KATGPU_OK;
}

