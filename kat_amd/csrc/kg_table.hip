// kg_table.hip -- the HBM-resident count table behind katgpu_table: geometry and layout choice, allocation, regrow
// (hash_counter::double_size), statistics, batch lookups and per-position profiles, record export / merge.
#include "kg_host.hpp"
#include "kg_kernels.hpp"
#include "kg_wide.hpp"

static const uint32_t g_region_slots = (uint32_t)hook_u64("KATGPU_TEST_REGION_SLOTS", REGION_SLOTS);
static const bool g_no_packed = hook("KATGPU_NO_PACKED") != nullptr;   // tests / A-B: every table in the KV12 layout

// like_p1/like_p2 != 0: adopt that region grid (so that comp can join region against region) and take up the capacity in the
// region size, if a region of the resulting size still fits LDS.
static const bool g_no_lazy_zero = hook("KATGPU_NO_LAZY_ZERO") != nullptr;   // A/B: every table cleared when it is made
static const uint64_t g_lazy_min_slots = hook_u64("KATGPU_TEST_LAZY_MIN_SLOTS", 1ULL << 26);   // tests: packed tables from this size on leave their clearing to the first sweep
int alloc_dev_table(katgpu_ctx* c, uint32_t k, int canonical, uint64_t cap, DevTable* out, uint32_t like_p1, uint32_t like_p2, bool* lazy_zero) {
    DevTable d{};
    const uint64_t like_r = (uint64_t)like_p1 * like_p2;
    // slots per region: a wide slot is 20 bytes (two key words + the count), and the wide apply kernel holds a region in LDS like the
    // narrow ones do (kg_partition_wide.hpp): 6144 slots = 120 KB
    const uint32_t rs = k > 32 ? std::min<uint32_t>(g_region_slots, REGION_SLOTS_WIDE) : g_region_slots;
    // capacity is a whole number of regions (kg_device.hpp: Probe); a table smaller than one region is a single short region
    // (regions of fewer than 256 slots are not worth a common grid: the spread of the region loads would eat the table's fill limit)
    if (like_r > 1 && (cap + like_r - 1) / like_r <= AP2_MAX_SLOTS - 4 && (cap + like_r - 1) / like_r >= 256) {
        d.p1 = like_p1; d.p2 = like_p2; d.n_regions = (uint32_t)like_r;
        d.region_slots = (uint32_t)((std::max<uint64_t>((cap + like_r - 1) / like_r, 16) + 3) & ~3ULL);   // whole 16-byte lines of keys and counts per region
    } else if (cap <= rs) { d.n_regions = d.p1 = d.p2 = 1; d.region_slots = (uint32_t)((cap + 3) & ~3ULL); }
    else {
        const uint64_t nr = (cap + rs - 1) / rs;
        if (nr > 0x3FFFFFFFULL) return fail(c, KATGPU_ERR_NOMEM, "table of %llu slots exceeds the region index", (unsigned long long)cap);
        uint32_t p2 = 1;
        while ((uint64_t)p2 * p2 < nr) ++p2;                   // two radix digits of about the same size
        if (k <= 32) { uint32_t q = 1; while (q < p2) q <<= 1; p2 = q; }   // one-word tables: the level-2 digit is a bit field of the placement hash
        d.p2 = p2; d.p1 = (uint32_t)((nr + p2 - 1) / p2);
        d.region_slots = rs;
        // Level 2 of the partitioned counter works one bucket per workgroup and CU at a time: 584 buckets on 256 CUs are three
        // passes of which the last keeps 72 CUs busy.  With more buckets than CUs, make them a whole number of passes -- fewer,
        // larger regions if the apply kernel's LDS holds them (AP2_MAX_SLOTS), else more, smaller ones.
        const uint32_t ncu = (uint32_t)c->n_cu;
        if (k <= 32 && g_region_slots == REGION_SLOTS && ncu && d.p1 > ncu && d.p1 % ncu) {
            auto slots_for = [&](uint32_t p1) { return (uint32_t)(((cap + (uint64_t)p1 * p2 - 1) / ((uint64_t)p1 * p2) + 3) & ~3ULL); };
            const uint32_t lo = d.p1 / ncu * ncu, hi = lo + ncu;
            if (slots_for(lo) <= AP2_MAX_SLOTS) { d.p1 = lo; d.region_slots = slots_for(lo); }
            else if (hi <= MAX_PARTS) { d.p1 = hi; d.region_slots = slots_for(hi); }
        }
        d.n_regions = d.p1 * d.p2;
    }
    cap = (uint64_t)d.n_regions * d.region_slots;
    d.cap = cap; d.k = k; d.canonical = canonical ? 1 : 0;
    if (k <= 32) {                                             // the placement hash's bit budget (kg_device.hpp "placement")
        if (d.p2 & (d.p2 - 1)) return fail(c, KATGPU_ERR_INVALID_ARG, "a one-word table needs a power-of-two level-2 digit (got p2 = %u)", d.p2);
        while ((1u << d.l2) < d.p2) ++d.l2;
        d.n1 = place_n1(k, d.p1);
        // Packed slots (kg_device.hpp "P8") whenever the placement hash leaves the slot word at least PACK_MIN_CBITS count bits: the
        // remainder has n1 - l2 bits, so this is every table of >= 2^(2k - 44) regions -- 8 M slots at k = 27, 2 G at k = 31.
        const Place pl = place_make(k, d.p1, d.n1, d.l2);
        if (!g_no_packed && d.n_regions > 1 && pl.rb + PACK_MIN_CBITS <= 64) d.cbits = std::min<uint32_t>(64 - pl.rb, 32);
        d.inv_slots = 1.0 / (double)d.region_slots;
    }
    const double t0 = now_ms();
    const bool wide = k > 32;                                  // two key words per slot (kg_device.hpp "wide keys"), one block
    const size_t key_bytes = cap * sizeof(uint64_t) * (wide ? 2 : 1);
    HIPCHK(c, pool_alloc(c, (void**)&d.keys, key_bytes, /* take_reservation: a table made "like" another -- kat comp's later inputs */ like_p1 != 0 || like_p2 != 0));
    if (wide) d.keys_b = d.keys + cap;
    if (g_trace) fprintf(stderr, "[katgpu] alloc %s %.1f GB: %.1f ms\n", d.cbits ? "packed slots" : "keys", cap * 8 / 1e9, now_ms() - t0);
    hipError_t e = d.cbits ? hipSuccess : pool_alloc(c, (void**)&d.counts, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d.ovf_keys, OVF_CAP * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d.ovf_hi, OVF_CAP * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d.ctrs, CTR_WORDS * sizeof(uint64_t));
    if (e != hipSuccess) {
        pool_release(c, d.keys); pool_release(c, d.counts); hipFree(d.ovf_keys); hipFree(d.ovf_hi); hipFree(d.ctrs);
        return fail(c, KATGPU_ERR_NOMEM, "device allocation of a %llu-slot table failed: %s", (unsigned long long)cap, hipGetErrorString(e));
    }
    // a packed table the partitioned counter will take (>= 64 M slots): cleared by its first round's apply, region by region (katgpu_table::zero_from)
    const bool lazy = lazy_zero && !g_no_lazy_zero && d.cbits && !wide && cap >= g_lazy_min_slots;
    if (lazy_zero) *lazy_zero = lazy;
    if (!lazy) HIPCHK(c, hipMemsetAsync(d.keys, d.cbits ? 0 : 0xFF, key_bytes, c->stream));
    if (d.counts) HIPCHK(c, hipMemsetAsync(d.counts, 0, cap * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ovf_keys, 0xFF, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ovf_hi, 0, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ctrs, 0, CTR_WORDS * sizeof(uint64_t), c->stream));
    if (g_trace) { const double t1 = now_ms(); hipStreamSynchronize(c->stream); fprintf(stderr, "[katgpu +%.0f ms] table alloc: mallocs+enqueue %.1f ms, memsets done after %.1f ms more\n", since_load(), t1 - t0, now_ms() - t1); }
    *out = d;
    return KATGPU_OK;
}

void free_dev_table(katgpu_ctx* c, DevTable& d) {
    pool_release(c, d.keys); pool_release(c, d.counts);
    hipFree(d.ovf_keys); hipFree(d.ovf_hi); hipFree(d.ctrs);
    d = DevTable{};
}

extern "C" int katgpu_table_create(katgpu_ctx* c, uint32_t k, int canonical, uint64_t size_hint, int disable_grow, katgpu_table** out) {
    if (!c || !out) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (k < 1 || k > KATGPU_MAX_K) return fail(c, KATGPU_ERR_K, "k = %u unsupported: this build keeps a k-mer in at most two 63-bit words (1 <= k <= %d)", k, KATGPU_MAX_K);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t cap = std::max<uint64_t>(size_hint ? size_hint : (1u << 20), 1024);
    katgpu_table* t = new katgpu_table();
    t->ctx = c; t->disable_grow = disable_grow;
    bool lazy = false;
    int rc = alloc_dev_table(c, k, canonical, cap, &t->dv, 0, 0, &lazy);
    if (rc) { delete t; return rc; }
    if (lazy) t->zero_from = 0;
    *out = t;
    return KATGPU_OK;
}

extern "C" int katgpu_table_create_like(katgpu_ctx* c, const katgpu_table* like, uint32_t k, int canonical, uint64_t size_hint,
                                        int disable_grow, katgpu_table** out) {
    if (!c || !out || !like) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (k < 1 || k > KATGPU_MAX_K) return fail(c, KATGPU_ERR_K, "k = %u unsupported: this build keeps a k-mer in at most two 63-bit words (1 <= k <= %d)", k, KATGPU_MAX_K);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t cap = std::max<uint64_t>(size_hint ? size_hint : (1u << 20), 1024);
    katgpu_table* t = new katgpu_table();
    t->ctx = c; t->disable_grow = disable_grow;
    // (no common grid across key widths; none for wide tables either: nothing joins or merges them region by region, and a grid
    // handed down could make regions the wide apply kernel cannot hold)
    bool lazy = false;
    int rc = k > 32 || like->dv.k > 32 ? alloc_dev_table(c, k, canonical, cap, &t->dv, 0, 0, &lazy)
                                       : alloc_dev_table(c, k, canonical, cap, &t->dv, like->dv.p1, like->dv.p2, &lazy);
    if (rc) { delete t; return rc; }
    if (lazy) t->zero_from = 0;
    *out = t;
    return KATGPU_OK;
}

int table_wait(katgpu_table* t) {
    std::lock_guard<std::mutex> lk(t->alloc_mu);
    if (t->alloc_thread.joinable()) t->alloc_thread.join();
    if (t->alloc_rc) return fail(t->ctx, t->alloc_rc, "%s", t->alloc_err.c_str());
    return KATGPU_OK;
}

extern "C" void katgpu_table_free(katgpu_table* t) {
    if (!t) return;
    (void)table_wait(t);
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    t->zero_from = ~0ULL;                                  // (what was never cleared need not be now)
    free_dev_table(t->ctx, t->dv);
    delete t;
}

extern "C" uint32_t katgpu_table_k(const katgpu_table* t) { return t ? t->dv.k : 0; }
extern "C" uint32_t katgpu_table_regrows(const katgpu_table* t) { return t ? t->n_regrows : 0; }
extern "C" uint32_t katgpu_table_slot_bytes(const katgpu_table* t) { return !t ? 0 : t->dv.keys_b ? 20 : t->dv.cbits ? 8 : 12; }
extern "C" int katgpu_table_canonical(const katgpu_table* t) { return t ? (int)t->dv.canonical : 0; }

// read the counter block back (one small D2H; synchronises the compute stream)
// may this table's slots be left uncleared for a first sweep to clear (katgpu_table::zero_from)?  What alloc_dev_table asks of a new table.
bool table_may_stay_uncleared(const DevTable& d) { return !g_no_lazy_zero && d.cbits && !d.keys_b && d.n_regions > 1 && d.cap >= g_lazy_min_slots; }

int refresh_counters(katgpu_table* t) {
    katgpu_ctx* c = t->ctx;
    if (t->zero_failed) return fail(c, KATGPU_ERR_DEVICE, "the table's unswept slots could not be cleared (hipMemsetAsync failed): its contents are not to be trusted");
    uint64_t h[CTR_WORDS];
    HIPCHK(c, hipMemcpyAsync(h, t->dv.ctrs, sizeof h, hipMemcpyDeviceToHost, c->stream));      // (the counters only: a table whose slots wait for their first sweep stays that way)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t d = 0;
    for (int i = 0; i < CTR_NSTRIPES; ++i) d += h[CTR_DISTINCT0 + i];
    t->ones = h[CTR_ONES];
    t->distinct = d + (t->ones ? 1 : 0);
    t->n_ovf = (uint32_t)h[CTR_OVF_USED];
    if (h[CTR_FULL]) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
    return KATGPU_OK;
}

// hash_counter::double_size (deps/jellyfish-2.2.0/include/jellyfish/hash_counter.hpp:204-244): allocate a larger array,
// re-insert every (key,count), swap.  Here one grid-stride kernel instead of a barrier-synchronised thread team.
int regrow(katgpu_table* t, uint64_t new_cap) {
    katgpu_ctx* c = t->ctx;
    DevTable nd{};
    int rc = alloc_dev_table(c, t->dev().k, t->dev().canonical, new_cap, &nd, t->dev().n_regions > 1 ? t->dev().p1 : 0, t->dev().n_regions > 1 ? t->dev().p2 : 0);
    if (rc) return rc;
    {
        ScopedTimer tm(c, KATGPU_K_REGROW, t->dev().cap);
        if (t->dev().keys_b) hipLaunchKernelGGL(k_regrow_w, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, nd, t->dev(), t->n_ovf);
        else hipLaunchKernelGGL(k_regrow, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, nd, t->dev(), t->n_ovf);
    }
    HIPCHK(c, hipMemcpyAsync(&nd.ctrs[CTR_ONES], &t->dev().ctrs[CTR_ONES], sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_dev_table(c, t->dev());
    t->dev() = nd;
    ++t->n_regrows;
    t->count_bound = 0xFFFFFFFFULL;           // full counts were folded back into the slots: the next unchecked launch sweeps first
    t->unchecked_adds = 0;
    return refresh_counters(t);
}

// The fill limit of the direct path.  A k-mer probes inside its region only, so it is the fullest REGION that must not run out
// of slots: regions get Binomial(n, 1/R) k-mers, and small regions (a table created "like" a much bigger one) need more slack
// than the 0.7 that suits regions of thousands of slots.
double load_limit(const DevTable& d) {
    if (d.region_slots >= 1024 || d.n_regions == 1) return 0.7;
    return std::max(0.25, 0.7 - 3.0 / std::sqrt((double)d.region_slots));
}

// Make room for up to `incoming` new distinct k-mers (an upper bound: one per window start) at load <= the fill limit.
int ensure_room(katgpu_table* t, uint64_t incoming) {
    int rc = refresh_counters(t);
    if (rc) return rc;
    const uint64_t need = t->distinct + incoming;
    if ((double)need <= load_limit(t->dev()) * (double)t->dev().cap) return KATGPU_OK;
    if (t->disable_grow) return fail(t->ctx, KATGPU_ERR_TABLE_FULL, "Hash full");
    uint64_t new_cap = t->dev().cap;
    while ((double)need > 0.5 * (double)new_cap) new_cap *= 2;
    return regrow(t, new_cap);
}

extern "C" int katgpu_table_stats(katgpu_table* t, uint64_t* distinct, uint64_t* total, uint64_t* capacity) {
    if (!t) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    if (distinct) *distinct = t->distinct;
    if (capacity) *capacity = t->dev().cap;
    if (total) {
        uint64_t* scratch = &t->dev().ctrs[CTR_SCRATCH];
        HIPCHK(c, hipMemsetAsync(scratch, 0, sizeof(uint64_t), c->stream));
        hipLaunchKernelGGL(k_total, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, scratch);
        uint64_t s = 0;
        HIPCHK(c, hipMemcpyAsync(&s, scratch, sizeof s, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *total = s + t->ones;
    }
    return KATGPU_OK;
}

extern "C" int katgpu_table_get(katgpu_table* t, const uint64_t* keys, size_t n, int canonicalise, uint64_t* counts) {
    if (!t || (n && (!keys || !counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_get: use katgpu_table_get_wide;");
    if (!n) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    uint64_t *dk = nullptr, *dc = nullptr;
    HIPCHK(c, hipMalloc(&dk, n * 8));
    if (hipMalloc(&dc, n * 8) != hipSuccess) { hipFree(dk); return fail(c, KATGPU_ERR_NOMEM, "lookup buffers"); }
    hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, dk, (uint64_t)n, canonicalise, dc);
    hipMemcpyAsync(counts, dc, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(dk); hipFree(dc);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

static int launch_profile(katgpu_table* t, const uint8_t* dev_bases, size_t n, int canonicalise, uint64_t* dev_counts) {
    katgpu_ctx* c = t->ctx;
    const bool wide = t->dev().keys_b != nullptr;
    const uint64_t n_out = n - t->dev().k + 1;
    const uint64_t per_chunk = wide ? WIDE_CHUNK_STARTS : CHUNK_STARTS;
    const uint64_t n_chunks = (n_out + per_chunk - 1) / per_chunk;
    const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * 8);
    const bool aligned = (reinterpret_cast<uintptr_t>(dev_bases) & 15) == 0 && (reinterpret_cast<uintptr_t>(dev_counts) & 15) == 0;
    ScopedTimer tm(c, KATGPU_K_PROFILE, n_out);
    if (wide && aligned)
        hipLaunchKernelGGL(k_profile_w<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else if (wide)
        hipLaunchKernelGGL(k_profile_w<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else if (aligned)
        hipLaunchKernelGGL(k_profile<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else
        hipLaunchKernelGGL(k_profile<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}

extern "C" int katgpu_table_profile_device(katgpu_table* t, const uint8_t* dev_bases, size_t n, int canonicalise, uint64_t* dev_counts) {
    if (!t || (n && (!dev_bases || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    if (n < t->dev().k) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    return launch_profile(t, dev_bases, n, canonicalise, dev_counts);
}

// Host form: the sequence goes through the device in batches of PROFILE_BATCH window starts (each batch re-sends the
// k-1 bases it shares with the next one), so any length fits next to the table.
extern "C" int katgpu_table_profile_host(katgpu_table* t, const char* bases, size_t n, int canonicalise, uint64_t* counts) {
    if (!t || (n && (!bases || !counts))) return KATGPU_ERR_INVALID_ARG;
    const uint32_t k = t->dev().k;
    if (n < k) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    const size_t n_out = n - k + 1;
    const size_t PROFILE_BATCH = (size_t)32 << 20;
    const size_t batch = std::min(n_out, PROFILE_BATCH);
    uint8_t* db = nullptr; uint64_t* dc = nullptr;
    HIPCHK(c, pool_alloc(c, (void**)&db, batch + 64));
    if (pool_alloc(c, (void**)&dc, batch * 8) != hipSuccess) { pool_release(c, db); return fail(c, KATGPU_ERR_NOMEM, "profile buffers"); }
    hipError_t e = hipSuccess;
    for (size_t pos = 0; pos < n_out && rc == KATGPU_OK && e == hipSuccess; pos += batch) {
        const size_t starts = std::min(batch, n_out - pos);
        const size_t nb = starts + k - 1;
        e = hipMemcpyAsync(db, bases + pos, nb, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) break;
        rc = launch_profile(t, db, nb, canonicalise, dc);
        if (rc) break;
        e = hipMemcpyAsync(counts + pos, dc, starts * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    hipStreamSynchronize(c->stream);
    pool_release(c, db); pool_release(c, dc);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

// ------------------------------------------------------------------ partition / export / merge -------

extern "C" int katgpu_table_partition_sizes(katgpu_table* t, uint32_t n_parts, uint64_t* sizes) {
    if (!t || !sizes || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemsetAsync(d, 0, n_parts * 8, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->dev().cap);
        if (t->dev().keys_b)
            hipLaunchKernelGGL(k_partition_w<0>, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, n_parts, d, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr);
        else
            hipLaunchKernelGGL(k_partition<0>, dim3(grid_for(c, t->dev().cap + 1, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, n_parts, d, (uint64_t*)nullptr, (uint64_t*)nullptr);
    }
    hipMemcpyAsync(sizes, d, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_partition(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets, uint64_t* dev_keys, uint64_t* dev_counts) {
    if (!t || !offsets || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_partition");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemcpyAsync(d, offsets, n_parts * 8, hipMemcpyHostToDevice, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->dev().cap);
        hipLaunchKernelGGL(k_partition<1>, dim3(grid_for(c, t->dev().cap + 1, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, n_parts, d, dev_keys, dev_counts);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_export(katgpu_table* t, uint64_t* keys, uint64_t* counts, size_t cap, size_t* n_out) {
    if (!t || !n_out) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_export: use katgpu_table_export_wide;");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    *n_out = (size_t)t->distinct;
    if (cap == 0) return KATGPU_OK;
    if (cap < t->distinct || !keys || !counts) return fail(c, KATGPU_ERR_INVALID_ARG, "export buffer too small: %zu < %llu", cap, (unsigned long long)t->distinct);
    if (!t->distinct) return KATGPU_OK;
    uint64_t *dk = nullptr, *dc = nullptr;
    HIPCHK(c, hipMalloc(&dk, t->distinct * 8));
    if (hipMalloc(&dc, t->distinct * 8) != hipSuccess) { hipFree(dk); return fail(c, KATGPU_ERR_NOMEM, "export buffers"); }
    uint64_t zero = 0;
    rc = katgpu_table_partition(t, 1, &zero, dk, dc);
    if (!rc) {
        hipError_t e = hipMemcpy(keys, dk, t->distinct * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(counts, dc, t->distinct * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "export: %s", hipGetErrorString(e));
    }
    hipFree(dk); hipFree(dc);
    return rc;
}

extern "C" int katgpu_table_merge_device(katgpu_table* t, const uint64_t* dev_keys, const uint64_t* dev_counts, size_t n) {
    if (!t || (n && (!dev_keys || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_merge_device");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    size_t pos = 0;
    while (pos < n) {
        int rc = refresh_counters(t); if (rc) return rc;
        uint64_t room = (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) > t->distinct ? (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) - t->distinct : 0;
        uint64_t want = n - pos;
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 8, 1024))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 2, 1024)));
            if (rc) return rc;
            continue;
        }
        uint64_t take = std::min(want, room);
        t->count_bound = 0xFFFFFFFFULL;          // merged amounts are arbitrary: the next k_count launch sweeps first
        {
            ScopedTimer tm(c, KATGPU_K_MERGE, take);
            hipLaunchKernelGGL(k_merge, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->dev(), dev_keys + pos, dev_counts + pos, take);
        }
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_host(katgpu_table* t, const uint64_t* keys, const uint64_t* counts, size_t n) {
    if (!t || (n && (!keys || !counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_merge_host: use katgpu_table_merge_host_wide;");
    if (!n) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t *dk = nullptr, *dc = nullptr;
    HIPCHK(c, hipMalloc(&dk, n * 8));
    if (hipMalloc(&dc, n * 8) != hipSuccess) { hipFree(dk); return fail(c, KATGPU_ERR_NOMEM, "merge buffers"); }
    hipError_t e = hipMemcpy(dk, keys, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc, counts, n * 8, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? katgpu_table_merge_device(t, dk, dc, n) : fail(c, KATGPU_ERR_DEVICE, "merge: %s", hipGetErrorString(e));
    hipFree(dk); hipFree(dc);
    return rc;
}

// ------------------------------------------------------------------ wide tables (33 <= k <= 63): records in and out ----

extern "C" int katgpu_table_export_wide(katgpu_table* t, uint64_t* keys_hi, uint64_t* keys_lo, uint64_t* counts, size_t cap, size_t* n_out) {
    if (!t || !n_out) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->dev().keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_export_wide is for k > 32 tables (k = %u): use katgpu_table_export", t->dev().k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    *n_out = (size_t)t->distinct;
    if (cap == 0) return KATGPU_OK;
    if (cap < t->distinct || !keys_hi || !keys_lo || !counts) return fail(c, KATGPU_ERR_INVALID_ARG, "export buffer too small: %zu < %llu", cap, (unsigned long long)t->distinct);
    if (!t->distinct) return KATGPU_OK;
    const size_t n = (size_t)t->distinct;
    uint64_t* d = nullptr;
    HIPCHK(c, hipMalloc(&d, (3 * n + 1) * 8));
    unsigned long long* cursor = (unsigned long long*)(d + 3 * n);
    hipMemsetAsync(cursor, 0, 8, c->stream);
    hipLaunchKernelGGL(k_export_w, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, d, d + n, d + 2 * n, cursor);
    hipMemcpyAsync(keys_hi, d, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(keys_lo, d + n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(counts, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_partition_wide(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets, uint64_t* dev_hi, uint64_t* dev_lo, uint64_t* dev_counts) {
    if (!t || !offsets || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->dev().keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_partition_wide is for k > 32 tables (k = %u): use katgpu_table_partition", t->dev().k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    if (t->distinct && (!dev_hi || !dev_lo || !dev_counts)) return KATGPU_ERR_INVALID_ARG;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemcpyAsync(d, offsets, n_parts * 8, hipMemcpyHostToDevice, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->dev().cap);
        hipLaunchKernelGGL(k_partition_w<1>, dim3(grid_for(c, t->dev().cap, 256, 8)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, n_parts, d, dev_hi, dev_lo, dev_counts);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_merge_device_wide(katgpu_table* t, const uint64_t* dev_hi, const uint64_t* dev_lo, const uint64_t* dev_counts, size_t n) {
    if (!t || (n && (!dev_hi || !dev_lo || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->dev().keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_merge_device_wide is for k > 32 tables (k = %u): use katgpu_table_merge_device", t->dev().k);
    HIPCHK(c, hipSetDevice(c->device));
    size_t pos = 0;
    while (pos < n) {
        int rc = refresh_counters(t); if (rc) return rc;
        const uint64_t room = (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) > t->distinct ? (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap) - t->distinct : 0;
        const uint64_t want = n - pos;
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 8, 1024))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->dev().cap / 2, 1024)));
            if (rc) return rc;
            continue;
        }
        const uint64_t take = std::min(want, room);
        ScopedTimer tm(c, KATGPU_K_MERGE, take);
        hipLaunchKernelGGL(k_merge_w, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->dev(), dev_hi + pos, dev_lo + pos, dev_counts + pos, (uint64_t)take);
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_host_wide(katgpu_table* t, const uint64_t* keys_hi, const uint64_t* keys_lo, const uint64_t* counts, size_t n) {
    if (!t || (n && (!keys_hi || !keys_lo || !counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->dev().keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_merge_host_wide is for k > 32 tables (k = %u): use katgpu_table_merge_host", t->dev().k);
    if (!n) return KATGPU_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t k = t->dev().k;
    const uint64_t hi_mask = (1ULL << (2 * k - 64)) - 1;           // 2 <= 2k - 64 <= 62
    for (size_t i = 0; i < n; ++i)
        if (keys_hi[i] & ~hi_mask) return fail(c, KATGPU_ERR_INVALID_ARG, "record %zu: key wider than 2k = %u bits", i, 2 * k);
    uint64_t* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 3 * n * 8));
    hipError_t e = hipMemcpy(d, keys_hi, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, keys_lo, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * n, counts, n * 8, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? katgpu_table_merge_device_wide(t, d, d + n, d + 2 * n, n) : fail(c, KATGPU_ERR_DEVICE, "merge: %s", hipGetErrorString(e));
    hipStreamSynchronize(c->stream);
    hipFree(d);
    return rc;
}

extern "C" int katgpu_table_get_wide(katgpu_table* t, const uint64_t* keys_hi, const uint64_t* keys_lo, size_t n, int canonicalise, uint64_t* counts) {
    if (!t || (n && (!keys_hi || !keys_lo || !counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->dev().keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_get_wide is for k > 32 tables (k = %u): use katgpu_table_get", t->dev().k);
    if (!n) return KATGPU_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    uint64_t* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 3 * n * 8));
    hipMemcpyAsync(d, keys_hi, n * 8, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d + n, keys_lo, n * 8, hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_get_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t->dev(), t->n_ovf, d, d + n, (uint64_t)n, canonicalise, d + 2 * n);
    hipMemcpyAsync(counts, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

