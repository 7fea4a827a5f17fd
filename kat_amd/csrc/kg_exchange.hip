// kg_exchange.hip -- the device side of the multi-GPU exchange: region-ordered extraction of a table's records by owner, the
// in-place merge of received runs region by region in LDS, and the direct paths beside them (kg_comm.hip drives these over RCCL;
// kat_amd/dist.py drives them over torch.distributed).
#include "kg_host.hpp"
#include "kg_kernels.hpp"

// ------------------------------------------------------------------ region-ordered exchange -----------

extern "C" int katgpu_place_keys(uint32_t k, uint32_t p1, uint32_t l2, const uint64_t* keys, size_t n, uint32_t* d1, uint32_t* d2, uint64_t* rem,
                                 uint64_t* back, uint32_t* rem_bits, uint32_t region_slots, uint32_t* offset) {
    if (k < 1 || k > 32 || p1 < 1 || p1 > MAX_PARTS || l2 > 10 || (n && (!keys || !d1 || !d2 || !rem || !back))) return KATGPU_ERR_INVALID_ARG;
    const Place pl = place_make(k, p1, place_n1(k, p1), l2);
    if (rem_bits) *rem_bits = pl.rb;
    for (size_t i = 0; i < n; ++i) {
        const Placed h = place_hash(keys[i], pl);
        d1[i] = h.d1; d2[i] = h.d2; rem[i] = h.rem;
        back[i] = place_key(place_base1(h.d1, pl), (pl.rb < 64 ? (uint64_t)h.d2 << pl.rb : 0ULL) | h.rem, pl);
        if (offset && region_slots) offset[i] = place_offset(h.rem, pl, region_slots);
    }
    return KATGPU_OK;
}

extern "C" int katgpu_table_geometry(const katgpu_table* t, katgpu_geometry* g) {
    if (!t || !g) return KATGPU_ERR_INVALID_ARG;
    if (t->dv.keys_b) return fail(t->ctx, KATGPU_ERR_K, "the multi-GPU exchange is not available for k > 32 (k = %u)", t->dv.k);
    g->k = t->dv.k; g->canonical = t->dv.canonical; g->n_regions = t->dv.n_regions; g->region_slots = t->dv.region_slots;
    g->p1 = t->dv.p1; g->p2 = t->dv.p2; g->capacity = t->dv.cap;
    return KATGPU_OK;
}

extern "C" int katgpu_table_extract_sizes(katgpu_table* t, uint32_t n_parts, uint32_t* dev_region_counts, uint64_t* part_sizes) {
    if (!t || !dev_region_counts || !part_sizes || n_parts == 0 || n_parts > MAX_EXCHANGE_PARTS) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d_tot = nullptr;
    HIPCHK(c, hipMalloc(&d_tot, n_parts * 8));
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->dev().cap);
        hipLaunchKernelGGL(k_extract_count, dim3(std::min<uint32_t>(t->dev().n_regions, (uint32_t)c->n_cu * 8)), dim3(EXTRACT_BLOCK), 0, c->stream, t->dev(), n_parts, dev_region_counts);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, (const uint32_t*)dev_region_counts, t->dev().n_regions, (uint64_t)t->dev().n_regions,
                           (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, 0, d_tot);
    }
    hipMemcpyAsync(part_sizes, d_tot, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d_tot);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_packed_records(const katgpu_table* t) {
    if (!t || t->dv.keys_b) return 0;
    return t->dv.cbits != 0;                                      // (a packed slot's remainder has at most 64 - PACK_MIN_CBITS = 44 bits: rec_xs <= 4, counts to 2^28 in the record)
}

static int extract_records(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint64_t* dev_keys, uint32_t* dev_rem_lo, uint8_t* dev_rem_hi, uint32_t* dev_counts,
                           uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big);

extern "C" int katgpu_table_extract(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint64_t* dev_keys, uint32_t* dev_counts,
                                    uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big) {
    if (!dev_keys) return KATGPU_ERR_INVALID_ARG;
    return extract_records(t, n_parts, dev_region_counts, dev_keys, nullptr, nullptr, dev_counts, big_keys, big_counts, big_cap, n_big);
}

extern "C" int katgpu_table_extract_packed(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint32_t* dev_rem_lo, uint8_t* dev_rem_hi, uint32_t* dev_counts,
                                           uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big) {
    if (!t || !dev_rem_lo || !dev_rem_hi) return KATGPU_ERR_INVALID_ARG;
    if (!katgpu_table_packed_records(t)) return fail(t->ctx, KATGPU_ERR_INVALID_ARG, "katgpu_table_extract_packed: not a packed table");
    return extract_records(t, n_parts, dev_region_counts, nullptr, dev_rem_lo, dev_rem_hi, dev_counts, big_keys, big_counts, big_cap, n_big);
}

static int extract_records(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint64_t* dev_keys, uint32_t* dev_rem_lo, uint8_t* dev_rem_hi, uint32_t* dev_counts,
                           uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big) {
    if (!t || !dev_region_counts || !dev_counts || !n_big || n_parts == 0 || n_parts > MAX_EXCHANGE_PARTS || (big_cap && (!big_keys || !big_counts)))
        return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    const uint32_t R = t->dev().n_regions;
    const uint32_t dev_big_cap = OVF_CAP + 8;
    uint8_t* tmp = nullptr;          // [off u64 n_parts*R | totals n_parts | base n_parts | big_n | big keys | big counts]
    const size_t off_bytes = (size_t)n_parts * R * 8;
    const size_t bytes = off_bytes + (size_t)n_parts * 16 + 8 + (size_t)dev_big_cap * 16;
    HIPCHK(c, hipMalloc((void**)&tmp, bytes));
    uint64_t* d_off = (uint64_t*)tmp;
    unsigned long long* d_tot = (unsigned long long*)(tmp + off_bytes);
    uint64_t* d_base = (uint64_t*)(d_tot + n_parts);
    unsigned long long* d_bign = (unsigned long long*)(d_base + n_parts);
    uint64_t* d_bk = (uint64_t*)(d_bign + 1);
    uint64_t* d_bc = d_bk + dev_big_cap;
    std::vector<uint64_t> tot(n_parts), base(n_parts);
    hipError_t e = hipSuccess;
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->dev().cap);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, dev_region_counts, R, (uint64_t)R, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, 0, d_tot);
        hipMemcpyAsync(tot.data(), d_tot, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
        e = hipStreamSynchronize(c->stream);
        uint64_t run = 0;
        for (uint32_t p = 0; p < n_parts; ++p) { base[p] = run; run += tot[p]; }
        hipMemcpyAsync(d_base, base.data(), n_parts * 8, hipMemcpyHostToDevice, c->stream);
        hipMemsetAsync(d_bign, 0, 8, c->stream);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, dev_region_counts, R, (uint64_t)R, (const uint64_t*)d_base, d_off, (uint64_t)R, 0, (unsigned long long*)nullptr);
        if (dev_keys) hipLaunchKernelGGL(k_extract_write<false>, dim3(std::min<uint32_t>(R, (uint32_t)c->n_cu * 8)), dim3(EXTRACT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, n_parts, (const uint64_t*)d_off,
                                         dev_keys, (uint32_t*)nullptr, (uint8_t*)nullptr, dev_counts, d_bk, d_bc, d_bign, dev_big_cap);
        else hipLaunchKernelGGL(k_extract_write<true>, dim3(std::min<uint32_t>(R, (uint32_t)c->n_cu * 8)), dim3(EXTRACT_BLOCK), 0, c->stream, t->dev(), t->n_ovf, n_parts, (const uint64_t*)d_off,
                                (uint64_t*)nullptr, dev_rem_lo, dev_rem_hi, dev_counts, d_bk, d_bc, d_bign, dev_big_cap);
    }
    unsigned long long nb = 0;
    hipMemcpyAsync(&nb, d_bign, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    uint32_t out = 0;
    rc = KATGPU_OK;
    if (e == hipSuccess) {
        if (nb > dev_big_cap) rc = fail(c, KATGPU_ERR_DEVICE, "extract: %llu counts above 32 bits", nb);
        else {
            std::vector<uint64_t> hk(nb), hc(nb);
            if (nb) { e = hipMemcpy(hk.data(), d_bk, nb * 8, hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemcpy(hc.data(), d_bc, nb * 8, hipMemcpyDeviceToHost); }
            if (t->ones) { hk.push_back(~0ULL); hc.push_back(t->ones); }          // the all-ones key has no slot (kg_device.hpp)
            if (hk.size() > big_cap) rc = fail(c, KATGPU_ERR_INVALID_ARG, "extract: big list needs %zu entries", hk.size());
            else { for (size_t i = 0; i < hk.size(); ++i) { big_keys[i] = hk[i]; big_counts[i] = hc[i]; } out = (uint32_t)hk.size(); }
        }
    }
    hipFree(tmp);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    *n_big = out;
    return rc;
}

extern "C" int katgpu_table_clear(katgpu_table* t) {
    if (!t) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    // A packed table of size is not cleared here: the region-ordered merge that refills it (katgpu_table_merge_regions*: the exchange)
    // starts every region from zeros in LDS and writes every one back -- neither the memset nor the merge's read of it; whoever else
    // touches the table first clears what has not been swept (katgpu_table::zero_from, as for a new table).
    DevTable& d = t->dv;
    if (table_may_stay_uncleared(d)) t->zero_from = 0;
    else {
        t->zero_from = ~0ULL;
        HIPCHK(c, hipMemsetAsync(d.keys, d.cbits ? 0 : 0xFF, d.cap * sizeof(uint64_t) * (d.keys_b ? 2 : 1), c->stream));     // (wide: keys_b follows keys)
    }
    if (d.counts) HIPCHK(c, hipMemsetAsync(d.counts, 0, d.cap * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ovf_keys, 0xFF, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ovf_hi, 0, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ctrs, 0, CTR_WORDS * sizeof(uint64_t), c->stream));
    t->n_ovf = 0; t->distinct = 0; t->ones = 0; t->count_bound = 0; t->unchecked_adds = 0; t->carry_n = 0;
    return KATGPU_OK;
}

static int merge_direct32(katgpu_table* t, const uint64_t* dev_keys, const uint32_t* dev_counts, size_t n) {
    katgpu_ctx* c = t->ctx;
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
        const uint64_t take = std::min(want, room);
        t->count_bound = 0xFFFFFFFFULL;
        ScopedTimer tm(c, KATGPU_K_MERGE, take);
        hipLaunchKernelGGL(k_merge32, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->dev(), dev_keys + pos, dev_counts + pos, (uint64_t)take);
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_device32(katgpu_table* t, const uint64_t* dev_keys, const uint32_t* dev_counts, size_t n) {
    if (!t || (n && (!dev_keys || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    return merge_direct32(t, dev_keys, dev_counts, n);
}

static const bool g_no_merge_apply = hook("KATGPU_NO_MERGE_APPLY") != nullptr;    // A/B switch + tests: every source through the direct path

// a source of either form: keys + counts, or packed records (rem_lo + rem_hi + counts: keys == null; their region counts are what orders them)
struct GSrc { const uint64_t* dev_keys; const uint32_t* dev_rem_lo; const uint8_t* dev_rem_hi; const uint32_t* dev_counts; const uint32_t* dev_region_counts; uint64_t n_records; uint32_t p1, p2; };
static int merge_regions_impl(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const GSrc* src);

extern "C" int katgpu_table_merge_regions(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const katgpu_merge_source* src) {
    if (!t || !src || n_src == 0 || g_lo > g_hi) return KATGPU_ERR_INVALID_ARG;
    std::vector<GSrc> g(n_src);
    for (uint32_t i = 0; i < n_src; ++i) {
        if (src[i].n_records && (!src[i].dev_keys || !src[i].dev_counts)) return KATGPU_ERR_INVALID_ARG;
        g[i] = GSrc{src[i].dev_keys, nullptr, nullptr, src[i].dev_counts, src[i].dev_region_counts, src[i].n_records, src[i].p1, src[i].p2};
    }
    return merge_regions_impl(t, g_lo, g_hi, n_src, g.data());
}

extern "C" int katgpu_table_merge_regions_packed(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const katgpu_merge_source_packed* src) {
    if (!t || !src || n_src == 0 || g_lo > g_hi) return KATGPU_ERR_INVALID_ARG;
    if (!t->dv.cbits || t->dv.keys_b) return fail(t->ctx, KATGPU_ERR_INVALID_ARG, "katgpu_table_merge_regions_packed: not a packed table");
    std::vector<GSrc> g(n_src);
    for (uint32_t i = 0; i < n_src; ++i) {
        if (src[i].n_records && (!src[i].dev_rem_lo || !src[i].dev_rem_hi || !src[i].dev_counts || !src[i].dev_region_counts || !src[i].p1 || !src[i].p2 || (src[i].p2 & (src[i].p2 - 1))))
            return KATGPU_ERR_INVALID_ARG;                    // (a packed record is nothing without its region: the counts and the grid are not optional)
        g[i] = GSrc{nullptr, src[i].dev_rem_lo, src[i].dev_rem_hi, src[i].dev_counts, src[i].dev_region_counts, src[i].n_records, src[i].p1, src[i].p2};
    }
    return merge_regions_impl(t, g_lo, g_hi, n_src, g.data());
}

static int merge_regions_impl(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const GSrc* src) {
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    if (!c->merge_attr_set) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_apply<MERGE_BLOCK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_apply<MERGE_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        c->merge_attr_set = true;
    }
    const size_t slot_bytes = t->dv.cbits ? 8 : 12;
    // sources ordered by this table's regions go through LDS; the rest (another grid, or the table has changed its grid) directly
    std::vector<uint32_t> aligned, direct;
    for (uint32_t i = 0; i < n_src; ++i) {
        if (src[i].n_records == 0) continue;
        const bool ok = !g_no_merge_apply && src[i].dev_region_counts && src[i].p1 == t->dv.p1 && src[i].p2 == t->dv.p2 && g_hi <= t->dv.n_regions &&
                        (size_t)t->dv.region_slots * slot_bytes <= 150 * 1024;
        (ok ? aligned : direct).push_back(i);
    }
    const uint32_t n_reg = g_hi - g_lo;
    for (size_t a0 = 0; a0 < aligned.size() && n_reg; a0 += MAX_MERGE_SRC) {
        const uint32_t na = (uint32_t)std::min<size_t>(MAX_MERGE_SRC, aligned.size() - a0);
        uint8_t* tmp = nullptr;        // [off: na * (n_reg + 1) u64 | deferred: n_reg u32 | n_deferred]
        const size_t off_bytes = (size_t)na * (n_reg + 1) * 8;
        HIPCHK(c, hipMalloc((void**)&tmp, off_bytes + (size_t)n_reg * 4 + 8));
        uint64_t* d_off = (uint64_t*)tmp;
        uint32_t* d_def = (uint32_t*)(tmp + off_bytes);
        unsigned long long* d_ndef = (unsigned long long*)(tmp + off_bytes + (size_t)n_reg * 4);
        MergeSrcs ms{};
        ms.n = na;
        ms.src_p1 = src[aligned[a0]].p1; ms.src_n1 = place_n1(t->dv.k, ms.src_p1); ms.src_l2 = (uint32_t)__builtin_ctz(src[aligned[a0]].p2);   // (aligned sources share one grid: this table's)
        uint64_t records = 0;
        for (uint32_t q = 0; q < na; ++q) {
            const GSrc& s = src[aligned[a0 + q]];
            hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, c->stream, s.dev_region_counts, n_reg, (uint64_t)n_reg, (const uint64_t*)nullptr,
                               d_off + (size_t)q * (n_reg + 1), (uint64_t)(n_reg + 1), 1, (unsigned long long*)nullptr);
            ms.s[q] = MergeSrc{s.dev_keys, s.dev_counts, d_off + (size_t)q * (n_reg + 1), s.dev_rem_lo, s.dev_rem_hi};
            records += s.n_records;
        }
        hipMemsetAsync(d_ndef, 0, 8, c->stream);
        t->count_bound = 0xFFFFFFFFULL;
        // A table whose slots wait for their first sweep (katgpu_table_clear left them): this launch is that sweep for [g_lo, g_hi) when those
        // are the next regions in line and every aligned source is in it -- each region starts from zeros in LDS and each is written back;
        // else what is left is cleared now.
        uint32_t zero_fill = 0;
        if (t->zero_from != ~0ULL) {
            if (t->dv.cbits && t->zero_from == g_lo && a0 == 0 && na == aligned.size()) zero_fill = 1;
            else { if (g_trace) fprintf(stderr, "[katgpu] merge: regions from %llu on cleared now (this merge begins at %u)\n", (unsigned long long)t->zero_from, g_lo); t->zero_rest(); }
        }
        if (g_trace && zero_fill) fprintf(stderr, "[katgpu] merge: regions [%u, %u) start from zeros in LDS (the table was left uncleared)\n", g_lo, g_hi);
        {
            ScopedTimer tm(c, KATGPU_K_MERGE, records);
            const size_t lds = (size_t)t->dv.region_slots * slot_bytes;
            const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds + 512), 2));
            const dim3 grid(std::min<uint32_t>(n_reg, (uint32_t)c->n_cu * per_cu));
            if (t->dv.cbits) hipLaunchKernelGGL((k_merge_apply<MERGE_BLOCK, true>), grid, dim3(MERGE_BLOCK), lds, c->stream, t->dv, g_lo, g_hi, ms, d_def, d_ndef, zero_fill);
            else hipLaunchKernelGGL((k_merge_apply<MERGE_BLOCK, false>), grid, dim3(MERGE_BLOCK), lds, c->stream, t->dv, g_lo, g_hi, ms, d_def, d_ndef, 0u);
        }
        if (zero_fill) t->zero_from = g_hi >= t->dv.n_regions ? ~0ULL : g_hi;       // (regions [g_lo, g_hi) have had their sweep)
        unsigned long long ndef = 0;
        hipMemcpyAsync(&ndef, d_ndef, 8, hipMemcpyDeviceToHost, c->stream);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipFree(tmp); return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e)); }
        if (ndef) {                                   // regions that could have overflowed: their runs go in directly (with growth)
            std::vector<uint32_t> regs(ndef);
            std::vector<uint64_t> off((size_t)na * (n_reg + 1));
            e = hipMemcpy(regs.data(), d_def, ndef * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(off.data(), d_off, off.size() * 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { hipFree(tmp); return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e)); }
            if (g_trace) fprintf(stderr, "[katgpu] merge: %llu region(s) deferred to the direct path\n", ndef);
            // Room first, per REGION: the runs of one region all land in that region, so the global load says nothing here.
            // After growing every region to hold what it has plus what arrives (at load 0.7) no insert below can fail.
            uint64_t max_in = 0;
            for (uint32_t g : regs) {
                uint64_t in = 0;
                for (uint32_t q = 0; q < na; ++q) { const uint64_t* o = off.data() + (size_t)q * (n_reg + 1); in += o[g - g_lo + 1] - o[g - g_lo]; }
                max_in = std::max(max_in, in);
            }
            const uint64_t need_s = (uint64_t)(((double)t->dv.region_slots + (double)max_in) / 0.7) + 1;
            if (need_s > t->dv.region_slots) {
                if (t->disable_grow) rc = fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
                else rc = regrow(t, (uint64_t)t->dv.n_regions * need_s);
            }
            if (rc == KATGPU_OK) {                    // one launch for all deferred regions: every region now has the room
                ScopedTimer tm(c, KATGPU_K_MERGE, max_in * ndef);
                hipLaunchKernelGGL(k_merge_deferred, dim3((unsigned)std::min<unsigned long long>(ndef, (unsigned long long)c->n_cu * 8)), dim3(256), 0, c->stream,
                                   t->dv, g_lo, ms, (const uint32_t*)d_def, (uint32_t)ndef);
                if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "deferred merge");
                else rc = refresh_counters(t);
            }
        }
        hipFree(tmp);
        if (rc) return rc;
        if (ndef && (src[aligned[a0]].p1 != t->dv.p1 || src[aligned[a0]].p2 != t->dv.p2)) {      // the growth changed the grid: the rest goes direct
            for (size_t a = a0 + na; a < aligned.size(); ++a) direct.push_back(aligned[a]);
            break;
        }
    }
    // The sources that go in directly are runs of CONSECUTIVE regions of their sender's grid (a chunk of the exchange): the same share of the
    // hash space as [g_lo, g_hi) is of this table's, since every grid orders its regions by the leading digits of one hash.  They fill that
    // share of the table, not the table: the fill limit has to hold there (the table's load as a whole says nothing about it -- a rank whose
    // own input was small receives far more than it held).  Senders' records may coincide, so this is an upper bound; growth keeps the grid.
    if (!direct.empty() && n_reg && n_reg < t->dev().n_regions && g_hi <= t->dev().n_regions) {
        rc = refresh_counters(t);
        if (rc) return rc;
        uint64_t direct_total = 0;
        for (uint32_t i : direct) direct_total += src[i].n_records;
        const double share = (double)n_reg / (double)t->dev().n_regions;
        unsigned long long occupied = 0;                          // what these regions hold already: counted (an exchange fills an emptied table share by share)
        {
            uint64_t* scratch = &t->dev().ctrs[CTR_SCRATCH];
            const uint64_t lo = (uint64_t)g_lo * t->dev().region_slots, hi = (uint64_t)g_hi * t->dev().region_slots;
            HIPCHK(c, hipMemsetAsync(scratch, 0, sizeof(uint64_t), c->stream));
            hipLaunchKernelGGL(k_occupied, dim3(grid_for(c, hi - lo, 256, 8)), dim3(256), 0, c->stream, t->dev(), lo, hi, (unsigned long long*)scratch);
            HIPCHK(c, hipMemcpyAsync(&occupied, scratch, sizeof occupied, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        const double present = (double)occupied;
        if (present + (double)direct_total > load_limit(t->dev()) * (double)t->dev().cap * share) {
            if (t->disable_grow) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
            uint64_t new_cap = t->dev().cap;
            while (present + (double)direct_total > 0.5 * (double)new_cap * share) new_cap *= 2;
            if (g_trace) fprintf(stderr, "[katgpu] merge: %llu record(s) for %u of %u regions: the table grows from %llu to %llu slots first\n", (unsigned long long)direct_total, n_reg,
                                 t->dev().n_regions, (unsigned long long)t->dev().cap, (unsigned long long)new_cap);
            rc = regrow(t, new_cap);
            if (rc) return rc;
        }
    }
    for (uint32_t i : direct) {
        if (src[i].dev_keys) { rc = merge_direct32(t, src[i].dev_keys, src[i].dev_counts, (size_t)src[i].n_records); if (rc) return rc; continue; }
        // packed records of another grid than the table's (it has grown since they were cut): the k-mer comes back from the SENDER's grid,
        // region by region, and goes through the direct path
        if (!n_reg) continue;
        rc = refresh_counters(t);
        if (rc) return rc;
        {
            const uint64_t lim = (uint64_t)(load_limit(t->dev()) * (double)t->dev().cap), room = lim > t->distinct ? lim - t->distinct : 0;
            if (room < src[i].n_records) { rc = ensure_room(t, src[i].n_records); if (rc) return rc; }
        }
        t->count_bound = 0xFFFFFFFFULL;
        uint8_t* tmp = nullptr;        // [off: (n_reg + 1) u64 | regions: n_reg u32]
        const size_t off_bytes = (size_t)(n_reg + 1) * 8;
        HIPCHK(c, hipMalloc((void**)&tmp, off_bytes + (size_t)n_reg * 4));
        uint64_t* d_off = (uint64_t*)tmp;
        uint32_t* d_regs = (uint32_t*)(tmp + off_bytes);
        std::vector<uint32_t> regs(n_reg);
        for (uint32_t g = 0; g < n_reg; ++g) regs[g] = g_lo + g;
        hipError_t e = hipMemcpyAsync(d_regs, regs.data(), (size_t)n_reg * 4, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, c->stream, src[i].dev_region_counts, n_reg, (uint64_t)n_reg, (const uint64_t*)nullptr, d_off, (uint64_t)(n_reg + 1), 1, (unsigned long long*)nullptr);
        MergeSrcs ms{};
        ms.n = 1;
        ms.src_p1 = src[i].p1; ms.src_n1 = place_n1(t->dev().k, src[i].p1); ms.src_l2 = (uint32_t)__builtin_ctz(src[i].p2);
        ms.s[0] = MergeSrc{nullptr, src[i].dev_counts, d_off, src[i].dev_rem_lo, src[i].dev_rem_hi};
        {
            ScopedTimer tm(c, KATGPU_K_MERGE, src[i].n_records);
            hipLaunchKernelGGL(k_merge_deferred, dim3(std::min<uint32_t>(n_reg, (uint32_t)c->n_cu * 8)), dim3(256), 0, c->stream, t->dev(), g_lo, ms, (const uint32_t*)d_regs, n_reg);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(tmp);
        if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    }
    return refresh_counters(t);
}

