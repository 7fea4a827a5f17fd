// kg_reduce.hip -- the reducers: katgpu_hist / katgpu_gcp / katgpu_comp / katgpu_comp3 (Histogram::bin, Gcp::analyse, Comp::compare).
#include "kg_host.hpp"
#include "kg_kernels.hpp"

// ------------------------------------------------------------------ reducers --------------------------

static const bool g_force_join = hook("KATGPU_FORCE_JOIN") != nullptr;  // tests: take the join form whenever it is legal
static const bool g_no_join = hook("KATGPU_NO_JOIN") != nullptr;       // A/B switch: force comp's probe form
static const bool g_no_seen = hook("KATGPU_NO_SEEN") != nullptr;       // A/B switch: pass 2 probes hash 1 even after a join pass 1
static const bool g_no_fused = hook("KATGPU_NO_FUSED") != nullptr;     // A/B switch: comp as two passes even where the fused join applies
// The comp and gcp kernels' LDS increments: one no-return atomic per lane (1), or aggregated per wave with ballots (0: a leader adds the
// count of the lanes that hit its cell).  The aggregation saves LDS conflicts on the hot cells and costs every slot ~14 instructions of a
// kernel that is bound by its instruction chain: same-box A/B (round 4) k_comp_fused 31.7 -> 27.9 ms at config 4, 46.8 -> 41.8 at config 5,
// k_gcp 3.96 -> 3.22 ms at config 3 with plain increments.  (k_hist keeps the aggregation: nearly every lane of a wave hits ONE bucket there.)
static const bool g_comp_plain_inc = hook_u64("KATGPU_COMP_PLAIN_INC", 1) != 0;
static const bool g_no_fold = hook("KATGPU_NO_FOLD") != nullptr;       // A/B switch: spectra by their own LDS atomics even for the tile's k-mers
static const uint32_t g_join_block = (uint32_t)hook_u64("KATGPU_JOIN_BLOCK", 512);   // A/B: threads per join workgroup (512 or 1024)

static int reducer_grid(katgpu_ctx* c, uint64_t slots, int blocks_per_cu) { return grid_for(c, slots, 256, blocks_per_cu); }

extern "C" int katgpu_hist(katgpu_table* t, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t* out, size_t nb) {
    if (!t || !out || nb == 0 || inc == 0 || ceil_ < base || nb != ceil_ + 1 - base) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, nb * 8));
    hipMemsetAsync(d, 0, nb * 8, c->stream);
    const uint32_t lds_bins = (uint32_t)std::min<uint64_t>(nb, 16384);          // 64 KB of u32 -> two blocks per CU
    {
        ScopedTimer tm(c, KATGPU_K_HIST, t->dev().cap);
        if (t->dev().cbits)
            hipLaunchKernelGGL(k_hist<true>, dim3(grid_for(c, t->dev().cap / 4, SCAN_BLOCK, 2)), dim3(SCAN_BLOCK), lds_bins * sizeof(uint32_t), c->stream,
                               t->dev(), t->n_ovf, base, ceil_, inc, (uint64_t)nb, d, lds_bins);
        else
            hipLaunchKernelGGL(k_hist<false>, dim3(grid_for(c, t->dev().cap / 4, SCAN_BLOCK, 2)), dim3(SCAN_BLOCK), lds_bins * sizeof(uint32_t), c->stream,
                               t->dev(), t->n_ovf, base, ceil_, inc, (uint64_t)nb, d, lds_bins);
    }
    hipMemcpyAsync(out, d, nb * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_gcp(katgpu_table* t, double cvg_scale, uint32_t cvg_bins, uint64_t* out) {
    if (!t || !out) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    const size_t cells = (size_t)t->dev().k * ((size_t)cvg_bins + 1);
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, cells * 8));
    hipMemsetAsync(d, 0, cells * 8, c->stream);
    const size_t lds = cells * sizeof(uint32_t);
    const uint32_t use_lds = lds <= 150 * 1024 ? (g_comp_plain_inc ? 2 : 1) : 0;   // 160 KB LDS per CU
    const int per_cu = use_lds && lds > 75 * 1024 ? 1 : 2;
#define KG_GCP(W, PK) do { \
        if (use_lds && lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_gcp<W, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ScopedTimer tm(c, KATGPU_K_GCP, t->dev().cap); \
        hipLaunchKernelGGL((k_gcp<W, PK>), dim3(grid_for(c, t->dev().cap / 4, SCAN_BLOCK, per_cu)), dim3(SCAN_BLOCK), use_lds ? lds : 0, c->stream, t->dev(), t->n_ovf, cvg_scale, cvg_bins, d, use_lds); } while (0)
    if (t->dev().keys_b) KG_GCP(true, false);
    else if (t->dev().cbits) {                                  // packed: a wave per region (kg_kernels.hpp: k_gcp_pk)
        if (use_lds && lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_gcp_pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ScopedTimer tm(c, KATGPU_K_GCP, t->dev().cap);
        const uint32_t want = (t->dev().n_regions + SCAN_BLOCK / 64 - 1) / (SCAN_BLOCK / 64);
        hipLaunchKernelGGL(k_gcp_pk, dim3(std::min<uint32_t>(want, (uint32_t)c->n_cu * per_cu)), dim3(SCAN_BLOCK), use_lds ? lds : 0, c->stream, t->dev(), t->n_ovf, cvg_scale, cvg_bins, d, use_lds);
    }
    else KG_GCP(false, false);
#undef KG_GCP
    HIPCHK(c, hipGetLastError());
    hipMemcpyAsync(out, d, cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_comp(katgpu_table* t1, katgpu_table* t2, int canon1, int canon2, double d1_scale, double d2_scale,
                           uint32_t d1_bins, uint32_t d2_bins, uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra) {
    (void)canon1;
    if (!t1 || !t2 || !main_mx || !counters || !spectra || d1_bins == 0 || d2_bins == 0) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t1->ctx;
    if (t1->ctx != t2->ctx) return fail(c, KATGPU_ERR_INVALID_ARG, "tables belong to different contexts");
    if (t1->dev().k != t2->dev().k)
        return fail(c, KATGPU_ERR_MISMATCH, "Cannot process hashes that were created with different K-mer lengths.  Expected: %u.  Key length was %u", t1->dev().k, t2->dev().k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t1); if (rc) return rc;
    rc = refresh_counters(t2); if (rc) return rc;
    const bool wide = t1->dev().keys_b != nullptr;               // same k => same key width in both tables
    const uint32_t ss = std::min(d1_bins, d2_bins);
    const size_t mx_cells = (size_t)d1_bins * d2_bins, total = mx_cells + 13 + 4 * (size_t)ss;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, total * 8));
    hipMemsetAsync(d, 0, total * 8, c->stream);
    CompArgs a{};
    a.d1_scale = d1_scale; a.d2_scale = d2_scale; a.d1_bins = d1_bins; a.d2_bins = d2_bins; a.spec_size = ss;
    a.canon_probe = canon2 ? 1 : 0;
    a.main_mx = d; a.counters = d + mx_cells; a.spectra = d + mx_cells + 13;
    const size_t lds1 = 16 * 8 + COMP_TILE * COMP_TILE * 4 + 3 * (size_t)ss * 4, lds2 = 16 * 8 + COMP_TILE * COMP_TILE * 4 + (size_t)ss * 4;
    if (lds1 > 150 * 1024) { hipFree(d); return fail(c, KATGPU_ERR_INVALID_ARG, "min(d1_bins,d2_bins) = %u too large for the LDS-privatised spectra", ss); }
    if (lds1 > 64 * 1024) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(wide ? k_comp<1, true> : k_comp<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(wide ? k_comp<2, true> : k_comp<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    }
    // Join form (region r of one table against region r of the other, in LDS) whenever the two tables share the region grid
    // and the probe key equals the stored key; probe form (random HBM probes) otherwise.  Tables of one grid have one layout
    // (the remainder bits follow from k and the grid), so the join is KV12 against KV12 or packed against packed.
    const bool pk = t1->dev().cbits != 0;
    const bool same_grid = !wide && t1->dev().p1 == t2->dev().p1 && t1->dev().p2 == t2->dev().p2 && t1->dev().cbits == t2->dev().cbits && t1->dev().n_regions > 1 && !g_no_join;
    const bool ident1 = t1->dev().canonical || !canon2;          // pass 1 probes canonical(key) iff input 2 is canonical
    const bool ident2 = t2->dev().canonical != 0;                // pass 2 always probes canonical(key)
    const size_t sb = pk ? 8 : 12;                           // bytes per slot
    const size_t join1 = ((lds1 + 15) & ~(size_t)15) + (size_t)t2->dev().region_slots * sb, join2 = ((lds2 + 15) & ~(size_t)15) + (size_t)t1->dev().region_slots * sb;
    // the join streams BOTH tables (~2.2 TB/s measured); probing costs ~1.3 random sector reads per scanned k-mer (~55 G/s):
    // a small table scanned against a big one is cheaper probed, a big one against a small one is cheaper joined
    auto join_pays = [&](const katgpu_table* scan, const katgpu_table* probe) {
        if (scan->dev().cap + probe->dev().cap < ((uint64_t)64 << 20)) return true;        // small either way: take the join
        return (double)sb * (double)(scan->dev().cap + probe->dev().cap) / 2.2e12 < 1.3 * (double)scan->distinct / 55e9;
    };
    // Both tables packed, on one grid, canonical: ONE kernel does both passes (k_comp_fused), the table with fewer k-mers streaming
    // past the other one's regions in LDS.
    const bool swap = t1->distinct < t2->distinct;           // hash 1 is the smaller one: it streams, hash 2 is resident
    const uint32_t s_res = swap ? t2->dev().region_slots : t1->dev().region_slots;
    const size_t fused_lds = ((16 * 8 + COMP_TILE * COMP_TILE * 4 + 4 * (size_t)ss * 4 + 15) & ~(size_t)15) + ((size_t)s_res + FUSED_STEP) * 8 + (size_t)(FUSED_BLOCK / 64) * FUSED_QCAP * 12 + (size_t)((s_res + 31) / 32) * 4;
    const uint32_t s_str = swap ? t1->dev().region_slots : t2->dev().region_slots;
    const bool fused = pk && same_grid && !g_no_fused && t1->dev().canonical && t2->dev().canonical && s_res <= (uint32_t)FUSED_KP * FUSED_BLOCK * 2 &&
                       s_str <= (uint32_t)FUSED_KP * FUSED_BLOCK * 2 && fused_lds <= 160 * 1024 - 512;
    // pass 1 as a join can leave a bit per slot of hash 2 ("hash 1 holds this k-mer"); pass 2 is then a scan of hash 2 (k_comp_seen)
    const bool join_1 = same_grid && ident1 && (g_force_join || join_pays(t1, t2));
    const uint32_t wpr = (t2->dev().region_slots + 31) / 32;
    const bool marked = !fused && join_1 && !g_no_seen && t1->dev().canonical && t2->dev().canonical && t2->ones == 0 && join1 + (size_t)wpr * 4 <= 150 * 1024;
    uint32_t* seen_bits = nullptr;
    if (marked) {
        if (pool_alloc(c, (void**)&seen_bits, (size_t)t2->dev().n_regions * wpr * 4) != hipSuccess) { (void)hipGetLastError(); seen_bits = nullptr; }
        a.seen = seen_bits; a.seen_wpr = wpr;
    }
    // unscaled matrices of more than COMP_TILE bins (KAT's defaults): the spectra of the k-mers that land in the LDS tile are its marginals
    a.plain_inc = g_comp_plain_inc ? 1 : 0;
    a.fold = !g_no_fold && d1_scale == 1.0 && d2_scale == 1.0 && d1_bins > COMP_TILE && d2_bins > COMP_TILE ? 1 : 0;
    const uint32_t jblk = g_join_block == 1024 ? 1024 : 512;
    auto join_grid = [&](size_t lds, uint32_t regions) {
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / (align_up(lds + 64, 1280)), 2048 / jblk));
        return std::min<uint32_t>(regions, (uint32_t)c->n_cu * (uint32_t)per_cu);
    };
    if (fused) {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS1, t1->dev().cap + t2->dev().cap);
        const uint32_t grid = std::min<uint32_t>(t1->dev().n_regions, (uint32_t)c->n_cu);
        // the streamed region in registers, 16-byte pairs of slots per lane (4 or 5: regions of up to 8192 / 10240 slots); 64-bit counts only
        // where a side table holds some (kg_kernels.hpp: k_comp_fused)
        const bool five = s_str > 4 * FUSED_BLOCK * 2, ovf = t1->n_ovf != 0 || t2->n_ovf != 0;
#define KG_FUSED(SWAP, JP, OVF) do { \
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_fused<SWAP, JP, OVF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); \
            hipLaunchKernelGGL((k_comp_fused<SWAP, JP, OVF>), dim3(grid), dim3(FUSED_BLOCK), fused_lds, c->stream, t1->dev(), t1->n_ovf, t2->dev(), t2->n_ovf, a); } while (0)
#define KG_FUSED_S(SWAP) do { if (five) { if (ovf) KG_FUSED(SWAP, 5, true); else KG_FUSED(SWAP, 5, false); } else { if (ovf) KG_FUSED(SWAP, 4, true); else KG_FUSED(SWAP, 4, false); } } while (0)
        if (swap) KG_FUSED_S(true); else KG_FUSED_S(false);
#undef KG_FUSED_S
#undef KG_FUSED
    }
#define KG_JOIN(PASS, TA, TB, LDS) do { \
        if (pk) { HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_join<PASS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
                  hipLaunchKernelGGL((k_comp_join<PASS, true>), dim3(join_grid(LDS, TA->dev().n_regions)), dim3(jblk), LDS, c->stream, TA->dev(), TA->n_ovf, TB->dev(), TB->n_ovf, a); } \
        else { HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_join<PASS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
               hipLaunchKernelGGL((k_comp_join<PASS, false>), dim3(join_grid(LDS, TA->dev().n_regions)), dim3(jblk), LDS, c->stream, TA->dev(), TA->n_ovf, TB->dev(), TB->n_ovf, a); } } while (0)
    if (!fused) {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS1, t1->dev().cap);
        if (join_1 && join1 <= 150 * 1024) {
            const size_t j1 = join1 + (a.seen ? (size_t)wpr * 4 : 0);
            KG_JOIN(1, t1, t2, j1);
        } else if (wide)
            hipLaunchKernelGGL((k_comp<1, true>), dim3(reducer_grid(c, t1->dev().cap + 1, 4)), dim3(256), lds1, c->stream, t1->dev(), t1->n_ovf, t2->dev(), t2->n_ovf, a);
        else
            hipLaunchKernelGGL((k_comp<1, false>), dim3(reducer_grid(c, t1->dev().cap + 1, 4)), dim3(256), lds1, c->stream, t1->dev(), t1->n_ovf, t2->dev(), t2->n_ovf, a);
    }
    if (!fused) {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS2, t2->dev().cap);
        if (a.seen && join1 <= 150 * 1024) {
            const uint32_t grid = std::min<uint32_t>(t2->dev().n_regions, (uint32_t)c->n_cu * 4);
            if (pk) hipLaunchKernelGGL(k_comp_seen<true>, dim3(grid), dim3(512), lds2, c->stream, t2->dev(), t2->n_ovf, a);
            else hipLaunchKernelGGL(k_comp_seen<false>, dim3(grid), dim3(512), lds2, c->stream, t2->dev(), t2->n_ovf, a);
        } else if (same_grid && ident2 && join2 <= 150 * 1024 && (g_force_join || join_pays(t2, t1))) {
            KG_JOIN(2, t2, t1, join2);
        } else if (wide)
            hipLaunchKernelGGL((k_comp<2, true>), dim3(reducer_grid(c, t2->dev().cap + 1, 4)), dim3(256), lds2, c->stream, t2->dev(), t2->n_ovf, t1->dev(), t1->n_ovf, a);
        else
            hipLaunchKernelGGL((k_comp<2, false>), dim3(reducer_grid(c, t2->dev().cap + 1, 4)), dim3(256), lds2, c->stream, t2->dev(), t2->n_ovf, t1->dev(), t1->n_ovf, a);
    }
#undef KG_JOIN
    HIPCHK(c, hipGetLastError());
    hipMemcpyAsync(main_mx, d, mx_cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(counters, d + mx_cells, 13 * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(spectra, d + mx_cells + 13, 4 * (size_t)ss * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (seen_bits) pool_release(c, seen_bits);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_comp3(katgpu_table* t1, katgpu_table* t2, katgpu_table* t3, int canon1, int canon2, int canon3,
                            double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins, uint64_t* main_mx,
                            uint64_t* ends_mx, uint64_t* middle_mx, uint64_t* mixed_mx, uint64_t counters[13], uint64_t* spectra) {
    if (!t3 || !ends_mx || !middle_mx || !mixed_mx) return KATGPU_ERR_INVALID_ARG;
    int rc = katgpu_comp(t1, t2, canon1, canon2, d1_scale, d2_scale, d1_bins, d2_bins, main_mx, counters, spectra);
    if (rc) return rc;
    katgpu_ctx* c = t1->ctx;
    if (t3->ctx != c) return fail(c, KATGPU_ERR_INVALID_ARG, "tables belong to different contexts");
    if (t3->dev().k != t1->dev().k)
        return fail(c, KATGPU_ERR_MISMATCH, "Cannot process hashes that were created with different K-mer lengths.  Expected: %u.  Key length was %u", t1->dev().k, t3->dev().k);
    rc = refresh_counters(t3); if (rc) return rc;
    const size_t cells = (size_t)d1_bins * d2_bins;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, (3 * cells + 13) * 8));
    hipMemsetAsync(d, 0, (3 * cells + 13) * 8, c->stream);
    Comp3Args a{};
    a.d1_scale = d1_scale; a.d2_scale = d2_scale; a.d1_bins = d1_bins; a.d2_bins = d2_bins;
    a.canon2 = canon2 ? 1 : 0; a.canon3 = canon3 ? 1 : 0;
    a.mx[0] = d; a.mx[1] = d + cells; a.mx[2] = d + 2 * cells;
    {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS1, t1->dev().cap);
        if (t1->dev().keys_b)
            hipLaunchKernelGGL(k_comp3_pass1<true>, dim3(reducer_grid(c, t1->dev().cap + 1, 3)), dim3(256), 3 * COMP_TILE * COMP_TILE * sizeof(uint32_t), c->stream,
                               t1->dev(), t1->n_ovf, t2->dev(), t2->n_ovf, t3->dev(), t3->n_ovf, a);
        else
            hipLaunchKernelGGL(k_comp3_pass1<false>, dim3(reducer_grid(c, t1->dev().cap + 1, 3)), dim3(256), 3 * COMP_TILE * COMP_TILE * sizeof(uint32_t), c->stream,
                               t1->dev(), t1->n_ovf, t2->dev(), t2->n_ovf, t3->dev(), t3->n_ovf, a);
        hipLaunchKernelGGL(k_comp3_pass3, dim3(reducer_grid(c, t3->dev().cap + 1, 8)), dim3(256), 0, c->stream, t3->dev(), t3->n_ovf, d + 3 * cells);
    }
    HIPCHK(c, hipGetLastError());
    uint64_t c3[13];
    hipMemcpyAsync(ends_mx, d, cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(middle_mx, d + cells, cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(mixed_mx, d + 2 * cells, cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(c3, d + 3 * cells, sizeof c3, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    counters[CC_H3_TOTAL] = c3[CC_H3_TOTAL];
    counters[CC_H3_DISTINCT] = c3[CC_H3_DISTINCT];
    return KATGPU_OK;
}

