// kg_context.hip -- the context behind katgpu_ctx: device selection, streams, the allocation pool that keeps tens of GB of freed
// table memory out of the driver's scrubber, per-class kernel timing, and the bench / test support calls (device buffers, the
// synthetic workload).
#include "kg_host.hpp"
#include "kg_kernels.hpp"

extern "C" const char* katgpu_version(void) { return "katgpu 0.1 (gfx950)"; }

extern "C" int katgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

extern "C" int katgpu_init(int device, katgpu_ctx** out) {
    if (!out) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return KATGPU_ERR_DEVICE;   // no CPU fallback, by design
    katgpu_ctx* c = new katgpu_ctx();
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    c->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return KATGPU_ERR_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "katgpu: device %d is %s; this library is built for gfx950 only\n", device, prop.gcnArchName);
        delete c; return KATGPU_ERR_DEVICE;
    }
    c->n_cu = prop.multiProcessorCount;
    {   // resident k_count blocks per CU (the API may over-report by one for SGPR-heavy kernels: keep it <= 8 and >= 1)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_count<true>), COUNT_BLOCK, 0) == hipSuccess && nb > 0)
            c->count_blocks_per_cu = std::min(nb, 8);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return KATGPU_ERR_DEVICE; }
    if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] context on device %d (%d CUs)\n", since_load(), device, c->n_cu);
    *out = c;
    return KATGPU_OK;
}

void resolve_pending(katgpu_ctx* c) {
    std::lock_guard<std::mutex> lk(c->prof_mu);
    for (auto& p : c->pending) {
        hipEventSynchronize(p.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->prof_ms[p.cls] += ms;
        c->event_pool.push_back(p.a); c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

extern "C" void katgpu_shutdown(katgpu_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream); hipStreamSynchronize(c->copy_stream);
    resolve_pending(c);
    if (c->reserve_thread.joinable()) c->reserve_thread.join();
    scan_cache_release(c);
    for (auto& b : c->pool) hipFree(b.p);
    c->pool.clear();
    if (c->reserved_p) hipFree(c->reserved_p);
    if (c->arena) hipFree(c->arena);
    for (auto e : c->event_pool) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (c->pinned[i]) hipHostFree(c->pinned[i]);
        if (c->ring[i]) hipFree(c->ring[i]);
        if (c->pin_free[i]) hipEventDestroy(c->pin_free[i]);
    }
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1);
    hipStreamDestroy(c->stream); hipStreamDestroy(c->copy_stream);
    delete c;
}

// Give the parked table arrays and the partition arena back to the driver (they are re-acquired on demand).  Callers that
// are about to allocate large buffers of their own on the same device (the multi-GPU exchange does) call this first.
// The memory of a table that will be asked for soon, allocated now on a thread of the library's and parked where the table's
// allocation finds it (the pool): a hipMalloc of 13 GB takes ~0.14 s, which `kat comp`'s second input would otherwise spend between
// its first input's last byte and its own first.  A hint: a table of another size simply does not find it (and a trim gives it back).
extern "C" int katgpu_reserve(katgpu_ctx* c, uint32_t k, uint64_t size_hint) {
    if (!c || k < 1 || k > KATGPU_MAX_K) return KATGPU_ERR_INVALID_ARG;
    const uint64_t cap = std::max<uint64_t>(size_hint, 1024);
    // packed 8-byte slots are what every one-word table of size gets (kg_table.hip); a grid's rounding adds at most four slots per region
    const size_t bytes = (size_t)((cap + ((uint64_t)4 << 20)) * 8);
    if (k > 32 || bytes < ((size_t)1 << 30)) return KATGPU_OK;                   // (wide tables are three arrays; small ones cost nothing to allocate)
    // ONE reservation at a time, kept apart from the pool: it is for a table made "like" another (katgpu_table_create_like: `kat comp`'s
    // second input), so the table being counted now -- often of the very same size, KAT's -H and -I share a default -- cannot walk
    // off with it; a second call while one is pending or parked is a no-op (a third input allocates when its turn comes).
    void* stale = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        const size_t have = c->reserve_bytes.load();
        if (have) {
            // the same size, give or take what pool_alloc accepts (up to a quarter more than asked for): the hint stands.  A PARKED block
            // of another size -- left by an earlier run, never matched by a create_like -- would sit in HBM (tens of GB) until a NOMEM
            // trim: it is given back and the new hint takes its place.  A reservation still on its way is left alone.
            if (!c->reserved_p || (have >= bytes && have <= bytes + bytes / 4)) return KATGPU_OK;
            stale = c->reserved_p; c->reserved_p = nullptr;
        }
        c->reserve_bytes.store(bytes);
    }
    if (stale) hipFree(stale);
    if (c->reserve_thread.joinable()) c->reserve_thread.join();
    c->reserve_thread = std::thread([c, bytes]() {
        hipSetDevice(c->device);
        // after what the count that follows needs sooner: its scan buffers, its table, its arena (katgpu_ctx::scan_waiting).  The caller
        // reserves, then counts: a moment for that count to announce itself, then its turn.
        { timespec ts{0, 100000000}; nanosleep(&ts, nullptr); }
        alloc_turn(c, true, 8000.0);
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); c->reserve_bytes.store(0); return; }
        std::lock_guard<std::mutex> lk(c->pool_mu);
        c->reserved_p = p;
    });
    return KATGPU_OK;
}

extern "C" int katgpu_release_scratch(katgpu_ctx* c) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    scan_cache_release(c);
    if (c->reserve_thread.joinable()) c->reserve_thread.join();
    pool_trim(c);                                                 // (the parked blocks and a parked reservation)
    if (c->arena) { hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0; }
    for (int i = 0; i < 2; ++i) if (c->ring[i]) { hipFree(c->ring[i]); c->ring[i] = nullptr; }
    c->ring_bytes = 0;
    c->arena_borrowed = false;
    return KATGPU_OK;
}

// Borrow the partitioned counter's arena as plain device scratch (grown to `bytes` if needed).  The multi-GPU exchange keeps
// its send / receive records here instead of allocating next to an arena that already holds most of the free HBM.  Valid
// until the next katgpu_count_* call on this context.
extern "C" int katgpu_scratch_acquire(katgpu_ctx* c, size_t bytes, void** dev_ptr, size_t* got_bytes) {
    if (!c || !dev_ptr) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->arena_bytes < bytes) {
        // the larger arena is tried BESIDE the old one first: the old one (sized to most of the free HBM) must survive a failure,
        // or a caller that could have made do with it -- the exchange simply takes more chunks -- is left with nothing
        for (auto& b : c->pool) hipFree(b.p);
        c->pool.clear();
        uint8_t* bigger = nullptr;
        hipError_t e = hipMalloc((void**)&bigger, bytes);
        if (e != hipSuccess && c->arena) {                       // no room for both: the old one goes, and comes back if that was not enough either
            (void)hipGetLastError();
            const size_t old_bytes = c->arena_bytes;
            hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0;
            e = hipMalloc((void**)&bigger, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (hipMalloc((void**)&c->arena, old_bytes) == hipSuccess) c->arena_bytes = old_bytes; else (void)hipGetLastError();
                return fail(c, KATGPU_ERR_NOMEM, "scratch of %zu bytes: %s (the arena keeps its %zu bytes)", bytes, hipGetErrorString(e), c->arena_bytes);
            }
        } else if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, KATGPU_ERR_NOMEM, "scratch of %zu bytes: %s", bytes, hipGetErrorString(e));
        }
        if (c->arena) hipFree(c->arena);
        c->arena = bigger; c->arena_bytes = bytes;
    }
    c->arena_borrowed = true;
    *dev_ptr = c->arena;
    if (got_bytes) *got_bytes = c->arena_bytes;
    return KATGPU_OK;
}

extern "C" const char* katgpu_last_error(const katgpu_ctx* c) { return c ? c->err.c_str() : "no context"; }

extern "C" int katgpu_sync(katgpu_ctx* c) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KATGPU_OK;
}

extern "C" int katgpu_profile_reset(katgpu_ctx* c) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    resolve_pending(c);
    memset(c->prof_launches, 0, sizeof c->prof_launches); memset(c->prof_units, 0, sizeof c->prof_units);
    for (auto& m : c->prof_ms) m = 0;
    return KATGPU_OK;
}

extern "C" int katgpu_profile_get(katgpu_ctx* c, int cls, uint64_t* launches, double* total_ms, uint64_t* units) {
    if (!c || cls < 0 || cls >= KATGPU_K_NCLASSES) return KATGPU_ERR_INVALID_ARG;
    resolve_pending(c);
    if (launches) *launches = c->prof_launches[cls];
    if (total_ms) *total_ms = c->prof_ms[cls];
    if (units) *units = c->prof_units[cls];
    return KATGPU_OK;
}

int grid_for(katgpu_ctx* c, uint64_t items, int block, int per_cu) {
    uint64_t need = (items + block - 1) / block;
    uint64_t cap = (uint64_t)c->n_cu * per_cu;            // persistent-style: a few resident blocks per CU, grid-stride the rest
    return (int)std::max<uint64_t>(1, std::min(need, cap));
}


// ------------------------------------------------------------------ allocation pool ------------------

static void pool_trim_locked(katgpu_ctx* c) {
    for (auto& b : c->pool) hipFree(b.p);
    c->pool.clear();
    if (c->reserved_p) { hipFree(c->reserved_p); c->reserved_p = nullptr; c->reserve_bytes.store(0); }      // (memory is short: the reservation was a hint)
}
void pool_trim(katgpu_ctx* c) {
    std::lock_guard<std::mutex> lk(c->pool_mu);
    pool_trim_locked(c);
}

hipError_t pool_alloc(katgpu_ctx* c, void** p, size_t bytes, bool take_reservation) {
    if (take_reservation) {   // a reservation in flight that could be what is asked for: wait for it (katgpu_reserve)
        const size_t rb = c->reserve_bytes.load();
        if (rb >= bytes && rb <= bytes + bytes / 4 && c->reserve_thread.joinable() && std::this_thread::get_id() != c->reserve_thread.get_id()) c->reserve_thread.join();
    }
    std::lock_guard<std::mutex> lk(c->pool_mu);
    size_t got = bytes;
    size_t* got_bytes = &got;
    struct Reg { katgpu_ctx* c; void** p; size_t* b; ~Reg() { if (*p) c->block_bytes[*p] = *b; } } reg{c, p, got_bytes};
    *p = nullptr;
    if (take_reservation && c->reserved_p) {                      // the block katgpu_reserve set aside for this table: handed out once
        const size_t rb = c->reserve_bytes.load();
        if (rb >= bytes && rb <= bytes + bytes / 4) {
            *p = c->reserved_p; *got_bytes = rb;
            c->reserved_p = nullptr; c->reserve_bytes.store(0);
            return hipSuccess;
        }
    }
    int best = -1;
    for (size_t i = 0; i < c->pool.size(); ++i)
        if (c->pool[i].bytes >= bytes && c->pool[i].bytes <= bytes + bytes / 4 && (best < 0 || c->pool[i].bytes < c->pool[best].bytes)) best = (int)i;
    if (best >= 0) {
        *p = c->pool[best].p; *got_bytes = c->pool[best].bytes;
        c->pool.erase(c->pool.begin() + best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    const bool arena_free = c->arena && !c->arena_borrowed && !c->arena_busy;
    if (e != hipSuccess && (!c->pool.empty() || arena_free)) {    // give cached scratch back and retry once
        (void)hipGetLastError();
        pool_trim_locked(c);
        if (arena_free) { hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0; }
        e = hipMalloc(p, bytes);
    }
    *got_bytes = bytes;
    return e;
}

void pool_release(katgpu_ctx* c, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    auto it = c->block_bytes.find(p);
    const size_t bytes = it == c->block_bytes.end() ? 0 : it->second;
    if (it != c->block_bytes.end()) c->block_bytes.erase(it);
    if (bytes < ((size_t)64 << 20) || c->pool.size() >= 8) { hipFree(p); return; }    // small blocks are not worth parking
    c->pool.push_back({p, bytes});
}


void release_arena(katgpu_ctx* c) {
    if (c->arena) { hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0; }
    c->arena_busy = false;
}

// ------------------------------------------------------------------ device buffers + synthetic workload

extern "C" int katgpu_dev_alloc(katgpu_ctx* c, size_t bytes, void** p) {
    if (!c || !p) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, bytes ? bytes : 16));
    return KATGPU_OK;
}
extern "C" int katgpu_dev_free(katgpu_ctx* c, void* p) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(p));
    return KATGPU_OK;
}
extern "C" int katgpu_dev_upload(katgpu_ctx* c, void* dst, const void* src, size_t n) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KATGPU_OK;
}
extern "C" int katgpu_dev_download(katgpu_ctx* c, void* dst, const void* src, size_t n) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KATGPU_OK;
}
extern "C" int katgpu_dev_mem_info(katgpu_ctx* c, uint64_t* free_b, uint64_t* total_b) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    if (free_b) *free_b = f;
    if (total_b) *total_b = t;
    return KATGPU_OK;
}

extern "C" int katgpu_synth_genome_device(katgpu_ctx* c, uint8_t* dev_out, uint64_t n, uint64_t seed, uint64_t contig_len) {
    if (!c || !dev_out) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_synth_genome, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, dev_out, n, seed, contig_len);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}

extern "C" int katgpu_synth_reads_device(katgpu_ctx* c, const uint8_t* dev_genome, uint64_t genome_len, uint8_t* dev_out,
                                         uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint32_t frag_len,
                                         uint32_t err_ppm, uint64_t seed) {
    if (!c || !dev_genome || !dev_out || read_len == 0 || read_len > 1023 || frag_len < read_len || genome_len < frag_len)
        return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t thresh = (uint32_t)(((uint64_t)err_ppm << 32) / 1000000ULL);
    hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(c, n_reads * (read_len + 1ULL), 256, 8)), dim3(256), 0, c->stream,
                       dev_genome, genome_len, dev_out, first_read, n_reads, read_len, frag_len, thresh, seed);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}

