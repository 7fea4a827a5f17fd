// katgpu.hip -- C ABI of libkatgpu.so (include/katgpu.h): context, HBM-resident table lifecycle, kernel launches.
// Host-side file ingest lives in kg_ingest.cpp.  gfx950 only; there is no CPU path in this library.
#include "../../include/katgpu.h"
#include "kg_ingest.hpp"
#include "kg_kernels.hpp"
#include "kg_partition.hpp"
#include "kg_superkmer.hpp"
#include "kg_wide.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace kg;

// ------------------------------------------------------------------ context ---------------------------

struct katgpu_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;        // all kernels
    hipStream_t copy_stream = nullptr;   // H2D staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    uint64_t prof_launches[KATGPU_K_NCLASSES] = {};
    double prof_ms[KATGPU_K_NCLASSES] = {};
    uint64_t prof_units[KATGPU_K_NCLASSES] = {};
    // pending (not yet read back) event pairs: timing is resolved lazily so launches stay asynchronous
    struct Pending { hipEvent_t a, b; int cls; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    // staging for host / file ingest: 2 pinned buffers feed 2 device rings (allocated on first use, kept)
    uint8_t* pinned[2] = {nullptr, nullptr};
    hipEvent_t pin_free[2] = {nullptr, nullptr};     // the H2D copy out of pinned[i] has finished
    size_t stage_bytes = 0;
    uint8_t* ring[2] = {nullptr, nullptr};           // resident stretches of the base stream, counted like any device-resident input
    size_t ring_bytes = 0;
    // Freed table arrays are parked here and handed out again to the next table of (nearly) the same size: on this
    // driver a hipMalloc of tens of GB right after a hipFree of that much stalls for seconds (VRAM scrubbing), which
    // would dominate a run that builds tables repeatedly.  Emptied by katgpu_shutdown or when an allocation fails.
    struct Block { void* p; size_t bytes; };
    std::vector<Block> pool;
    std::unordered_map<void*, size_t> block_bytes;      // real size of every live pooled-class allocation
    // scratch arena of the partitioned counter (level-1 / level-2 buffers, histograms); kept across calls
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    bool part_attr_set = false;
    bool arena_busy = false;              // a partition round is using it: pool_alloc must not free it to satisfy a table growth
    bool arena_borrowed = false;          // katgpu_scratch_acquire handed the arena out: it must not be freed behind the caller's back
    int count_blocks_per_cu = 6;
};

struct katgpu_table {
    katgpu_ctx* ctx = nullptr;
    DevTable d{};
    int disable_grow = 0;
    uint32_t n_ovf = 0;          // refreshed by refresh_counters()
    uint64_t distinct = 0;       // idem (slots in use + all-ones key)
    uint64_t ones = 0;
    // overflow guard of the unchecked (no-return) +1 adds: no 32-bit counter exceeds count_bound + unchecked_adds
    uint64_t count_bound = 0;    // largest counter value possible at the last sweep (0 for a fresh table)
    uint64_t unchecked_adds = 0; // window starts launched through k_count since then
    bool retrying = false;       // retry_failed is at work (refresh_counters must not re-enter it)
    uint32_t n_regrows = 0;      // how often the table had to grow (the host mirror words the reference's warning from it)
    uint8_t carry[64];           // last k-1 bytes of the previous host batch of the current file
    uint32_t carry_n = 0;
    bool lazy = false;           // the slots have not been initialised yet (KATGPU_LAZY_INIT, "lazy tables" below): TOUCH() or the first partition round does it
};

static int fail(katgpu_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}

// entry points that handle one-word k-mers only (lookups by 64-bit key, .jf, the multi-GPU exchange, sect/cold profiles)
#define NARROW_ONLY(t, what)                                                                             \
    do {                                                                                                 \
        if ((t)->d.keys_b) return fail((t)->ctx, KATGPU_ERR_K, "%s is not available for k > 32 (k = %u)", what, (t)->d.k); \
    } while (0)

#define HIPCHK(c, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return fail((c), _e == hipErrorOutOfMemory ? KATGPU_ERR_NOMEM : KATGPU_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char* katgpu_version(void) { return "katgpu 0.1 (gfx950)"; }

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
    *out = c;
    return KATGPU_OK;
}

static void resolve_pending(katgpu_ctx* c) {
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
    for (auto& b : c->pool) hipFree(b.p);
    c->pool.clear();
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
extern "C" int katgpu_release_scratch(katgpu_ctx* c) {
    if (!c) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& b : c->pool) hipFree(b.p);
    c->pool.clear();
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

// HIP events around a launch on the ctx stream; elapsed time is collected lazily.
struct ScopedTimer {
    katgpu_ctx* c; int cls; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(katgpu_ctx* c_, int cls_, uint64_t units) : c(c_), cls(cls_) {
        auto take = [&]() { hipEvent_t e = nullptr; if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else hipEventCreate(&e); return e; };
        a = take(); b = take();
        hipEventRecord(a, c->stream);
        c->prof_launches[cls] += 1; c->prof_units[cls] += units;
    }
    ~ScopedTimer() {
        hipEventRecord(b, c->stream);
        c->pending.push_back({a, b, cls});
        if (c->pending.size() > 4096) resolve_pending(c);
    }
};

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

static int grid_for(katgpu_ctx* c, uint64_t items, int block, int per_cu) {
    uint64_t need = (items + block - 1) / block;
    uint64_t cap = (uint64_t)c->n_cu * per_cu;            // persistent-style: a few resident blocks per CU, grid-stride the rest
    return (int)std::max<uint64_t>(1, std::min(need, cap));
}

// ------------------------------------------------------------------ table lifecycle ------------------

static double now_ms() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
static const bool g_trace = getenv("KATGPU_TRACE") != nullptr;
// Test hooks (KATGPU_TEST_*) and A/B switches are read only when KATGPU_TESTING is set: a production process ignores them, and the
// kernels that honour one (the spill hook of the first-edition apply) are separate instantiations it never launches.
// What a deployment may tune stays plain: KATGPU_TRACE, KATGPU_ARENA_FRACTION, KATGPU_RING_MB, KATGPU_PART_MIN_STARTS, KATGPU_INGEST_*.
static const bool g_testing = getenv("KATGPU_TESTING") != nullptr;
static const char* hook(const char* name) { return g_testing ? getenv(name) : nullptr; }

static void pool_trim(katgpu_ctx* c) {
    for (auto& b : c->pool) hipFree(b.p);
    c->pool.clear();
}

static hipError_t pool_alloc(katgpu_ctx* c, void** p, size_t bytes) {
    size_t got = bytes;
    size_t* got_bytes = &got;
    struct Reg { katgpu_ctx* c; void** p; size_t* b; ~Reg() { if (*p) c->block_bytes[*p] = *b; } } reg{c, p, got_bytes};
    *p = nullptr;
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
        pool_trim(c);
        if (arena_free) { hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0; }
        e = hipMalloc(p, bytes);
    }
    *got_bytes = bytes;
    return e;
}

static void pool_release(katgpu_ctx* c, void* p) {
    if (!p) return;
    auto it = c->block_bytes.find(p);
    const size_t bytes = it == c->block_bytes.end() ? 0 : it->second;
    if (it != c->block_bytes.end()) c->block_bytes.erase(it);
    if (bytes < ((size_t)64 << 20) || c->pool.size() >= 8) { hipFree(p); return; }    // small blocks are not worth parking
    c->pool.push_back({p, bytes});
}

static const bool g_force_join = hook("KATGPU_FORCE_JOIN") != nullptr;  // tests: take the join form whenever it is legal
static const bool g_no_join = hook("KATGPU_NO_JOIN") != nullptr;       // A/B switch: force comp's probe form
static const bool g_no_seen = hook("KATGPU_NO_SEEN") != nullptr;       // A/B switch: pass 2 probes hash 1 even after a join pass 1
static const uint32_t g_region_slots = hook("KATGPU_TEST_REGION_SLOTS") ? (uint32_t)strtoul(hook("KATGPU_TEST_REGION_SLOTS"), nullptr, 10) : REGION_SLOTS;

constexpr uint32_t MAX_REGION_SLOTS = 12288;        // 144 KB of LDS in the apply / join kernels
constexpr uint32_t AP2_MAX_SLOTS = 10240;           // the second-edition apply: 120 KB of region + 36 KB of queues (k_p3_apply2<1024, 5, ...>)
constexpr int AP2_QCAP_BIG = 192;                   // its queue entries per wave for regions beyond 8192 slots

// like_p1/like_p2 != 0: adopt that region grid (so that comp can join region against region) and take up the capacity in the
// region size, if a region of the resulting size still fits LDS.
// Minimizer regions (kg_device.hpp): tables of at least g_mz_min_regions regions key their regions by the k-mer's minimizer, which is
// what lets the super-k-mer counter partition runs of k-mers instead of k-mers.  Below that the spread of the minimizers' weights
// would not average out over a region.  like_mz: -1 = decide by size, else adopt (a table compared or merged with another must
// share its region function as well as its grid).
static const uint32_t g_mz_min_regions = hook("KATGPU_MZ_MIN_REGIONS") ? (uint32_t)strtoul(hook("KATGPU_MZ_MIN_REGIONS"), nullptr, 10) : 0xFFFFFFFFu;

static int alloc_dev_table(katgpu_ctx* c, uint32_t k, int canonical, uint64_t cap, DevTable* out, uint32_t like_p1 = 0, uint32_t like_p2 = 0, int like_mz = -1,
                           bool lazy = false /* leave the slots as they are: see "lazy tables" */) {
    DevTable d{};
    const uint64_t like_r = (uint64_t)like_p1 * like_p2;
    // capacity is a whole number of regions (kg_device.hpp: Probe); a table smaller than one region is a single short region
    // (regions of fewer than 256 slots are not worth a common grid: the spread of the region loads would eat the table's fill limit)
    if (like_r > 1 && (cap + like_r - 1) / like_r <= MAX_REGION_SLOTS - 4 && (cap + like_r - 1) / like_r >= 256) {
        d.p1 = like_p1; d.p2 = like_p2; d.n_regions = (uint32_t)like_r;
        d.region_slots = (uint32_t)((std::max<uint64_t>((cap + like_r - 1) / like_r, 16) + 3) & ~3ULL);   // whole 16-byte lines of keys and counts per region
    } else if (cap <= g_region_slots) { d.n_regions = d.p1 = d.p2 = 1; d.region_slots = (uint32_t)((cap + 3) & ~3ULL); }
    else {
        const uint64_t nr = (cap + g_region_slots - 1) / g_region_slots;
        if (nr > 0x3FFFFFFFULL) return fail(c, KATGPU_ERR_NOMEM, "table of %llu slots exceeds the region index", (unsigned long long)cap);
        uint32_t p2 = 1;
        while ((uint64_t)p2 * p2 < nr) ++p2;                   // two radix digits of about the same size
        if (k <= 32) { uint32_t q = 1; while (q < p2) q <<= 1; p2 = q; }   // one-word tables: the level-2 digit is a bit field of the placement hash
        d.p2 = p2; d.p1 = (uint32_t)((nr + p2 - 1) / p2);
        d.region_slots = g_region_slots;
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
    }
    const bool adopted = like_r > 1 && d.p1 == like_p1 && d.p2 == like_p2;
    d.mz = k > 32 ? 0 : (adopted && like_mz >= 0 ? (uint32_t)like_mz : (d.n_regions > 1 && d.n_regions >= g_mz_min_regions ? 1u : 0u));
    const double t0 = now_ms();
    const bool wide = k > 32;                                  // two key words per slot (kg_device.hpp "wide keys"), one block
    const size_t key_bytes = cap * sizeof(uint64_t) * (wide ? 2 : 1);
    HIPCHK(c, pool_alloc(c, (void**)&d.keys, key_bytes));
    if (wide) d.keys_b = d.keys + cap;
    if (g_trace) fprintf(stderr, "[katgpu] alloc keys %.1f GB: %.1f ms\n", cap * 8 / 1e9, now_ms() - t0);
    hipError_t e = pool_alloc(c, (void**)&d.counts, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d.ovf_keys, OVF_CAP * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d.ovf_hi, OVF_CAP * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d.ctrs, CTR_WORDS * sizeof(uint64_t));
    if (e == hipSuccess && d.mz) e = hipMalloc(&d.fail_buf, (size_t)2 * FAIL_CAP * sizeof(uint64_t));
    if (e != hipSuccess) {
        pool_release(c, d.keys); pool_release(c, d.counts); hipFree(d.ovf_keys); hipFree(d.ovf_hi); hipFree(d.ctrs); hipFree(d.fail_buf);
        return fail(c, KATGPU_ERR_NOMEM, "device allocation of a %llu-slot table failed: %s", (unsigned long long)cap, hipGetErrorString(e));
    }
    if (!lazy) {
        HIPCHK(c, hipMemsetAsync(d.keys, 0xFF, key_bytes, c->stream));
        HIPCHK(c, hipMemsetAsync(d.counts, 0, cap * sizeof(uint32_t), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(d.ovf_keys, 0xFF, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ovf_hi, 0, OVF_CAP * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.ctrs, 0, CTR_WORDS * sizeof(uint64_t), c->stream));
    if (g_trace) { const double t1 = now_ms(); hipStreamSynchronize(c->stream); fprintf(stderr, "[katgpu] table alloc: mallocs+enqueue %.1f ms, memsets done after %.1f ms more\n", t1 - t0, now_ms() - t1); }
    *out = d;
    return KATGPU_OK;
}

static void free_dev_table(katgpu_ctx* c, DevTable& d) {
    pool_release(c, d.keys); pool_release(c, d.counts);
    hipFree(d.ovf_keys); hipFree(d.ovf_hi); hipFree(d.ctrs); hipFree(d.fail_buf);
    d = DevTable{};
}

// ---- lazy tables (KATGPU_LAZY_INIT, a test hook: OFF) ----
// A fresh table is memset (39 + 20 GB of writes for config 4's first table) and then READ by its first partition round.  A lazy
// table skips the memset: its first partition round fills every region in LDS without loading it (k_p3_apply2<..., INIT>) and
// writes every region back, visited or not by k-mers.  Anything else that looks at the slots first -- TOUCH() at the top of every
// other entry point, the direct counter, a regrow -- initialises them the ordinary way.
static const bool g_lazy_init = hook("KATGPU_LAZY_INIT") != nullptr;
static int materialize(katgpu_table* t) {
    if (!t->lazy) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(t->d.keys, 0xFF, t->d.cap * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(t->d.counts, 0, t->d.cap * sizeof(uint32_t), c->stream));
    t->lazy = false;
    return KATGPU_OK;
}
#define TOUCH(T) do { katgpu_table* t__ = const_cast<katgpu_table*>(T); if (t__ && t__->lazy) { int rc__ = materialize(t__); if (rc__) return rc__; } } while (0)

extern "C" int katgpu_table_create(katgpu_ctx* c, uint32_t k, int canonical, uint64_t size_hint, int disable_grow, katgpu_table** out) {
    if (!c || !out) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (k < 1 || k > KATGPU_MAX_K) return fail(c, KATGPU_ERR_K, "k = %u unsupported: this build keeps a k-mer in at most two 63-bit words (1 <= k <= %d)", k, KATGPU_MAX_K);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t cap = std::max<uint64_t>(size_hint ? size_hint : (1u << 20), 1024);
    katgpu_table* t = new katgpu_table();
    t->ctx = c; t->disable_grow = disable_grow;
    t->lazy = g_lazy_init && k <= 32;
    int rc = alloc_dev_table(c, k, canonical, cap, &t->d, 0, 0, -1, t->lazy);
    if (rc) { delete t; return rc; }
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
    t->lazy = g_lazy_init && k <= 32;
    int rc = (k > 32) != (like->d.k > 32) ? alloc_dev_table(c, k, canonical, cap, &t->d, 0, 0, -1, t->lazy)     // no common grid across key widths
                                          : alloc_dev_table(c, k, canonical, cap, &t->d, like->d.p1, like->d.p2, (int)like->d.mz, t->lazy);
    if (rc) { delete t; return rc; }
    *out = t;
    return KATGPU_OK;
}

extern "C" void katgpu_table_free(katgpu_table* t) {
    if (!t) return;
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    free_dev_table(t->ctx, t->d);
    delete t;
}

extern "C" uint32_t katgpu_table_k(const katgpu_table* t) { return t ? t->d.k : 0; }
extern "C" uint32_t katgpu_table_regrows(const katgpu_table* t) { return t ? t->n_regrows : 0; }
extern "C" int katgpu_table_canonical(const katgpu_table* t) { return t ? (int)t->d.canonical : 0; }

static int retry_failed(katgpu_table* t, uint64_t n_failed);

// read the counter block back (one small D2H; synchronises the compute stream)
static int refresh_counters(katgpu_table* t) {
    katgpu_ctx* c = t->ctx;
    uint64_t h[CTR_WORDS];
    HIPCHK(c, hipMemcpyAsync(h, t->d.ctrs, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t d = 0;
    for (int i = 0; i < CTR_NSTRIPES; ++i) d += h[CTR_DISTINCT0 + i];
    t->ones = h[CTR_ONES];
    t->distinct = d + (t->ones ? 1 : 0);
    t->n_ovf = (uint32_t)h[CTR_OVF_USED];
    if (h[CTR_FULL]) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
    if (h[CTR_FAIL_N] && !t->retrying) return retry_failed(t, h[CTR_FAIL_N]);
    return KATGPU_OK;
}

// hash_counter::double_size (deps/jellyfish-2.2.0/include/jellyfish/hash_counter.hpp:204-244): allocate a larger array,
// re-insert every (key,count), swap.  Here one grid-stride kernel instead of a barrier-synchronised thread team.
static int regrow(katgpu_table* t, uint64_t new_cap) {
    katgpu_ctx* c = t->ctx;
    DevTable nd{};
    int rc = alloc_dev_table(c, t->d.k, t->d.canonical, new_cap, &nd, t->d.n_regions > 1 ? t->d.p1 : 0, t->d.n_regions > 1 ? t->d.p2 : 0, (int)t->d.mz, t->lazy);
    if (rc) return rc;
    if (t->lazy) {                                             // never touched: there is nothing to re-insert (and nothing to read)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_dev_table(c, t->d);
        t->d = nd;
        ++t->n_regrows;
        return refresh_counters(t);
    }
    {
        ScopedTimer tm(c, KATGPU_K_REGROW, t->d.cap);
        if (t->d.keys_b) hipLaunchKernelGGL(k_regrow_w, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, nd, t->d, t->n_ovf);
        else hipLaunchKernelGGL(k_regrow, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, nd, t->d, t->n_ovf);
    }
    HIPCHK(c, hipMemcpyAsync(&nd.ctrs[CTR_ONES], &t->d.ctrs[CTR_ONES], sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_dev_table(c, t->d);
    t->d = nd;
    ++t->n_regrows;
    t->count_bound = 0xFFFFFFFFULL;           // full counts were folded back into the slots: the next unchecked launch sweeps first
    t->unchecked_adds = 0;
    return refresh_counters(t);
}

// Minimizer-region tables: k-mers whose region was full were parked by the kernels (kg_device.hpp: table_park).  Grow and add them
// again -- a growth re-places every k-mer, and may park some itself, hence the loop.  With growth disabled a full region is what a
// full table is to the reference: "Hash full".
static int retry_failed(katgpu_table* t, uint64_t n_failed) {
    katgpu_ctx* c = t->ctx;
    if (t->disable_grow) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
    struct Guard { katgpu_table* t; explicit Guard(katgpu_table* t_) : t(t_) { t->retrying = true; } ~Guard() { t->retrying = false; } } guard(t);
    for (int attempt = 0; attempt < 12 && n_failed; ++attempt) {
        const uint64_t n = std::min<uint64_t>(n_failed, FAIL_CAP);
        uint64_t* tmp = nullptr;
        HIPCHK(c, hipMalloc((void**)&tmp, n * 16));
        HIPCHK(c, hipMemcpyAsync(tmp, t->d.fail_buf, n * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(tmp + n, t->d.fail_buf + FAIL_CAP, n * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(&t->d.ctrs[CTR_FAIL_N], 0, sizeof(uint64_t), c->stream));
        if (g_trace) fprintf(stderr, "[katgpu] %llu k-mers found their (minimizer) region full: growing %llu -> %llu slots\n", (unsigned long long)n_failed,
                             (unsigned long long)t->d.cap, (unsigned long long)t->d.cap * 2);
        int rc = regrow(t, t->d.cap * 2);                         // (its refresh_counters does not recurse: `retrying`)
        if (rc == KATGPU_OK) {
            ScopedTimer tm(c, KATGPU_K_MERGE, n);
            hipLaunchKernelGGL(k_merge, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, t->d, (const uint64_t*)tmp, (const uint64_t*)(tmp + n), n);
        }
        uint64_t h[CTR_WORDS];
        if (rc == KATGPU_OK && (hipMemcpyAsync(h, t->d.ctrs, sizeof h, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
            rc = fail(c, KATGPU_ERR_DEVICE, "retry of parked k-mers");
        hipFree(tmp);
        if (rc) return rc;
        if (h[CTR_FULL]) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
        n_failed = h[CTR_FAIL_N];
    }
    if (n_failed) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
    return refresh_counters(t);
}

// The fill limit of the direct path.  A k-mer probes inside its region only, so it is the fullest REGION that must not run out
// of slots: regions get Binomial(n, 1/R) k-mers, and small regions (a table created "like" a much bigger one) need more slack
// than the 0.7 that suits regions of thousands of slots.
static double load_limit(const DevTable& d) {
    if (d.region_slots >= 1024 || d.n_regions == 1) return 0.7;
    return std::max(0.25, 0.7 - 3.0 / std::sqrt((double)d.region_slots));
}

// Make room for up to `incoming` new distinct k-mers (an upper bound: one per window start) at load <= the fill limit.
static int ensure_room(katgpu_table* t, uint64_t incoming) {
    int rc = refresh_counters(t);
    if (rc) return rc;
    const uint64_t need = t->distinct + incoming;
    if ((double)need <= load_limit(t->d) * (double)t->d.cap) return KATGPU_OK;
    if (t->disable_grow) return fail(t->ctx, KATGPU_ERR_TABLE_FULL, "Hash full");
    uint64_t new_cap = t->d.cap;
    while ((double)need > 0.5 * (double)new_cap) new_cap *= 2;
    return regrow(t, new_cap);
}

// ------------------------------------------------------------------ counting --------------------------

// test hooks (tests/test_gpu_parity.py): shrink the sweep threshold / the launch size so small inputs exercise them
static const uint64_t g_test_sweep_thr = hook("KATGPU_TEST_SWEEP_THR") ? strtoull(hook("KATGPU_TEST_SWEEP_THR"), nullptr, 10) : 0;
static const uint64_t g_test_max_starts = hook("KATGPU_TEST_MAX_STARTS") ? strtoull(hook("KATGPU_TEST_MAX_STARTS"), nullptr, 10) : 0;

// k_count adds with no-return atomics and cannot see a 32-bit wrap; make one impossible.  Invariant: every counter
// <= count_bound + unchecked_adds.  When the next launch could break "<= 2^32-1", k_sweep moves multiples of thr out of
// the large counters into the side table and reports the new maximum.
static int maybe_sweep(katgpu_table* t, uint64_t next_starts) {
    katgpu_ctx* c = t->ctx;
    const uint64_t limit = g_test_sweep_thr ? 2 * g_test_sweep_thr - 1 : 0xFFFFFFFFULL;
    if (t->count_bound + t->unchecked_adds + next_starts <= limit) return KATGPU_OK;
    const uint32_t thr = g_test_sweep_thr ? (uint32_t)g_test_sweep_thr : 0x80000000u;
    unsigned long long* scratch = (unsigned long long*)&t->d.ctrs[CTR_SCRATCH];
    HIPCHK(c, hipMemsetAsync(scratch, 0, sizeof(uint64_t), c->stream));
    {
        ScopedTimer tm(c, KATGPU_K_REGROW, t->d.cap);
        hipLaunchKernelGGL(k_sweep, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, t->d, thr, scratch);
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
    if (n < t->d.k) return KATGPU_OK;
    if (t->d.keys_b) {                                         // wide k-mers: checked adds, nothing to sweep
        const uint64_t n_chunks = (n + WIDE_CHUNK_STARTS - 1) / WIDE_CHUNK_STARTS;
        const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * 4);
        ScopedTimer tm(c, KATGPU_K_COUNT, n);
        if ((reinterpret_cast<uintptr_t>(dev_bases) & 15) == 0)
            hipLaunchKernelGGL(k_count_w<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, dev_bases, (uint64_t)n, n_chunks);
        else
            hipLaunchKernelGGL(k_count_w<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, dev_bases, (uint64_t)n, n_chunks);
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
        hipLaunchKernelGGL(k_count<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, dev_bases, (uint64_t)n, n_chunks);
    else
        hipLaunchKernelGGL(k_count<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, dev_bases, (uint64_t)n, n_chunks);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}


// ------------------------------------------------------------------ partitioned counter (kg_partition.hpp) ----

static const uint64_t g_part_min_starts = getenv("KATGPU_PART_MIN_STARTS") ? strtoull(getenv("KATGPU_PART_MIN_STARTS"), nullptr, 10) : (32ULL << 20);
static const uint64_t g_test_round_items = hook("KATGPU_TEST_ROUND_ITEMS") ? strtoull(hook("KATGPU_TEST_ROUND_ITEMS"), nullptr, 10) : 0;
// share of the free HBM the partition arena may take (multi-GPU runs may lower it; bench.py sets 0.75 there)
static const double g_arena_fraction = getenv("KATGPU_ARENA_FRACTION") ? std::min(0.95, std::max(0.05, atof(getenv("KATGPU_ARENA_FRACTION")))) : 0.85;
static const uint32_t g_p1_wgs = hook("KATGPU_P1_WGS") ? std::max<uint32_t>(1, (uint32_t)strtoul(hook("KATGPU_P1_WGS"), nullptr, 10)) : 3;   // level-1 workgroups per CU
static const uint32_t g_apply_v = hook("KATGPU_APPLY_V") ? (uint32_t)strtoul(hook("KATGPU_APPLY_V"), nullptr, 10) : 2;   // 1: first-edition walk (A/B)
static const bool g_apply_noinline = hook("KATGPU_APPLY_NOINLINE") != nullptr;   // A/B: no inline claims in a table's first round
static const uint32_t g_apply_block = hook("KATGPU_APPLY_BLOCK") ? (uint32_t)strtoul(hook("KATGPU_APPLY_BLOCK"), nullptr, 10) : 0;   // 0: by region size
// level 2 without its histogram pass (kg_partition.hpp: k_p2_fast): 0 = never, 1 = when the mean run is long enough for the
// capacity slack to cover the noise, 2 = always (tests).  KATGPU_TEST_P2_OVF_CAP shrinks the overflow list (tests: forces the
// fall back to the exact kernel).
// level 1 without its counting pass (kg_partition.hpp: k_p1v2_scatter<true>, one fixed-capacity segment per workgroup and bucket):
// 0 = never, 1 = for rounds of at least 64 M k-mers (the default), 2 = always (tests)
static const uint32_t g_test_l1_cpb = hook("KATGPU_TEST_L1_CPB") ? (uint32_t)strtoul(hook("KATGPU_TEST_L1_CPB"), nullptr, 10) : 0;   // tests: segment capacity (forces overflow)
static const uint32_t g_l1_fast = hook("KATGPU_L1_FAST") ? (uint32_t)strtoul(hook("KATGPU_L1_FAST"), nullptr, 10) : 1;
static const uint32_t g_p2_fast = hook("KATGPU_P2_FAST") ? (uint32_t)strtoul(hook("KATGPU_P2_FAST"), nullptr, 10) : 1;
static const uint64_t g_test_p2_ovf_cap = hook("KATGPU_TEST_P2_OVF_CAP") ? strtoull(hook("KATGPU_TEST_P2_OVF_CAP"), nullptr, 10) : 0;
static const uint32_t g_test_spill_mod = hook("KATGPU_TEST_SPILL_MOD") ? (uint32_t)strtoul(hook("KATGPU_TEST_SPILL_MOD"), nullptr, 10) : 0;

static const bool g_p2_stamp = hook("KATGPU_P2_STAMP") != nullptr;       // diagnostic: per-phase cycle stamps of k_p2_fast
static const uint32_t g_test_hb = hook("KATGPU_TEST_HB") ? (uint32_t)strtoul(hook("KATGPU_TEST_HB"), nullptr, 10) : 0;   // A/B: wider level-2 items than needed (1, 2, 4)
static bool part_geometry(const DevTable& d, PartGeom* g) {
    g->R = d.n_regions; g->S = d.region_slots; g->P1 = d.p1; g->P2 = d.p2; g->l2 = d.l2;
    g->b_lo = 0; g->b_hi = d.p1;
    g->pl = place_make(d.k, d.p1, d.n1, d.l2);
    g->hb = std::max(l2_hi_bytes(g->pl.rb), g_test_hb);
    return d.k <= 32 && g->pl.rb <= 63 /* all-ones is "no item" */ && g->P1 <= MAX_PARTS && g->P2 <= MAX_PARTS && (size_t)g->S * 12 <= 150 * 1024;
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
static double arena_bytes_per_item(uint32_t hb, uint32_t passes) { return 8.0 * (1 + 1.0 / 24) + (4.0 + hb) * (1 + 1.0 / 24) * l2_items_per_l1_item(hb) / passes + 0.25 + 0.02; }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static const bool g_test_grow_nomem = hook("KATGPU_TEST_GROW_NOMEM") != nullptr;   // tests: table growth "fails" while the arena is busy

static void release_arena(katgpu_ctx* c) {
    if (c->arena) { hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0; }
    c->arena_busy = false;
}

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
        if (min_cap > t->d.cap) {
            if (t->disable_grow) return fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
            uint64_t nc = t->d.cap; while (nc < min_cap) nc *= 2;
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
            hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(c, m, 256, 6)), dim3(256), 0, c->stream, t->d, (const uint64_t*)d, (uint64_t)m);
            if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "spill insert");
        }
        pool_release(c, d);
    }
    return rc;
}

// Count a resident, 16-byte aligned base stream through partition rounds.  *done = number of window starts consumed
// (all of them unless the geometry stops fitting, in which case the caller finishes with the direct kernel).
static int count_partitioned(katgpu_table* t, const uint8_t* dev_bases, size_t n, size_t* done) {
    katgpu_ctx* c = t->ctx;
    const uint32_t k = t->d.k;
    const size_t n_starts = n - k + 1;
    *done = 0;
    c->arena_borrowed = false;                    // a borrowed arena is only promised until the next count call
    if (n < 64) return KATGPU_OK;                 // (the tile loader reads whole 16-byte pieces: direct path)
    // A table hopelessly small for this input (KAT's default -H against a whole run) would spill nearly every k-mer of the
    // first round: give it room for 1/16 of the starts first -- cheap while it is still small, and before the arena exists.
    if (!g_test_round_items && !t->disable_grow && t->d.cap < n_starts / 16) {
        uint64_t nc = t->d.cap; while (nc < n_starts / 16) nc *= 2;
        int grc = regrow(t, nc);
        if (grc) return grc;
    }
    const uint32_t W = (uint32_t)c->n_cu * std::min<uint32_t>(g_p1_wgs, 4);                     // level-1 workgroups (rows of hist1 / offs)
    const uint32_t W2 = (uint32_t)c->n_cu;                                                      // level-2 / apply: one per CU
    const size_t tile_starts = P1_TILE_STARTS;
    if (!c->part_attr_set) {
#define KG_LDS_ATTR(K, BYTES) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
#define KG_FOR_HB(M) M(0) M(1) M(2) M(4)
#define KG_ATTR_HB(HB) \
        KG_LDS_ATTR((k_p2<HB>), sizeof(P2Lds<HB>)); KG_LDS_ATTR((k_p2_fast<HB>), sizeof(P2Lds<HB>)); \
        KG_LDS_ATTR((k_p3_apply2<1024, 4, 4, 3, HB>), 150 * 1024); KG_LDS_ATTR((k_p3_apply2<1024, 4, 4, 3, HB, false, true, true>), 150 * 1024); \
        KG_LDS_ATTR((k_p3_apply2<1024, 5, 4, 3, HB, false, false, true, AP2_QCAP_BIG>), 160 * 1024 - 256); KG_LDS_ATTR((k_p3_apply2<1024, 5, 4, 3, HB, false, true, true, AP2_QCAP_BIG>), 160 * 1024 - 256); \
        KG_LDS_ATTR((k_p3_apply2<512, 4, 4, 3, HB>), 150 * 1024); KG_LDS_ATTR((k_p3_apply2<512, 4, 4, 3, HB, false, true, true>), 150 * 1024); \
        KG_LDS_ATTR((k_p3_apply2<512, 2, 4, 3, HB>), 150 * 1024); KG_LDS_ATTR((k_p3_apply2<512, 2, 4, 3, HB, false, true, true>), 150 * 1024);
        KG_FOR_HB(KG_ATTR_HB)
#undef KG_ATTR_HB
        if (g_lazy_init) {
#define KG_ATTR_INIT(HB) \
            KG_LDS_ATTR((k_p3_apply2<1024, 4, 4, 3, HB, false, true, true, AP2_QCAP, true>), 150 * 1024); KG_LDS_ATTR((k_p3_apply2<512, 4, 4, 3, HB, false, true, true, AP2_QCAP, true>), 150 * 1024); \
            KG_LDS_ATTR((k_p3_apply2<512, 2, 4, 3, HB, false, true, true, AP2_QCAP, true>), 150 * 1024); KG_LDS_ATTR((k_p3_apply2<1024, 5, 4, 3, HB, false, true, true, AP2_QCAP_BIG, true>), 160 * 1024 - 256);
            KG_ATTR_INIT(1) KG_ATTR_INIT(2)
#undef KG_ATTR_INIT
        }
#define KG_ATTR_AP1(B, SPT) KG_LDS_ATTR((k_p3_apply<B, SPT, 4, false>), 150 * 1024); if (g_testing) KG_LDS_ATTR((k_p3_apply<B, SPT, 4, true>), 150 * 1024);
        KG_ATTR_AP1(1024, 4) KG_ATTR_AP1(1024, 8) KG_ATTR_AP1(1024, 12) KG_ATTR_AP1(512, 8) KG_ATTR_AP1(512, 16) KG_ATTR_AP1(512, 24)
#undef KG_ATTR_AP1
        c->part_attr_set = true;
    }
    // ---- arena: [hist1 | offs | l1_off | off2 | cnt2 | bend | spill_n, ovf_n | L1 buffer | L2 buffer | overflow list] ----
    // L1 buffer: a round's k-mers + 1/24 + 64 per workgroup and bucket (segment slack of k_p1v2_scatter<true>);
    // L2 buffer: items of 4 + hb bytes in groups of four (kg_partition.hpp "the level-2 buffer"): the L1 count + 1/16 + 16 per region
    // (capacity slack of k_p2_fast) + two items per tile and region (group padding); overflow list: 1/32.
    // 14.8 bytes per k-mer of a round at hb = 1 (k = 27 at the bench size), 18.5 at hb = 4 -- with one pass; the level-2 buffer
    // holds one PASS of level 2 + apply (a CU-full of buckets, see the rounds below): 11.7 bytes with two passes.
    PartGeom g0;
    if (!part_geometry(t->d, &g0)) return KATGPU_OK;                              // direct path
    const uint32_t hb0 = g0.hb;                                                  // a table that grows has more regions: never more remainder bits
    const uint32_t passes0 = std::max<uint32_t>(1, g0.P1 / pass_buckets(g0.P1, (uint32_t)c->n_cu));   // (rounded down: the buffer never too small)
    const double per_item = arena_bytes_per_item(hb0, passes0);
    constexpr size_t SEG_PAD = 64;
    const size_t fixed_l1 = (size_t)W * MAX_PARTS * SEG_PAD;
    const size_t fixed_l2 = (size_t)((double)fixed_l1 * l2_items_per_l1_item(hb0) / passes0) + (size_t)MAX_PARTS * MAX_PARTS * 32 + 8192;
    const size_t small_bytes = align_up((size_t)W * MAX_PARTS * 4, 256) + align_up((size_t)W * MAX_PARTS * 8, 256) +   /* W <= 4 * CUs */
                               align_up((MAX_PARTS + 1) * 8, 256) + align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256) +
                               align_up((size_t)MAX_PARTS * MAX_PARTS * 4, 256) + align_up((size_t)MAX_PARTS * 8, 256) + align_up((size_t)MAX_PARTS * 4, 256) + 256 +
                               (fixed_l1 + fixed_l2 + 4096) * 8 + 4096;
    size_t want_items = n_starts;
    if (g_test_round_items) want_items = std::min<size_t>(want_items, g_test_round_items);
    const size_t want_bytes = small_bytes + (size_t)((per_item + 0.5) * (double)want_items);
    if (c->arena_bytes < want_bytes) {                                           // the arena could be more useful than it is
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        free_b += c->arena_bytes;
        size_t bytes = std::min<size_t>(want_bytes, (size_t)(g_arena_fraction * (double)free_b));
        // re-allocate only for a substantially larger arena (fewer rounds): a fresh hipMalloc of this size is not free
        if (bytes > c->arena_bytes + c->arena_bytes / 2 || c->arena_bytes < small_bytes + 18 * std::min<size_t>(want_items, (size_t)64 << 20)) {
            if (c->arena) { HIPCHK(c, hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
            if (!g_test_round_items && bytes < small_bytes + 18 * ((size_t)1 << 20)) return KATGPU_OK;   // no room for a useful round: direct path
            if (hipMalloc((void**)&c->arena, bytes) != hipSuccess) { (void)hipGetLastError(); c->arena = nullptr; return KATGPU_OK; }   // direct path
            c->arena_bytes = bytes;
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
    const size_t round_items = std::min<size_t>(want_items, (size_t)((double)(c->arena_bytes - small_bytes) / per_item));
    const size_t l1_items = round_items + round_items / 24 + fixed_l1;
    const size_t l2_items = ((size_t)((double)l1_items * l2_items_per_l1_item(hb0) / passes0) + (size_t)MAX_PARTS * MAX_PARTS * 32 + 4096 + 3) & ~(size_t)3;
    uint64_t* l1_buf = (uint64_t*)a;
    uint8_t* l2_buf = (uint8_t*)(l1_buf + l1_items);                               // level-2 items, groups of 4
    uint64_t* ovf_buf = (uint64_t*)(l2_buf + align_up(l2_items / 4 * l2_group_bytes(hb0), 16));
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
        if ((double)t->distinct > 0.6 * (double)t->d.cap) {
            bool lost = false;
            rc = grow_beside_arena(t, 0, t->d.cap * 2, KeyLists(), &lost);
            if (rc) return rc;
            if (lost) break;                                                      // the caller re-enters with a fresh arena
        }
        PartGeom g;
        if (!part_geometry(t->d, &g) || t->d.mz) break;                           // table too large for two levels: direct path; grown into minimizer regions: the other counter
        if (g.hb > hb0) break;                                                    // (cannot happen: see hb0) the level-2 carve would not hold these items
        // a lazy table: this round's apply initialises every region itself (k_p3_apply2<..., INIT>) when it is the kind of round that can
        const uint32_t blk0 = g_apply_block ? g_apply_block : (g.S <= 4096 ? 512 : 1024);
        const bool init_round = t->lazy && t->distinct == 0 && !g_apply_noinline && g_apply_v != 1 && !g_test_spill_mod && g.S % 4 == 0 && g.S >= 64 &&
                                g.S <= AP2_MAX_SLOTS && (blk0 == 512 ? g.S <= 4096 : true) && (g.hb == 1 || g.hb == 2);
        if (t->lazy && !init_round) { rc = materialize(t); if (rc) return rc; }
        // (the segmented level 1 sizes its segments from this ratio, so it wants it even when one round takes everything)
        if (!ratio_known && !g_test_round_items && (n_starts - pos > round_items || (l1_fast_ok && n_starts - pos >= ((size_t)64 << 20)))) {
            const size_t probe_m = std::min<size_t>(n_starts - pos, (size_t)64 << 20) / tile_starts * tile_starts;
            const uint64_t pt = probe_m / tile_starts, ptw = (pt + W - 1) / W;
            hipLaunchKernelGGL(k_p1v2_count, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, dev_bases + pos, (uint64_t)(probe_m + k - 1), pt, ptw, hist1);
            hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            uint64_t probe_items = 0;
            HIPCHK(c, hipMemcpyAsync(&probe_items, &l1_off[g.P1], sizeof probe_items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // (the probe's all-ones tally must not count twice: the real count pass over the same prefix follows)
            if (t->d.k == 32 && !t->d.canonical) HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
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
        uint64_t seg_cap = est_items / ((uint64_t)W * g.P1);
        seg_cap += seg_cap / 24 + SEG_PAD;
        if (g_test_l1_cpb) seg_cap = std::min<uint64_t>(seg_cap, g_test_l1_cpb);
        const bool seg = l1_fast_ok && (g_l1_fast == 2 || (ratio_known && est_items >= ((uint64_t)64 << 20))) && (uint64_t)W * g.P1 * seg_cap <= l1_items &&
                         seg_cap <= 0xFFFFFFFFULL /* the kernel's segment arithmetic is 32 x 32 -> 64 bits */;
        const uint64_t seg_slots = seg ? (uint64_t)W * seg_cap : 0;                // slots of one bucket
        uint64_t items = 0;
        unsigned long long ovf_l1 = 0;
        HIPCHK(c, hipMemsetAsync(spill_n, 0, 2 * sizeof(unsigned long long), c->stream));          // spill_n, ovf_n
        if (seg) {
            items = est_items;                                                    // the exact number is not needed (and not known)
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, items);
                hipLaunchKernelGGL(k_p1v2_scatter<true>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (const uint64_t*)nullptr, l1_buf,
                                   seg_cap, ovf_buf, ovf_n, ovf_cap);
            }
            HIPCHK(c, hipMemcpyAsync(&ovf_l1, ovf_n, sizeof ovf_l1, hipMemcpyDeviceToHost, c->stream));      // read at the next synchronisation
            if (g_trace) fprintf(stderr, "[katgpu] partition round (segmented level 1): %zu starts, ~%llu items, %llu k-mers per segment (arena %.1f GB)\n", m, (unsigned long long)items, (unsigned long long)seg_cap, c->arena_bytes / 1e9);
        } else {
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1, m);
                hipLaunchKernelGGL(k_p1v2_count, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, hist1);
                hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            }
            HIPCHK(c, hipMemcpyAsync(&items, &l1_off[g.P1], sizeof items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (g_trace) fprintf(stderr, "[katgpu] partition round: %zu starts -> %llu items (buffer %zu items, arena %.1f GB, ratio %.3f)\n", m, (unsigned long long)items, round_items, c->arena_bytes / 1e9, items_per_start);
            if (items > round_items) {                      // denser than the prefix suggested: redo this round smaller
                if (t->d.k == 32 && !t->d.canonical) HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                items_per_start = std::min(1.0, (double)items / (double)m * 1.02);
                continue;
            }
            if (items) {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, items);
                hipLaunchKernelGGL(k_p1v2_scatter<false>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (const uint64_t*)offs, l1_buf,
                                   (uint64_t)0, (uint64_t*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
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
            auto lbeg = [&](uint32_t b) -> uint64_t { return seg ? (uint64_t)b * seg_slots : h_l1_off[b]; };
            const uint32_t tile2 = l2_tile_items(g.hb);
            auto pass_extent = [&](uint32_t b_lo, uint32_t b_hi) -> uint64_t {       // bound of what level 2 writes for these buckets, in items (either edition)
                const uint64_t nn = lbeg(b_hi) - lbeg(b_lo);
                return nn + nn / 16 + 2ULL * g.P2 * (nn / tile2 + 1) + (uint64_t)(b_hi - b_lo) * g.P2 * 32 + 64;
            };
            if (seg) HIPCHK(c, hipStreamSynchronize(c->stream));                   // ovf_l1 has arrived
            uint32_t step = pass_buckets(g.P1, (uint32_t)c->n_cu);
            auto fits = [&](uint32_t st) { for (uint32_t b = 0; b < g.P1; b += st) if (pass_extent(b, std::min(g.P1, b + st)) > l2_items) return false; return true; };
            while (step > 1 && !fits(step)) step = (step + 1) / 2;
            if (!fits(step)) {                                                      // (a single bucket beyond the buffer: direct path)
                if (seg) HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));   // the scatter tallied the all-ones key
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
                uint64_t* spill_buf = l1_buf + lbeg(b_lo);                         // this pass's part of the level-1 buffer: dead once its level 2 is through
                const uint32_t* run_len = nullptr;
                unsigned long long overflowed = ovf_total;
                const bool try_fast = try_fast0 && p2_fast_ok;
                const uint32_t grid_l2 = std::min<uint32_t>(g.b_hi - g.b_lo, W2);
                HIPCHK(c, hipMemsetAsync(spill_n, 0, sizeof(unsigned long long), c->stream));
                if (try_fast && g_p2_stamp && g.hb == 1) {         // diagnostic: cycle stamps of wave 0 of every workgroup
                    unsigned long long* d_st = nullptr;
                    HIPCHK(c, hipMalloc((void**)&d_st, 64));
                    HIPCHK(c, hipMemsetAsync(d_st, 0, 64, c->stream));
                    KG_LDS_ATTR((k_p2_fast<1, true>), sizeof(P2Lds<1>));
                    hipLaunchKernelGGL((k_p2_fast<1, true>), dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<1>), c->stream, g, l1_off, l1_buf, l2_buf,
                                       off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, d_st);
                    unsigned long long h[8];
                    HIPCHK(c, hipMemcpyAsync(h, d_st, 48, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    hipFree(d_st);
                    const double n = (double)std::max<unsigned long long>(1, h[5]);
                    fprintf(stderr, "[katgpu] k_p2_fast stamps per tile (cycles, wave 0): loads %.0f, hash+rank %.0f, scan %.0f, staging %.0f, copy-out %.0f; %llu tiles\n",
                            h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5]);
                } else
                if (try_fast) {
                    ScopedTimer tm(c, KATGPU_K_PART_L2, pass_items);
#define KG_P2F(HB) case HB: hipLaunchKernelGGL(k_p2_fast<HB>, dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<HB>), c->stream, g, l1_off, l1_buf, l2_buf, \
                                               off2, cnt2, ovf_buf, ovf_n, ovf_cap, seg_slots, (unsigned long long*)nullptr); break;
                    switch (g.hb) { KG_FOR_HB(KG_P2F) }
#undef KG_P2F
                }
                if (try_fast || (seg && b_lo == 0)) {
                    HIPCHK(c, hipMemcpyAsync(&overflowed, ovf_n, sizeof overflowed, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (seg && ovf_l1 > ovf_cap) {               // the level-1 buffer itself is incomplete: this round again, exactly
                        if (g_trace) fprintf(stderr, "[katgpu] segmented level 1: %llu k-mers beyond their segments (list holds %llu): exact level 1 from here on\n", ovf_l1, (unsigned long long)ovf_cap);
                        l1_fast_ok = false;
                        HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));   // the scatter tallied the all-ones key
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
#define KG_P2(HB) case HB: hipLaunchKernelGGL(k_p2<HB>, dim3(grid_l2), dim3(PART_BLOCK), sizeof(P2Lds<HB>), c->stream, g, l1_off, l1_buf, l2_buf, \
                                              off2, seg_slots, bend); break;
                    switch (g.hb) { KG_FOR_HB(KG_P2) }
#undef KG_P2
                }
                ovf_total = overflowed;
                const uint64_t* bucket_end = !run_len ? bend : nullptr;             // exact level 2: a bucket's runs stop short of the next bucket's
                {
                    ScopedTimer tm(c, KATGPU_K_PART_APPLY, pass_items);
                    // as many workgroups per CU as the regions' LDS footprint (and the 2048-thread limit) admits
                    const uint32_t blk = g_apply_block ? g_apply_block : (g.S <= 4096 ? 512 : 1024);
                    // second edition (batched walk, per-wave straggler queues behind the region in LDS) unless a test hook needs the first
                    const bool v2 = g_apply_v != 1 && !g_test_spill_mod && g.S % 4 == 0 && g.S >= 64 && g.S <= AP2_MAX_SLOTS && (blk == 512 ? g.S <= 4096 : true);
                    if (v2) {
                        const bool big = g.S > 8192;
                        const size_t lds2 = (size_t)g.S * 12 + (size_t)(blk / 64) * (big ? AP2_QCAP_BIG : AP2_QCAP) * 12;
                        const uint32_t per_cu2 = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds2 + 512), 2048 / blk));
                        const uint32_t grid2 = std::min<uint32_t>((g.b_hi - g.b_lo) * g.P2, W2 * per_cu2);
                        // a table that is still empty sees nothing but new keys in this round: they are claimed inside the probe rounds
                        // (INLINE_CLAIM) instead of all going through the queues; any later round loses by that (kg_partition.hpp)
                        const bool fresh = t->distinct == 0 && !g_apply_noinline;
    #define KG_APPLY2(B, KP, HB) do { \
                            if (init_round) { if constexpr (HB == 1 || HB == 2) hipLaunchKernelGGL((k_p3_apply2<B, KP, 4, 3, HB, false, true, true, AP2_QCAP, true>), dim3(grid2), dim3(B), lds2, c->stream, t->d, g, off2, \
                                                          (const uint8_t*)l2_buf, spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); } \
                            else if (fresh) hipLaunchKernelGGL((k_p3_apply2<B, KP, 4, 3, HB, false, true, true>), dim3(grid2), dim3(B), lds2, c->stream, t->d, g, off2, (const uint8_t*)l2_buf, \
                                                          spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); \
                            else hipLaunchKernelGGL((k_p3_apply2<B, KP, 4, 3, HB>), dim3(grid2), dim3(B), lds2, c->stream, t->d, g, off2, (const uint8_t*)l2_buf, \
                                                    spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); } while (0)
    #define KG_APPLY2_BIG(HB) do { \
                            if (init_round) { if constexpr (HB == 1 || HB == 2) hipLaunchKernelGGL((k_p3_apply2<1024, 5, 4, 3, HB, false, true, true, AP2_QCAP_BIG, true>), dim3(grid2), dim3(1024), lds2, c->stream, t->d, g, off2, \
                                                          (const uint8_t*)l2_buf, spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); } \
                            else if (fresh) hipLaunchKernelGGL((k_p3_apply2<1024, 5, 4, 3, HB, false, true, true, AP2_QCAP_BIG>), dim3(grid2), dim3(1024), lds2, c->stream, t->d, g, off2, (const uint8_t*)l2_buf, \
                                                          spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); \
                            else hipLaunchKernelGGL((k_p3_apply2<1024, 5, 4, 3, HB, false, false, true, AP2_QCAP_BIG>), dim3(grid2), dim3(1024), lds2, c->stream, t->d, g, off2, (const uint8_t*)l2_buf, \
                                                    spill_buf, spill_n, run_len, bucket_end, (unsigned long long*)nullptr); } while (0)
    #define KG_APPLY2_HB(HB) case HB: if (blk == 512) { if (g.S <= 2048) KG_APPLY2(512, 2, HB); else KG_APPLY2(512, 4, HB); } else if (big) KG_APPLY2_BIG(HB); else KG_APPLY2(1024, 4, HB); break;
                        switch (g.hb) { KG_FOR_HB(KG_APPLY2_HB) }
    #undef KG_APPLY2_HB
    #undef KG_APPLY2_BIG
    #undef KG_APPLY2
                    } else {
                    const size_t lds = (size_t)g.S * 12;
                    // small regions (a table created "like" a bigger one): 512-thread workgroups, four per CU instead of two
                    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds + 512), 2048 / blk));
                    const uint32_t grid = std::min<uint32_t>((g.b_hi - g.b_lo) * g.P2, W2 * per_cu);
    #define KG_APPLY1(B, SPT, HOOKED) hipLaunchKernelGGL((k_p3_apply<B, SPT, 4, HOOKED>), dim3(grid), dim3(B), lds, c->stream, t->d, g, off2, (const uint8_t*)l2_buf, spill_buf, spill_n, g_test_spill_mod, run_len, bucket_end)
    #define KG_APPLY(B, SPT) do { if (g_test_spill_mod) KG_APPLY1(B, SPT, true); else KG_APPLY1(B, SPT, false); } while (0)
                    const uint32_t spt = (g.S + blk - 1) / blk;                      // region slots each lane carries while prefetching
                    if (blk == 512) { if (spt <= 8) KG_APPLY(512, 8); else if (spt <= 16) KG_APPLY(512, 16); else KG_APPLY(512, 24); }
                    else { if (spt <= 4) KG_APPLY(1024, 4); else if (spt <= 8) KG_APPLY(1024, 8); else KG_APPLY(1024, 12); }
    #undef KG_APPLY
    #undef KG_APPLY1
                    }
                }
                HIPCHK(c, hipGetLastError());
                unsigned long long spilled = 0;
                HIPCHK(c, hipMemcpyAsync(&spilled, spill_n, sizeof spilled, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (spilled) lists.push_back({spill_buf, spilled});
            }
            if (redo_round) continue;
            if (init_round) t->lazy = false;                                       // every region has been written
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
                    hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(c, l.second, 256, 6)), dim3(256), 0, c->stream, t->d, l.first, (uint64_t)l.second);
                }
            }
        }
        pos += m;
    }
    *done = pos;
    return refresh_counters(t);
}

// ------------------------------------------------------------------ super-k-mer counter (kg_superkmer.hpp) ----
// The partitioned counter for tables with minimizer regions: rounds of  S1 (items of up to 8 consecutive k-mers, sorted by the
// level-1 digit into per-workgroup segments; exact count + scan + scatter when a segment list overflows or the round is small)
// -> S2 (exact level 2) -> S3 (apply).  Same contract as count_partitioned.
static int count_superkmer(katgpu_table* t, const uint8_t* dev_bases, size_t n, size_t* done) {
    TOUCH(t);
    katgpu_ctx* c = t->ctx;
    const uint32_t k = t->d.k;
    const size_t n_starts = n - k + 1;
    *done = 0;
    c->arena_borrowed = false;
    if (!g_test_round_items && !t->disable_grow && t->d.cap < n_starts / 16) {
        uint64_t nc = t->d.cap; while (nc < n_starts / 16) nc *= 2;
        int grc = regrow(t, nc);
        if (grc) return grc;
    }
    const uint32_t W = (uint32_t)c->n_cu * 2;                                   // level-1 workgroups: two per CU (68 KB of LDS each)
    const uint32_t W2 = (uint32_t)c->n_cu;
    const size_t tile_starts = P1_TILE_STARTS;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_s2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(S2Lds)));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_apply<1024, 4, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_apply<1024, 4, 8, 3, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_apply<512, 4, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_apply<512, 2, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    // ---- arena: [hist1 | offs | l1_off | off2 | cnt2 | kmers | spill_n, ovf_n | L1 items | L2 items | overflow items]; 16-byte items ----
    constexpr size_t SEG_PAD = 16;
    const size_t fixed_items = (size_t)W * MAX_PARTS * SEG_PAD;
    const size_t small_bytes = align_up((size_t)W * MAX_PARTS * 4, 256) + align_up((size_t)W * MAX_PARTS * 8, 256) + align_up((MAX_PARTS + 1) * 8, 256) +
                               align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256) + align_up((size_t)MAX_PARTS * MAX_PARTS * 4, 256) + align_up((size_t)W * 8, 256) + 256 +
                               (2 * fixed_items + fixed_items / 16 + 8192) * 16;
    size_t want_starts = n_starts;
    if (g_test_round_items) want_starts = std::min<size_t>(want_starts, g_test_round_items);
    const size_t want_bytes = small_bytes + 8 * want_starts + ((size_t)1 << 20);                      // ~0.2 items per start x 16 B x (level 1 + level 2 + slack)
    if (c->arena_bytes < want_bytes) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        free_b += c->arena_bytes;
        const size_t bytes = std::min<size_t>(want_bytes, (size_t)(g_arena_fraction * (double)free_b));
        if (bytes > c->arena_bytes + c->arena_bytes / 2 || c->arena_bytes < small_bytes + 8 * std::min<size_t>(want_starts, (size_t)64 << 20)) {
            if (c->arena) { HIPCHK(c, hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
            if (!g_test_round_items && bytes < small_bytes + 8 * ((size_t)1 << 20)) return KATGPU_OK;
            if (hipMalloc((void**)&c->arena, bytes) != hipSuccess) { (void)hipGetLastError(); c->arena = nullptr; return KATGPU_OK; }
            c->arena_bytes = bytes;
        }
    }
    struct Busy { katgpu_ctx* c; explicit Busy(katgpu_ctx* c_) : c(c_) { c->arena_busy = true; } ~Busy() { c->arena_busy = false; } } busy(c);
    uint8_t* a = c->arena;
    uint32_t* hist1 = (uint32_t*)a;               a += align_up((size_t)W * MAX_PARTS * 4, 256);
    uint64_t* offs = (uint64_t*)a;                a += align_up((size_t)W * MAX_PARTS * 8, 256);
    uint64_t* l1_off = (uint64_t*)a;              a += align_up((MAX_PARTS + 1) * 8, 256);
    uint64_t* off2 = (uint64_t*)a;                a += align_up(((size_t)MAX_PARTS * MAX_PARTS + 1) * 8, 256);
    uint32_t* cnt2 = (uint32_t*)a;                a += align_up((size_t)MAX_PARTS * MAX_PARTS * 4, 256);
    unsigned long long* kmers = (unsigned long long*)a;   a += align_up((size_t)W * 8, 256);
    unsigned long long* spill_n = (unsigned long long*)a;
    unsigned long long* ovf_n = spill_n + 1;      a += 256;
    // what is left: the level-1 buffer, the level-2 buffer (it mirrors the level-1 layout) and the overflow list (1/32 of a buffer)
    const size_t item_room = (size_t)(c->arena + c->arena_bytes - a) / 16;
    const size_t l1_cap = (size_t)((double)item_room * 32.0 / 65.0);
    const size_t ovf_cap_items = g_test_p2_ovf_cap ? (size_t)g_test_p2_ovf_cap : l1_cap / 32;
    if (l1_cap < fixed_items + 4096) return KATGPU_OK;
    u32x4* l1_items = (u32x4*)a;
    u32x4* l2_items = l1_items + l1_cap;
    u32x4* ovf_items = l2_items + l1_cap;
    uint64_t* spill_buf = (uint64_t*)l1_items;                                     // the apply's spilled k-mers: level 1 is dead by then
    const uint64_t spill_cap = (uint64_t)l1_cap * 2;
    bool seg_ok = g_l1_fast != 0;

    double items_per_start = 0.0;                                                  // measured on a prefix before the first round
    size_t pos = 0;
    while (pos < n_starts) {
        int rc = refresh_counters(t);
        if (rc) return rc;
        if ((double)t->distinct > 0.6 * (double)t->d.cap) {
            bool lost = false;
            rc = grow_beside_arena(t, 0, t->d.cap * 2, KeyLists(), &lost);
            if (rc) return rc;
            if (lost) break;
        }
        PartGeom g;
        if (!part_geometry(t->d, &g) || !t->d.mz || g.S % 4 || g.S < 64 || g.S > 8192) break;       // not a table for this counter: direct path
        if (items_per_start == 0.0) {
            const size_t probe_m = std::max<size_t>(tile_starts, std::min<size_t>(n_starts - pos, (size_t)64 << 20) / tile_starts * tile_starts);
            const uint64_t pt = (std::min(probe_m, n_starts - pos) + tile_starts - 1) / tile_starts, ptw = (pt + W - 1) / W;
            const size_t pm = std::min(probe_m, n_starts - pos);
            HIPCHK(c, hipMemsetAsync(kmers, 0, (size_t)W * 8, c->stream));
            hipLaunchKernelGGL(k_s1<0>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, dev_bases + pos, (uint64_t)(pm + k - 1), pt, ptw, hist1, kmers, (const uint64_t*)nullptr,
                               (u32x4*)nullptr, (uint64_t)0, (u32x4*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
            hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            uint64_t probe_items = 0;
            HIPCHK(c, hipMemcpyAsync(&probe_items, &l1_off[g.P1], sizeof probe_items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (t->d.k == 32 && !t->d.canonical) HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));   // the real pass tallies again
            items_per_start = std::max(0.01, (double)probe_items / (double)pm) * 1.03;
        }
        const size_t round_starts = g_test_round_items ? (size_t)g_test_round_items
                                                       : (size_t)std::max(1.0, ((double)l1_cap - (double)fixed_items) / (items_per_start * (1.0 + 1.0 / 24.0)));
        size_t m = n_starts - pos;
        if (m > round_starts) {
            const size_t rounds_left = (m + round_starts - 1) / round_starts;          // balance the remaining rounds
            m = (m + rounds_left - 1) / rounds_left;
            m += tile_starts - m % tile_starts;                                        // whole tiles, keeps the next round 16-byte aligned
            m = std::min(m, n_starts - pos);
        }
        const size_t nb = m + k - 1;
        const uint8_t* p = dev_bases + pos;
        t->count_bound = 0xFFFFFFFFULL;
        const uint64_t n_tiles = (m + tile_starts - 1) / tile_starts;
        const uint64_t tiles_per_wg = (n_tiles + W - 1) / W;
        const uint64_t est_items = (uint64_t)((double)m * items_per_start) + 1;
        uint64_t seg_cap = est_items / ((uint64_t)W * g.P1);
        seg_cap += seg_cap / 24 + SEG_PAD;
        if (g_test_l1_cpb) seg_cap = std::min<uint64_t>(seg_cap, g_test_l1_cpb);
        const bool seg = seg_ok && (g_l1_fast == 2 || est_items >= ((uint64_t)16 << 20)) && (uint64_t)W * g.P1 * seg_cap <= l1_cap;
        const uint64_t seg_slots = seg ? (uint64_t)W * seg_cap : 0;
        uint64_t items = 0;
        unsigned long long ovf_l1 = 0;
        HIPCHK(c, hipMemsetAsync(spill_n, 0, 2 * sizeof(unsigned long long), c->stream));
        if (seg) {
            items = est_items;
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, m);
                hipLaunchKernelGGL(k_s1<2>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (uint32_t*)nullptr, (unsigned long long*)nullptr,
                                   (const uint64_t*)nullptr, l1_items, seg_cap, ovf_items, ovf_n, (uint64_t)ovf_cap_items);
            }
            HIPCHK(c, hipMemcpyAsync(&ovf_l1, ovf_n, sizeof ovf_l1, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (g_trace) fprintf(stderr, "[katgpu] super-k-mer round (segmented level 1): %zu starts, ~%llu items (%.3f per start), %llu per segment, %llu overflowed (arena %.1f GB)\n", m,
                                 (unsigned long long)items, items_per_start, (unsigned long long)seg_cap, ovf_l1, c->arena_bytes / 1e9);
            if (ovf_l1 > ovf_cap_items) {                   // level 1 is incomplete: this round again, exactly
                seg_ok = false;
                HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));   // the scatter tallied the all-ones key
                continue;
            }
        } else {
            {
                ScopedTimer tm(c, KATGPU_K_PART_L1, m);
                hipLaunchKernelGGL(k_s1<0>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, hist1, kmers, (const uint64_t*)nullptr,
                                   (u32x4*)nullptr, (uint64_t)0, (u32x4*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
                hipLaunchKernelGGL(k_p1_scan, dim3(1), dim3(PART_BLOCK), 0, c->stream, g, W, hist1, offs, l1_off);
            }
            HIPCHK(c, hipMemcpyAsync(&items, &l1_off[g.P1], sizeof items, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (g_trace) fprintf(stderr, "[katgpu] super-k-mer round (exact level 1): %zu starts -> %llu items (buffer %zu)\n", m, (unsigned long long)items, l1_cap);
            if (items > l1_cap) {                           // denser than the prefix suggested: this round again, smaller
                if (t->d.k == 32 && !t->d.canonical) HIPCHK(c, hipMemcpyAsync(&t->d.ctrs[CTR_ONES], &t->ones, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                items_per_start = (double)items / (double)m * 1.05;
                if (g_test_round_items) return fail(c, KATGPU_ERR_NOMEM, "super-k-mer round of %zu starts does not fit the arena", m);
                continue;
            }
            if (items) {
                ScopedTimer tm(c, KATGPU_K_PART_L1S, m);
                hipLaunchKernelGGL(k_s1<1>, dim3(W), dim3(P1_BLOCK), 0, c->stream, t->d, g, p, (uint64_t)nb, n_tiles, tiles_per_wg, (uint32_t*)nullptr, (unsigned long long*)nullptr,
                                   (const uint64_t*)offs, l1_items, (uint64_t)0, (u32x4*)nullptr, (unsigned long long*)nullptr, (uint64_t)0);
            }
        }
        if (items) {
            {
                ScopedTimer tm(c, KATGPU_K_PART_L2, items);
                hipLaunchKernelGGL(k_s2, dim3(std::min<uint32_t>(g.P1, W2)), dim3(PART_BLOCK), sizeof(S2Lds), c->stream, g, (const uint64_t*)l1_off, seg_slots, (const u32x4*)l1_items, l2_items, off2, cnt2);
            }
            {
                ScopedTimer tm(c, KATGPU_K_PART_APPLY, items);
                const uint32_t blk = g.S <= 4096 ? 512 : 1024;
                const size_t lds2 = (size_t)g.S * 12 + (size_t)(blk / 64) * AP2_QCAP * 12;
                const uint32_t per_cu2 = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds2 + 512), 2048 / blk));
                const uint32_t grid2 = std::min<uint32_t>(g.R, W2 * per_cu2);
                const bool fresh = t->distinct == 0 && !g_apply_noinline;
#define KG_S3(B, KP, ...) hipLaunchKernelGGL((k_s3_apply<B, KP, 8, 3, __VA_ARGS__>), dim3(grid2), dim3(B), lds2, c->stream, t->d, g, (const uint64_t*)off2, (const u32x4*)l2_items, spill_buf, spill_n, (const uint32_t*)cnt2, (const uint64_t*)nullptr, (unsigned long long*)nullptr)
                if (blk == 512) { if (g.S <= 2048) KG_S3(512, 2, false, false, true); else KG_S3(512, 4, false, false, true); }
                else if (fresh) KG_S3(1024, 4, false, true, true);
                else KG_S3(1024, 4, false, false, true);
#undef KG_S3
            }
            HIPCHK(c, hipGetLastError());
            unsigned long long spilled = 0;
            HIPCHK(c, hipMemcpyAsync(&spilled, spill_n, sizeof spilled, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (spilled > spill_cap) return fail(c, KATGPU_ERR_NOMEM, "%llu k-mers found their regions full in one round (the list holds %llu): size the table (-H) for the input", spilled, (unsigned long long)spill_cap);
            if (ovf_l1) {                               // level 1's overflow items through the direct path
                rc = ensure_room(t, (uint64_t)ovf_l1 * SK_MAXN);
                if (rc == KATGPU_ERR_NOMEM) rc = KATGPU_OK;              // (table_park takes care of a region that fills)
                if (rc) return rc;
                ScopedTimer tm(c, KATGPU_K_COUNT, ovf_l1);
                hipLaunchKernelGGL(k_insert_items, dim3(grid_for(c, ovf_l1, 256, 6)), dim3(256), 0, c->stream, t->d, (const u32x4*)ovf_items, (uint64_t)ovf_l1);
            }
            if (spilled) {
                bool lost = false;
                rc = grow_beside_arena(t, spilled, 0, KeyLists{{spill_buf, spilled}}, &lost);
                if (rc) return rc;
                if (lost) { pos += m; break; }
                ScopedTimer tm(c, KATGPU_K_COUNT, spilled);
                hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(c, spilled, 256, 6)), dim3(256), 0, c->stream, t->d, (const uint64_t*)spill_buf, (uint64_t)spilled);
            }
        }
        pos += m;
    }
    *done = pos;
    return refresh_counters(t);
}

// Count a resident base stream.  The stream is cut into sub-batches so that "distinct + sub-batch starts" stays under
// the load limit (the table can then never fill in the middle of a launch); consecutive sub-batches overlap by k-1.
static int count_resident(katgpu_table* t, const uint8_t* dev_bases, size_t n) {
    katgpu_ctx* c = t->ctx;
    const uint32_t k = t->d.k;
    if (n < k) return KATGPU_OK;
    size_t pos = 0;
    const size_t n_starts = n - k + 1;
    // Large, aligned inputs go through the partitioned counter (no global atomic per k-mer); whatever it leaves (nothing,
    // normally) and everything small goes through the direct kernel below.
    while (!t->d.keys_b && n_starts - pos >= std::max<uint64_t>(g_part_min_starts, 1) && (reinterpret_cast<uintptr_t>(dev_bases + pos) & 15) == 0) {
        size_t done = 0;                        // returns early (done < remaining) when a table growth cost it the arena
        int prc = t->d.mz ? count_superkmer(t, dev_bases + pos, n - pos, &done) : count_partitioned(t, dev_bases + pos, n - pos, &done);
        if (prc) return prc;
        if (!done) break;
        pos += done;
    }
    while (pos < n_starts) {
        TOUCH(t);                               // (a lazy table that the partitioned counter did not take)
        int rc = refresh_counters(t);
        if (rc) return rc;
        // largest batch that provably fits; if even a minimal one does not, grow first
        uint64_t room = (uint64_t)(load_limit(t->d) * (double)t->d.cap) > t->distinct ? (uint64_t)(load_limit(t->d) * (double)t->d.cap) - t->distinct : 0;
        uint64_t want = std::min<uint64_t>(n_starts - pos, (uint64_t)CHUNK_STARTS * 65536);   // <= 266 M starts per launch
        if (g_test_max_starts) want = std::min<uint64_t>(want, g_test_max_starts);
        // As the table fills, launches shrink to the remaining room (each adds far fewer distinct k-mers than window
        // starts on real coverage, so the room shrinks slowly); only when the room is down to 1/64 of the table do we grow.
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 64, CHUNK_STARTS))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 2, CHUNK_STARTS)));
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

static int ensure_staging(katgpu_ctx* c) {
    if (!c->stage_bytes) {
        const size_t bytes = (size_t)64 << 20;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipHostMalloc((void**)&c->pinned[i], bytes, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&c->pin_free[i], hipEventDisableTiming));
        }
        c->stage_bytes = bytes;
    }
    if (!c->ring_bytes) {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipMalloc((void**)&c->ring[i], g_ring_bytes);
            if (e != hipSuccess && c->arena && !c->arena_borrowed && !c->arena_busy) {       // the cached arena holds most of the free HBM: give it back
                (void)hipGetLastError();
                hipFree(c->arena); c->arena = nullptr; c->arena_bytes = 0;
                e = hipMalloc((void**)&c->ring[i], g_ring_bytes);
            }
            if (e != hipSuccess) { for (int j = 0; j < i; ++j) { hipFree(c->ring[j]); c->ring[j] = nullptr; } HIPCHK(c, e); }
        }
        c->ring_bytes = g_ring_bytes;
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
            int rc = worker_rc ? worker_rc : count_resident(t, c->ring[j.ring], j.n);
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
    int begin() {
        int rc = ensure_staging(c); if (rc) return rc;
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
        const uint32_t want = t->d.k - 1;
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
    int rc = f.begin(); if (rc) return rc;
    rc = f.push(bases, n); if (rc) return rc;
    return f.finish();
}

extern "C" int katgpu_count_files(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p) {
    if (!t || !paths) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    t->carry_n = 0;
    HostFeeder f(t);
    int rc = f.begin(); if (rc) return rc;
    // the group's files -> one base stream (kg_ingest.hpp: thread team for large plain files, concurrent readers for gzip & co.)
    std::string err;
    rc = kg::stream_group(paths, n_paths, trim5p, t->d.k, [&](const uint8_t* p, size_t n) { return f.push(p, n); }, &err);
    if (rc) return err.empty() ? rc : fail(c, rc, "%s", err.c_str());
    return f.finish();
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
    katgpu_table* t = nullptr;
    int rc = katgpu_table_create(c, k, canonical, size_hint, disable_grow, &t);
    if (rc) return rc;
    rc = katgpu_count_files(t, paths, n_paths, trim5p);
    if (rc) { katgpu_table_free(t); return rc; }
    *out = t;
    return KATGPU_OK;
}

extern "C" int katgpu_table_stats(katgpu_table* t, uint64_t* distinct, uint64_t* total, uint64_t* capacity) {
    TOUCH(t);
    if (!t) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    if (distinct) *distinct = t->distinct;
    if (capacity) *capacity = t->d.cap;
    if (total) {
        uint64_t* scratch = &t->d.ctrs[CTR_SCRATCH];
        HIPCHK(c, hipMemsetAsync(scratch, 0, sizeof(uint64_t), c->stream));
        hipLaunchKernelGGL(k_total, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, scratch);
        uint64_t s = 0;
        HIPCHK(c, hipMemcpyAsync(&s, scratch, sizeof s, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *total = s + t->ones;
    }
    return KATGPU_OK;
}

extern "C" int katgpu_table_get(katgpu_table* t, const uint64_t* keys, size_t n, int canonicalise, uint64_t* counts) {
    TOUCH(t);
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
    hipLaunchKernelGGL(k_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t->d, t->n_ovf, dk, (uint64_t)n, canonicalise, dc);
    hipMemcpyAsync(counts, dc, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(dk); hipFree(dc);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

static int launch_profile(katgpu_table* t, const uint8_t* dev_bases, size_t n, int canonicalise, uint64_t* dev_counts) {
    katgpu_ctx* c = t->ctx;
    const bool wide = t->d.keys_b != nullptr;
    const uint64_t n_out = n - t->d.k + 1;
    const uint64_t per_chunk = wide ? WIDE_CHUNK_STARTS : CHUNK_STARTS;
    const uint64_t n_chunks = (n_out + per_chunk - 1) / per_chunk;
    const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * 8);
    const bool aligned = (reinterpret_cast<uintptr_t>(dev_bases) & 15) == 0 && (reinterpret_cast<uintptr_t>(dev_counts) & 15) == 0;
    ScopedTimer tm(c, KATGPU_K_PROFILE, n_out);
    if (wide && aligned)
        hipLaunchKernelGGL(k_profile_w<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else if (wide)
        hipLaunchKernelGGL(k_profile_w<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else if (aligned)
        hipLaunchKernelGGL(k_profile<true>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    else
        hipLaunchKernelGGL(k_profile<false>, dim3(grid), dim3(COUNT_BLOCK), 0, c->stream, t->d, t->n_ovf, canonicalise, dev_bases, (uint64_t)n, n_chunks, dev_counts);
    HIPCHK(c, hipGetLastError());
    return KATGPU_OK;
}

extern "C" int katgpu_table_profile_device(katgpu_table* t, const uint8_t* dev_bases, size_t n, int canonicalise, uint64_t* dev_counts) {
    TOUCH(t);
    if (!t || (n && (!dev_bases || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    if (n < t->d.k) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    return launch_profile(t, dev_bases, n, canonicalise, dev_counts);
}

// Host form: the sequence goes through the device in batches of PROFILE_BATCH window starts (each batch re-sends the
// k-1 bases it shares with the next one), so any length fits next to the table.
extern "C" int katgpu_table_profile_host(katgpu_table* t, const char* bases, size_t n, int canonicalise, uint64_t* counts) {
    TOUCH(t);
    if (!t || (n && (!bases || !counts))) return KATGPU_ERR_INVALID_ARG;
    const uint32_t k = t->d.k;
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
    TOUCH(t);
    if (!t || !sizes || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemsetAsync(d, 0, n_parts * 8, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->d.cap);
        if (t->d.keys_b)
            hipLaunchKernelGGL(k_partition_w<0>, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, n_parts, d, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr);
        else
            hipLaunchKernelGGL(k_partition<0>, dim3(grid_for(c, t->d.cap + 1, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, n_parts, d, (uint64_t*)nullptr, (uint64_t*)nullptr);
    }
    hipMemcpyAsync(sizes, d, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_partition(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets, uint64_t* dev_keys, uint64_t* dev_counts) {
    TOUCH(t);
    if (!t || !offsets || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_partition");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemcpyAsync(d, offsets, n_parts * 8, hipMemcpyHostToDevice, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->d.cap);
        hipLaunchKernelGGL(k_partition<1>, dim3(grid_for(c, t->d.cap + 1, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, n_parts, d, dev_keys, dev_counts);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_export(katgpu_table* t, uint64_t* keys, uint64_t* counts, size_t cap, size_t* n_out) {
    TOUCH(t);
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
    TOUCH(t);
    if (!t || (n && (!dev_keys || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "katgpu_table_merge_device");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    size_t pos = 0;
    while (pos < n) {
        int rc = refresh_counters(t); if (rc) return rc;
        uint64_t room = (uint64_t)(load_limit(t->d) * (double)t->d.cap) > t->distinct ? (uint64_t)(load_limit(t->d) * (double)t->d.cap) - t->distinct : 0;
        uint64_t want = n - pos;
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 8, 1024))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 2, 1024)));
            if (rc) return rc;
            continue;
        }
        uint64_t take = std::min(want, room);
        t->count_bound = 0xFFFFFFFFULL;          // merged amounts are arbitrary: the next k_count launch sweeps first
        {
            ScopedTimer tm(c, KATGPU_K_MERGE, take);
            hipLaunchKernelGGL(k_merge, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->d, dev_keys + pos, dev_counts + pos, take);
        }
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_host(katgpu_table* t, const uint64_t* keys, const uint64_t* counts, size_t n) {
    TOUCH(t);
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
    TOUCH(t);
    if (!t || !n_out) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->d.keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_export_wide is for k > 32 tables (k = %u): use katgpu_table_export", t->d.k);
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
    hipLaunchKernelGGL(k_export_w, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, d, d + n, d + 2 * n, cursor);
    hipMemcpyAsync(keys_hi, d, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(keys_lo, d + n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipMemcpyAsync(counts, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_partition_wide(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets, uint64_t* dev_hi, uint64_t* dev_lo, uint64_t* dev_counts) {
    TOUCH(t);
    if (!t || !offsets || n_parts == 0 || n_parts > 4096) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->d.keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_partition_wide is for k > 32 tables (k = %u): use katgpu_table_partition", t->d.k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    if (t->distinct && (!dev_hi || !dev_lo || !dev_counts)) return KATGPU_ERR_INVALID_ARG;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n_parts * 8));
    hipMemcpyAsync(d, offsets, n_parts * 8, hipMemcpyHostToDevice, c->stream);
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->d.cap);
        hipLaunchKernelGGL(k_partition_w<1>, dim3(grid_for(c, t->d.cap, 256, 8)), dim3(256), 0, c->stream, t->d, t->n_ovf, n_parts, d, dev_hi, dev_lo, dev_counts);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_merge_device_wide(katgpu_table* t, const uint64_t* dev_hi, const uint64_t* dev_lo, const uint64_t* dev_counts, size_t n) {
    TOUCH(t);
    if (!t || (n && (!dev_hi || !dev_lo || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->d.keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_merge_device_wide is for k > 32 tables (k = %u): use katgpu_table_merge_device", t->d.k);
    HIPCHK(c, hipSetDevice(c->device));
    size_t pos = 0;
    while (pos < n) {
        int rc = refresh_counters(t); if (rc) return rc;
        const uint64_t room = (uint64_t)(load_limit(t->d) * (double)t->d.cap) > t->distinct ? (uint64_t)(load_limit(t->d) * (double)t->d.cap) - t->distinct : 0;
        const uint64_t want = n - pos;
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 8, 1024))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 2, 1024)));
            if (rc) return rc;
            continue;
        }
        const uint64_t take = std::min(want, room);
        ScopedTimer tm(c, KATGPU_K_MERGE, take);
        hipLaunchKernelGGL(k_merge_w, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->d, dev_hi + pos, dev_lo + pos, dev_counts + pos, (uint64_t)take);
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_host_wide(katgpu_table* t, const uint64_t* keys_hi, const uint64_t* keys_lo, const uint64_t* counts, size_t n) {
    TOUCH(t);
    if (!t || (n && (!keys_hi || !keys_lo || !counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->d.keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_merge_host_wide is for k > 32 tables (k = %u): use katgpu_table_merge_host", t->d.k);
    if (!n) return KATGPU_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t k = t->d.k;
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
    TOUCH(t);
    if (!t || (n && (!keys_hi || !keys_lo || !counts))) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    if (!t->d.keys_b) return fail(c, KATGPU_ERR_K, "katgpu_table_get_wide is for k > 32 tables (k = %u): use katgpu_table_get", t->d.k);
    if (!n) return KATGPU_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    uint64_t* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 3 * n * 8));
    hipMemcpyAsync(d, keys_hi, n * 8, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d + n, keys_lo, n * 8, hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_get_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t->d, t->n_ovf, d, d + n, (uint64_t)n, canonicalise, d + 2 * n);
    hipMemcpyAsync(counts, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

// ------------------------------------------------------------------ region-ordered exchange -----------

extern "C" int katgpu_place_keys(uint32_t k, uint32_t p1, uint32_t l2, const uint64_t* keys, size_t n, uint32_t* d1, uint32_t* d2, uint64_t* rem,
                                 uint64_t* back, uint32_t* rem_bits) {
    if (k < 1 || k > 32 || p1 < 1 || p1 > MAX_PARTS || l2 > 10 || (n && (!keys || !d1 || !d2 || !rem || !back))) return KATGPU_ERR_INVALID_ARG;
    const Place pl = place_make(k, p1, place_n1(k, p1), l2);
    if (rem_bits) *rem_bits = pl.rb;
    for (size_t i = 0; i < n; ++i) {
        const Placed h = place_hash(keys[i], pl);
        d1[i] = h.d1; d2[i] = h.d2; rem[i] = h.rem;
        back[i] = place_key(place_base1(h.d1, pl.n, pl.p1), (pl.rb < 64 ? (uint64_t)h.d2 << pl.rb : 0ULL) | h.rem, pl);
    }
    return KATGPU_OK;
}

extern "C" int katgpu_table_geometry(const katgpu_table* t, katgpu_geometry* g) {
    if (!t || !g) return KATGPU_ERR_INVALID_ARG;
    if (t->d.keys_b) return fail(t->ctx, KATGPU_ERR_K, "the multi-GPU exchange is not available for k > 32 (k = %u)", t->d.k);
    g->k = t->d.k; g->canonical = t->d.canonical; g->n_regions = t->d.n_regions; g->region_slots = t->d.region_slots;
    g->p1 = t->d.p1 | (t->d.mz << 31); g->p2 = t->d.p2; g->capacity = t->d.cap;      // bit 31 of p1: minimizer regions -- "same grid" includes the region function
    return KATGPU_OK;
}

extern "C" int katgpu_table_extract_sizes(katgpu_table* t, uint32_t n_parts, uint32_t* dev_region_counts, uint64_t* part_sizes) {
    TOUCH(t);
    if (!t || !dev_region_counts || !part_sizes || n_parts == 0 || n_parts > MAX_EXCHANGE_PARTS) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d_tot = nullptr;
    HIPCHK(c, hipMalloc(&d_tot, n_parts * 8));
    {
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->d.cap);
        hipLaunchKernelGGL(k_extract_count, dim3(std::min<uint32_t>(t->d.n_regions, (uint32_t)c->n_cu * 8)), dim3(EXTRACT_BLOCK), 0, c->stream, t->d, n_parts, dev_region_counts);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, (const uint32_t*)dev_region_counts, t->d.n_regions, (uint64_t)t->d.n_regions,
                           (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, 0, d_tot);
    }
    hipMemcpyAsync(part_sizes, d_tot, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d_tot);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_table_extract(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint64_t* dev_keys, uint32_t* dev_counts,
                                    uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big) {
    TOUCH(t);
    if (!t || !dev_region_counts || !dev_keys || !dev_counts || !n_big || n_parts == 0 || n_parts > MAX_EXCHANGE_PARTS || (big_cap && (!big_keys || !big_counts)))
        return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    const uint32_t R = t->d.n_regions;
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
        ScopedTimer tm(c, KATGPU_K_PARTITION, t->d.cap);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, dev_region_counts, R, (uint64_t)R, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, 0, d_tot);
        hipMemcpyAsync(tot.data(), d_tot, n_parts * 8, hipMemcpyDeviceToHost, c->stream);
        e = hipStreamSynchronize(c->stream);
        uint64_t run = 0;
        for (uint32_t p = 0; p < n_parts; ++p) { base[p] = run; run += tot[p]; }
        hipMemcpyAsync(d_base, base.data(), n_parts * 8, hipMemcpyHostToDevice, c->stream);
        hipMemsetAsync(d_bign, 0, 8, c->stream);
        hipLaunchKernelGGL(k_rows_scan, dim3(n_parts), dim3(1024), 0, c->stream, dev_region_counts, R, (uint64_t)R, (const uint64_t*)d_base, d_off, (uint64_t)R, 0, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_extract_write, dim3(std::min<uint32_t>(R, (uint32_t)c->n_cu * 8)), dim3(EXTRACT_BLOCK), 0, c->stream, t->d, t->n_ovf, n_parts, (const uint64_t*)d_off,
                           dev_keys, dev_counts, d_bk, d_bc, d_bign, dev_big_cap);
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
    TOUCH(t);
    if (!t) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    DevTable& d = t->d;
    HIPCHK(c, hipMemsetAsync(d.keys, 0xFF, d.cap * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(d.counts, 0, d.cap * sizeof(uint32_t), c->stream));
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
        uint64_t room = (uint64_t)(load_limit(t->d) * (double)t->d.cap) > t->distinct ? (uint64_t)(load_limit(t->d) * (double)t->d.cap) - t->distinct : 0;
        uint64_t want = n - pos;
        if (room < std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 8, 1024))) {
            rc = ensure_room(t, std::min<uint64_t>(want, std::max<uint64_t>(t->d.cap / 2, 1024)));
            if (rc) return rc;
            continue;
        }
        const uint64_t take = std::min(want, room);
        t->count_bound = 0xFFFFFFFFULL;
        ScopedTimer tm(c, KATGPU_K_MERGE, take);
        hipLaunchKernelGGL(k_merge32, dim3(grid_for(c, take, 256, 8)), dim3(256), 0, c->stream, t->d, dev_keys + pos, dev_counts + pos, (uint64_t)take);
        pos += take;
    }
    return refresh_counters(t);
}

extern "C" int katgpu_table_merge_device32(katgpu_table* t, const uint64_t* dev_keys, const uint32_t* dev_counts, size_t n) {
    TOUCH(t);
    if (!t || (n && (!dev_keys || !dev_counts))) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    return merge_direct32(t, dev_keys, dev_counts, n);
}

static const bool g_no_merge_apply = hook("KATGPU_NO_MERGE_APPLY") != nullptr;    // A/B switch + tests: every source through the direct path

extern "C" int katgpu_table_merge_regions(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const katgpu_merge_source* src) {
    TOUCH(t);
    if (!t || !src || n_src == 0 || g_lo > g_hi) return KATGPU_ERR_INVALID_ARG;
    NARROW_ONLY(t, "the multi-GPU exchange");
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_apply<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    // sources ordered by this table's regions go through LDS; the rest (another grid, or the table has changed its grid) directly
    std::vector<uint32_t> aligned, direct;
    for (uint32_t i = 0; i < n_src; ++i) {
        if (src[i].n_records == 0) continue;
        if (!src[i].dev_keys || !src[i].dev_counts) return KATGPU_ERR_INVALID_ARG;
        const bool ok = !g_no_merge_apply && src[i].dev_region_counts && src[i].p1 == (t->d.p1 | (t->d.mz << 31)) && src[i].p2 == t->d.p2 && g_hi <= t->d.n_regions &&
                        (size_t)t->d.region_slots * 12 <= 150 * 1024;
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
        uint64_t records = 0;
        for (uint32_t q = 0; q < na; ++q) {
            const katgpu_merge_source& s = src[aligned[a0 + q]];
            hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, c->stream, s.dev_region_counts, n_reg, (uint64_t)n_reg, (const uint64_t*)nullptr,
                               d_off + (size_t)q * (n_reg + 1), (uint64_t)(n_reg + 1), 1, (unsigned long long*)nullptr);
            ms.s[q] = MergeSrc{s.dev_keys, s.dev_counts, d_off + (size_t)q * (n_reg + 1)};
            records += s.n_records;
        }
        hipMemsetAsync(d_ndef, 0, 8, c->stream);
        t->count_bound = 0xFFFFFFFFULL;
        {
            ScopedTimer tm(c, KATGPU_K_MERGE, records);
            const size_t lds = (size_t)t->d.region_slots * 12;
            const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds + 512), 2));
            hipLaunchKernelGGL(k_merge_apply<1024>, dim3(std::min<uint32_t>(n_reg, (uint32_t)c->n_cu * per_cu)), dim3(1024), lds, c->stream, t->d, g_lo, g_hi, ms, d_def, d_ndef);
        }
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
            const uint64_t need_s = (uint64_t)(((double)t->d.region_slots + (double)max_in) / 0.7) + 1;
            if (need_s > t->d.region_slots) {
                if (t->disable_grow) rc = fail(c, KATGPU_ERR_TABLE_FULL, "Hash full");
                else rc = regrow(t, (uint64_t)t->d.n_regions * need_s);
            }
            if (rc == KATGPU_OK) {                    // one launch for all deferred regions: every region now has the room
                ScopedTimer tm(c, KATGPU_K_MERGE, max_in * ndef);
                hipLaunchKernelGGL(k_merge_deferred, dim3((unsigned)std::min<unsigned long long>(ndef, (unsigned long long)c->n_cu * 8)), dim3(256), 0, c->stream,
                                   t->d, g_lo, ms, (const uint32_t*)d_def, (uint32_t)ndef);
                if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, KATGPU_ERR_DEVICE, "deferred merge");
                else rc = refresh_counters(t);
            }
        }
        hipFree(tmp);
        if (rc) return rc;
        if (ndef && (src[aligned[a0]].p1 != (t->d.p1 | (t->d.mz << 31)) || src[aligned[a0]].p2 != t->d.p2)) {      // the growth changed the grid: the rest goes direct
            for (size_t a = a0 + na; a < aligned.size(); ++a) direct.push_back(aligned[a]);
            break;
        }
    }
    for (uint32_t i : direct) {
        rc = merge_direct32(t, src[i].dev_keys, src[i].dev_counts, (size_t)src[i].n_records);
        if (rc) return rc;
    }
    return refresh_counters(t);
}

// ------------------------------------------------------------------ reducers --------------------------

static int reducer_grid(katgpu_ctx* c, uint64_t slots, int blocks_per_cu) { return grid_for(c, slots, 256, blocks_per_cu); }

extern "C" int katgpu_hist(katgpu_table* t, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t* out, size_t nb) {
    TOUCH(t);
    if (!t || !out || nb == 0 || inc == 0 || ceil_ < base || nb != ceil_ + 1 - base) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, nb * 8));
    hipMemsetAsync(d, 0, nb * 8, c->stream);
    const uint32_t lds_bins = (uint32_t)std::min<uint64_t>(nb, 16384);          // 64 KB of u32 -> two blocks per CU
    {
        ScopedTimer tm(c, KATGPU_K_HIST, t->d.cap);
        hipLaunchKernelGGL(k_hist, dim3(grid_for(c, t->d.cap / 4, SCAN_BLOCK, 2)), dim3(SCAN_BLOCK), lds_bins * sizeof(uint32_t), c->stream,
                           t->d, t->n_ovf, base, ceil_, inc, (uint64_t)nb, d, lds_bins);
    }
    hipMemcpyAsync(out, d, nb * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_gcp(katgpu_table* t, double cvg_scale, uint32_t cvg_bins, uint64_t* out) {
    TOUCH(t);
    if (!t || !out) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t); if (rc) return rc;
    const size_t cells = (size_t)t->d.k * ((size_t)cvg_bins + 1);
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, cells * 8));
    hipMemsetAsync(d, 0, cells * 8, c->stream);
    const size_t lds = cells * sizeof(uint32_t);
    const uint32_t use_lds = lds <= 150 * 1024 ? 1 : 0;                         // 160 KB LDS per CU
    if (use_lds && lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(t->d.keys_b ? k_gcp<true> : k_gcp<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ScopedTimer tm(c, KATGPU_K_GCP, t->d.cap);
        if (t->d.keys_b)
            hipLaunchKernelGGL(k_gcp<true>, dim3(grid_for(c, t->d.cap / 4, SCAN_BLOCK, use_lds && lds > 75 * 1024 ? 1 : 2)), dim3(SCAN_BLOCK), use_lds ? lds : 0, c->stream,
                               t->d, t->n_ovf, cvg_scale, cvg_bins, d, use_lds);
        else
            hipLaunchKernelGGL(k_gcp<false>, dim3(grid_for(c, t->d.cap / 4, SCAN_BLOCK, use_lds && lds > 75 * 1024 ? 1 : 2)), dim3(SCAN_BLOCK), use_lds ? lds : 0, c->stream,
                               t->d, t->n_ovf, cvg_scale, cvg_bins, d, use_lds);
    }
    HIPCHK(c, hipGetLastError());
    hipMemcpyAsync(out, d, cells * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KATGPU_ERR_DEVICE, "%s", hipGetErrorString(e));
    return KATGPU_OK;
}

extern "C" int katgpu_comp(katgpu_table* t1, katgpu_table* t2, int canon1, int canon2, double d1_scale, double d2_scale,
                           uint32_t d1_bins, uint32_t d2_bins, uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra) {
    TOUCH(t1);
    TOUCH(t2);
    (void)canon1;
    if (!t1 || !t2 || !main_mx || !counters || !spectra || d1_bins == 0 || d2_bins == 0) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = t1->ctx;
    if (t1->ctx != t2->ctx) return fail(c, KATGPU_ERR_INVALID_ARG, "tables belong to different contexts");
    if (t1->d.k != t2->d.k)
        return fail(c, KATGPU_ERR_MISMATCH, "Cannot process hashes that were created with different K-mer lengths.  Expected: %u.  Key length was %u", t1->d.k, t2->d.k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = refresh_counters(t1); if (rc) return rc;
    rc = refresh_counters(t2); if (rc) return rc;
    const bool wide = t1->d.keys_b != nullptr;               // same k => same key width in both tables
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
    // and the probe key equals the stored key; probe form (random HBM probes) otherwise.
    const bool same_grid = !wide && t1->d.p1 == t2->d.p1 && t1->d.p2 == t2->d.p2 && t1->d.mz == t2->d.mz && t1->d.n_regions > 1 && !g_no_join;   // the join holds 12-byte slots
    const bool ident1 = t1->d.canonical || !canon2;          // pass 1 probes canonical(key) iff input 2 is canonical
    const bool ident2 = t2->d.canonical != 0;                // pass 2 always probes canonical(key)
    const size_t join1 = ((lds1 + 15) & ~(size_t)15) + (size_t)t2->d.region_slots * 12, join2 = ((lds2 + 15) & ~(size_t)15) + (size_t)t1->d.region_slots * 12;
    // the join streams BOTH tables (~2.2 TB/s measured); probing costs ~1.3 random sector reads per scanned k-mer (~55 G/s):
    // a small table scanned against a big one is cheaper probed, a big one against a small one is cheaper joined
    auto join_pays = [](const katgpu_table* scan, const katgpu_table* probe) {
        if (scan->d.cap + probe->d.cap < ((uint64_t)64 << 20)) return true;        // small either way: take the join
        return 12.0 * (double)(scan->d.cap + probe->d.cap) / 2.2e12 < 1.3 * (double)scan->distinct / 55e9;
    };
    // pass 1 as a join can leave a bit per slot of hash 2 ("hash 1 holds this k-mer"); pass 2 is then a scan of hash 2 (k_comp_seen)
    const bool join_1 = same_grid && ident1 && (g_force_join || join_pays(t1, t2));
    const uint32_t wpr = (t2->d.region_slots + 31) / 32;
    const bool marked = join_1 && !g_no_seen && t1->d.canonical && t2->d.canonical && t2->ones == 0 && join1 + (size_t)wpr * 4 <= 150 * 1024;
    uint32_t* seen_bits = nullptr;
    if (marked) {
        if (pool_alloc(c, (void**)&seen_bits, (size_t)t2->d.n_regions * wpr * 4) != hipSuccess) { (void)hipGetLastError(); seen_bits = nullptr; }
        a.seen = seen_bits; a.seen_wpr = wpr;
    }
    auto join_grid = [&](size_t lds, uint32_t regions) { return std::min<uint32_t>(regions, (uint32_t)c->n_cu * (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (lds + 512), 4))); };
    {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS1, t1->d.cap);
        if (join_1 && join1 <= 150 * 1024) {
            const size_t j1 = join1 + (a.seen ? (size_t)wpr * 4 : 0);
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_join<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            hipLaunchKernelGGL(k_comp_join<1>, dim3(join_grid(j1, t1->d.n_regions)), dim3(512), j1, c->stream, t1->d, t1->n_ovf, t2->d, t2->n_ovf, a);
        } else if (wide)
            hipLaunchKernelGGL((k_comp<1, true>), dim3(reducer_grid(c, t1->d.cap + 1, 4)), dim3(256), lds1, c->stream, t1->d, t1->n_ovf, t2->d, t2->n_ovf, a);
        else
            hipLaunchKernelGGL((k_comp<1, false>), dim3(reducer_grid(c, t1->d.cap + 1, 4)), dim3(256), lds1, c->stream, t1->d, t1->n_ovf, t2->d, t2->n_ovf, a);
    }
    {
        ScopedTimer tm(c, KATGPU_K_COMP_PASS2, t2->d.cap);
        if (a.seen && join1 <= 150 * 1024) {
            hipLaunchKernelGGL(k_comp_seen, dim3(std::min<uint32_t>(t2->d.n_regions, (uint32_t)c->n_cu * 4)), dim3(512), lds2, c->stream, t2->d, t2->n_ovf, a);
        } else if (same_grid && ident2 && join2 <= 150 * 1024 && (g_force_join || join_pays(t2, t1))) {
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_join<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            hipLaunchKernelGGL(k_comp_join<2>, dim3(join_grid(join2, t2->d.n_regions)), dim3(512), join2, c->stream, t2->d, t2->n_ovf, t1->d, t1->n_ovf, a);
        } else if (wide)
            hipLaunchKernelGGL((k_comp<2, true>), dim3(reducer_grid(c, t2->d.cap + 1, 4)), dim3(256), lds2, c->stream, t2->d, t2->n_ovf, t1->d, t1->n_ovf, a);
        else
            hipLaunchKernelGGL((k_comp<2, false>), dim3(reducer_grid(c, t2->d.cap + 1, 4)), dim3(256), lds2, c->stream, t2->d, t2->n_ovf, t1->d, t1->n_ovf, a);
    }
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
    TOUCH(t3);
    if (!t3 || !ends_mx || !middle_mx || !mixed_mx) return KATGPU_ERR_INVALID_ARG;
    int rc = katgpu_comp(t1, t2, canon1, canon2, d1_scale, d2_scale, d1_bins, d2_bins, main_mx, counters, spectra);
    if (rc) return rc;
    katgpu_ctx* c = t1->ctx;
    if (t3->ctx != c) return fail(c, KATGPU_ERR_INVALID_ARG, "tables belong to different contexts");
    if (t3->d.k != t1->d.k)
        return fail(c, KATGPU_ERR_MISMATCH, "Cannot process hashes that were created with different K-mer lengths.  Expected: %u.  Key length was %u", t1->d.k, t3->d.k);
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
        ScopedTimer tm(c, KATGPU_K_COMP_PASS1, t1->d.cap);
        if (t1->d.keys_b)
            hipLaunchKernelGGL(k_comp3_pass1<true>, dim3(reducer_grid(c, t1->d.cap + 1, 3)), dim3(256), 3 * COMP_TILE * COMP_TILE * sizeof(uint32_t), c->stream,
                               t1->d, t1->n_ovf, t2->d, t2->n_ovf, t3->d, t3->n_ovf, a);
        else
            hipLaunchKernelGGL(k_comp3_pass1<false>, dim3(reducer_grid(c, t1->d.cap + 1, 3)), dim3(256), 3 * COMP_TILE * COMP_TILE * sizeof(uint32_t), c->stream,
                               t1->d, t1->n_ovf, t2->d, t2->n_ovf, t3->d, t3->n_ovf, a);
        hipLaunchKernelGGL(k_comp3_pass3, dim3(reducer_grid(c, t3->d.cap + 1, 8)), dim3(256), 0, c->stream, t3->d, t3->n_ovf, d + 3 * cells);
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
