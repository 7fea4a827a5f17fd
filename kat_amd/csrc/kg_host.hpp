// kg_host.hpp -- what the host-side translation units of libkatgpu.so share: the context and table objects behind the opaque
// handles of include/katgpu.h, error plumbing, the allocation pool, launch timing.  The library is split by concern:
//   kg_context.hip   context, pool, profile counters, device buffers, the synthetic workload
//   kg_table.hip     table life cycle (create / regrow / stats), lookups and profiles, record export / merge
//   kg_count.hip     counting: the direct kernel, the partitioned counter's host loop, the host feeder, katgpu_count*
//   kg_scan.hip      device-side record scan of raw FASTQ / FASTA bytes (katgpu_count_files' fast path)
//   kg_exchange.hip  region-ordered extraction / merge for the multi-GPU exchange
//   kg_comm.hip      the exchange itself over RCCL (katgpu_comm_*)
//   kg_reduce.hip    hist / gcp / comp
// gfx950 only; there is no CPU path in this library.
#pragma once
#include "../../include/katgpu.h"
#include "kg_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
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
    std::mutex prof_mu;                  // the profile counters and the event pool: the feeders count on a worker thread
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
    std::mutex pool_mu;                    // (the pool is reached from the caller's thread, a table's allocation thread and the reservation thread)
    std::thread reserve_thread;            // katgpu_reserve: memory of a table to come, being allocated beside the caller's work
    std::atomic<size_t> reserve_bytes{0};  // ... of this size (0: none pending or parked)
    void* reserved_p = nullptr;            // ... parked here, apart from the pool (pool_mu): only a table made "like" another takes it (pool_alloc's take_reservation)
    std::unordered_map<void*, size_t> block_bytes;      // real size of every live pooled-class allocation
    // scratch arena of the partitioned counter (level-1 / level-2 buffers, histograms); kept across calls
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    std::unordered_set<const void*> lds_attr;   // kernels whose dynamic-LDS ceiling has been raised on this device
    bool part_attr_set = false, merge_attr_set = false;   // the dynamic-LDS attributes of the partition / merge kernels have been set on this device
    bool arena_busy = false;              // a partition round is using it: pool_alloc must not free it to satisfy a table growth
    // Who allocates first when several threads ask the driver for tens of GB at once.  On some boxes a hipMalloc costs ~1 ms per 40 MB
    // (the driver clears what it hands out) and the driver serves one request at a time: the 9 GB of scan buffers, which the readers
    // need before a byte can move, waited 2.1 s behind the table's 39 GB and the arena -- measured, round 5 -- where they take 54 ms
    // elsewhere.  So: scan buffers first (scan_waiting: katgpu_count raises it, the first feeder's setup lowers it), then the table and
    // the arena (big_alloc_running), then a reservation (katgpu_reserve).  Waits are bounded: a hint about order, not a lock.
    std::atomic<int> scan_waiting{0}, big_alloc_running{0};
    size_t arena_limit = 0;               // != 0: count calls size the arena to at most this (the file feeders: their rounds are bounded by what arrives)
    bool arena_borrowed = false;          // katgpu_scratch_acquire handed the arena out: it must not be freed behind the caller's back
    int count_blocks_per_cu = 6;
    // the device scan's buffers (kg_scan.hip), kept across files and tables: one pinned segment per reader thread (pinning a GiB costs a
    // quarter of a second -- more than moving it: the readers stage through 8 MiB each, not through whole batches), two batches of raw
    // bytes and of output on the device, the line arrays.  Freed by katgpu_release_scratch / katgpu_shutdown.
    struct ScanCache {
        std::vector<uint8_t*> pin_seg; std::vector<hipStream_t> seg_stream; size_t pin_seg_bytes = 0;
        uint8_t *raw[2] = {nullptr, nullptr}, *acc[2] = {nullptr, nullptr}, *raw_al = nullptr;
        size_t acc_bytes = 0;
        uint32_t *tile_cnt = nullptr, *NL = nullptr, *len_off = nullptr, *line_tile_sum = nullptr;
        uint64_t *tile_off = nullptr, *line_tile_off = nullptr;
        unsigned long long* flags = nullptr;
        hipStream_t up[2] = {nullptr, nullptr};
        size_t buf_bytes = 0; int n_buf = 0;
        uint64_t cap_lines = 0;
    } scan;
};
void scan_cache_release(katgpu_ctx* c);      // kg_scan.hip

struct katgpu_table {
    katgpu_ctx* ctx = nullptr;
    // The device-side table.  Every reader goes through dev(); `dv` itself is for those who know about the zeroing below.
    // Lazy zeroing (packed tables of size, katgpu_table_create): the slots are NOT cleared when the table is made -- the partitioned
    // counter's first round visits every region once, in order, and its apply starts a region from zeros in LDS instead of loading it
    // (kg_partition.hpp: k_p3_apply_pk, zero_fill): the table's first sweep costs neither a memset nor a read.  zero_from = the first region
    // that has not been zeroed yet (~0: all have); whoever else touches the table first -- dev() -- clears the rest.
    mutable DevTable dv{};
    mutable uint64_t zero_from = ~0ULL;
    DevTable& dev() const { if (zero_from != ~0ULL) zero_rest(); return dv; }
    void zero_rest() const {
        const uint64_t from = zero_from;
        zero_from = ~0ULL;
        if (from < dv.n_regions && dv.keys && ctx &&
            hipMemsetAsync(dv.keys + from * dv.region_slots, 0, (size_t)(dv.n_regions - from) * dv.region_slots * sizeof(uint64_t), ctx->stream) != hipSuccess) {
            (void)hipGetLastError();
            zero_failed = true;                  // the slots past `from` still hold what the memory held: refresh_counters() fails every result read from this table
        }
    }
    mutable bool zero_failed = false;
    int disable_grow = 0;
    uint32_t n_ovf = 0;          // refreshed by refresh_counters()
    uint64_t distinct = 0;       // idem (slots in use + all-ones key)
    uint64_t ones = 0;
    // overflow guard of the unchecked (no-return) +1 adds: no 32-bit counter exceeds count_bound + unchecked_adds
    uint64_t count_bound = 0;    // largest counter value possible at the last sweep (0 for a fresh table)
    uint64_t unchecked_adds = 0; // window starts launched through k_count since then
    uint32_t n_regrows = 0;      // how often the table had to grow (the host mirror words the reference's warning from it)
    uint8_t carry[64];           // last k-1 bytes of the previous host batch of the current file
    uint32_t carry_n = 0;
    // katgpu_count allocates the table on a thread of its own while the feeders already read and parse (tens of GB of hipMalloc cost a
    // second: as long as the first GBs of the files take to arrive); whoever needs the slots first calls table_wait
    std::thread alloc_thread;
    std::mutex alloc_mu;
    int alloc_rc = 0;
    std::string alloc_err;
};

inline int fail(katgpu_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}

// entry points that handle one-word k-mers only (lookups by 64-bit key, .jf, the multi-GPU exchange, sect/cold profiles)
#define NARROW_ONLY(t, what)                                                                             \
    do {                                                                                                 \
        if ((t)->dv.keys_b) return fail((t)->ctx, KATGPU_ERR_K, "%s is not available for k > 32 (k = %u)", what, (t)->dv.k);      /* (dv: metadata only -- a table left uncleared stays that way) */ \
    } while (0)

#define HIPCHK(c, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return fail((c), _e == hipErrorOutOfMemory ? KATGPU_ERR_NOMEM : KATGPU_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- shared helpers (defined in kg_context.hip / kg_table.hip) ----
void resolve_pending(katgpu_ctx* c);
int grid_for(katgpu_ctx* c, uint64_t items, int block, int per_cu);
void pool_trim(katgpu_ctx* c);
hipError_t pool_alloc(katgpu_ctx* c, void** p, size_t bytes, bool take_reservation = false);
void pool_release(katgpu_ctx* c, void* p);
void release_arena(katgpu_ctx* c);

// a string inside a JSON line (the katgpu_timing lines): quotes, backslashes and control characters escaped
inline std::string json_escaped(const char* s) {
    std::string o;
    for (; s && *s; ++s) {
        const unsigned char ch = (unsigned char)*s;
        if (ch == '"' || ch == '\\') { o += '\\'; o += (char)ch; }
        else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
        else o += (char)ch;
    }
    return o;
}
inline double now_ms();
// (see katgpu_ctx::scan_waiting) wait, at most max_ms, until the allocations that should go first have been made
inline void alloc_turn(katgpu_ctx* c, bool also_big, double max_ms);
inline double now_ms() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline void alloc_turn(katgpu_ctx* c, bool also_big, double max_ms) {
    const double t0 = now_ms();
    while ((c->scan_waiting.load(std::memory_order_acquire) || (also_big && c->big_alloc_running.load(std::memory_order_acquire))) && now_ms() - t0 < max_ms) {
        timespec ts{0, 1000000};
        nanosleep(&ts, nullptr);
    }
}
static const double g_t_loaded = now_ms();           // when the library was loaded: KATGPU_TRACE stamps are relative to it
inline double since_load() { return now_ms() - g_t_loaded; }
// Test hooks (KATGPU_TEST_*) and A/B switches are read only when KATGPU_TESTING is set: a production process ignores them, and the
// kernels that honour one (the spill hook of the apply kernels) are separate instantiations it never launches.
// What a deployment may tune stays plain (INTEGRATION.md lists them): KATGPU_TRACE, KATGPU_ARENA_FRACTION, KATGPU_RING_MB,
// KATGPU_PART_MIN_STARTS, KATGPU_INGEST_*, KATGPU_DEVICE_SCAN.
static const bool g_trace = getenv("KATGPU_TRACE") != nullptr;
static const bool g_timing = getenv("KATGPU_TIMING") != nullptr;   // one "katgpu_timing {json}" line on stderr per file / phase: what bench.py's end_to_end.breakdown is made of
static const bool g_testing = getenv("KATGPU_TESTING") != nullptr;
inline const char* hook(const char* name) { return g_testing ? getenv(name) : nullptr; }
inline uint64_t hook_u64(const char* name, uint64_t dflt) { const char* v = hook(name); return v ? strtoull(v, nullptr, 10) : dflt; }

constexpr int MAX_EXCHANGE_PARTS_HOST = 256;        // ranks of an exchange (kg_kernels.hpp: MAX_EXCHANGE_PARTS)
// table geometry limits
constexpr uint32_t AP2_MAX_SLOTS = 10240;           // the apply kernels: a region of at most this many slots (120 KB of KV12 region + 36 KB of queues; 80 KB packed)
constexpr int AP2_QCAP_BIG = 192;                   // KV12 apply: queue entries per wave for regions beyond 8192 slots

// table life cycle (kg_table.hip)
int alloc_dev_table(katgpu_ctx* c, uint32_t k, int canonical, uint64_t cap, DevTable* out, uint32_t like_p1 = 0, uint32_t like_p2 = 0,
                    bool* lazy_zero = nullptr /* non-null: a packed table of size may leave its slots uncleared and say so (katgpu_table::zero_from) */);
void free_dev_table(katgpu_ctx* c, DevTable& d);
int refresh_counters(katgpu_table* t);
int regrow(katgpu_table* t, uint64_t new_cap);
bool table_may_stay_uncleared(const DevTable& d);
double load_limit(const DevTable& d);
int ensure_room(katgpu_table* t, uint64_t incoming);
int table_wait(katgpu_table* t);               // the table's device arrays exist (katgpu_count allocates them asynchronously)
// counting (kg_count.hip)
int count_resident(katgpu_table* t, const uint8_t* dev_bases, size_t n);
// large plain FASTQ / FASTA files: raw bytes to the device, record scan there (kg_scan.hip).  *took = false: not a file for this path
bool device_scan_applies(const char* path, uint32_t trim5p, uint64_t* size_out, uint8_t* first_byte);
size_t scan_arena_bytes(uint8_t first_byte);   // the partition arena such a file's count calls will want (FASTQ stripped on the host: bigger rounds)
int count_file_device_scan(katgpu_table* t, const char* path, uint32_t trim5p, bool* took, int rank = 0, int world = 1);   // world > 1: this rank's batches of a FASTQ file

// HIP events around a launch on the ctx stream; elapsed time is collected lazily.
struct ScopedTimer {
    katgpu_ctx* c; int cls; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(katgpu_ctx* c_, int cls_, uint64_t units) : c(c_), cls(cls_) {
        std::lock_guard<std::mutex> lk(c->prof_mu);
        auto take = [&]() { hipEvent_t e = nullptr; if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else hipEventCreate(&e); return e; };
        a = take(); b = take();
        hipEventRecord(a, c->stream);
        c->prof_launches[cls] += 1; c->prof_units[cls] += units;
    }
    ~ScopedTimer() {
        hipEventRecord(b, c->stream);
        bool full;
        { std::lock_guard<std::mutex> lk(c->prof_mu); c->pending.push_back({a, b, cls}); full = c->pending.size() > 4096; }
        if (full) resolve_pending(c);
    }
};

