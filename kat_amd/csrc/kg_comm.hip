// kg_comm.hip -- the multi-GPU exchange behind the C ABI: katgpu_comm_* / katgpu_exchange_merge / katgpu_allreduce_u64.
//
// What it replaces: the reference's only cross-worker reductions -- ThreadedSparseMatrix::mergeThreadedMatricies
// (lib/include/kat/sparse_matrix.hpp:324-335), ThreadedCompCounters::merge (lib/src/comp_counters.cc:230-254), Histogram::merge
// (src/histogram.cc:146-160) -- which sum per-thread results inside the one process Comp::execute drives (src/comp.cc:108-183).
// Here the workers are one process per GPU.  Each counts its share of the reads into a LOCAL table; the tables are then merged by
// OWNER (a hash of the canonical k-mer, kg_device.hpp: owner_of): every (k-mer, count) record travels to its owner rank, which adds
// the counts -- exact integer sums, so the result is bit-identical to one GPU's.  Reducers run on the owned shards and their small
// results are summed with one all-reduce (katgpu_allreduce_u64).
//
// The exchange is region-ordered and in place (include/katgpu.h "region-ordered exchange"; the device side is kg_exchange.hip):
// the table is extracted once into a send list in the arena (8-byte key + 4-byte count, grouped by owner, ordered by region inside an
// owner), emptied -- it becomes the owner table -- and the list travels in chunks of consecutive regions, chunk c on the wire
// while chunk c-1 is applied region by region in LDS (k_merge_apply).
//
// Transports.  RCCL (dlopen'ed: a single-GPU process never loads it): one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per
// chunk on a stream of its own -- every peer at once, which is the shape xGMI's point-to-point links want -- and ncclAllGather /
// ncclAllReduce for the small things.  SHM: ranks of one node stage their records through files in /dev/shm; slow, exact, needs
// nothing but a shared file system -- it is what carries ranks that SHARE a GPU (the test suite on a one-GPU box: RCCL refuses two
// ranks on one device).  Ranks on DISTINCT devices never take it by accident: a communicator that cannot have RCCL there fails
// (katgpu_comm_init says why) unless the caller asked for the staging transport by name (KATGPU_COMM_TRANSPORT=shm) or allowed the
// fall-back (KATGPU_COMM_ALLOW_SHM=1) -- a /dev/shm number must not pass for an xGMI one.  KATGPU_COMM_TRANSPORT = rccl | shm | auto.
// The protocol above the transport is the same code.
//
// Liveness.  Ranks may reach a collective minutes apart (`kat --gpus N` deals whole .gz files rank by rank), so no wait is bounded by
// a wall clock: every rank's communicator runs a heartbeat (a counter in the shared block, advanced every 50 ms by a thread of its
// own), and a wait gives up only when a peer has raised the abort flag or when a peer's heartbeat has stood still for
// KATGPU_COMM_TIMEOUT_S (default 60 s): a peer that died, not one that is busy.  KATGPU_COMM_MAX_WAIT_S (default: none) bounds a
// single wait by the clock for harnesses that prefer an error to a wedged link (bench.py sets it).
#include "kg_host.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// ---- RCCL through dlopen ----
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // (tests: KATGPU_RCCL_LIB names a stand-in that lets ranks SHARING a GPU take this branch -- tests/native/fake_rccl.cc; read only
    // under KATGPU_TESTING=1, like every hook)
    if (const char* test_lib = hook("KATGPU_RCCL_LIB")) r.lib = dlopen(test_lib, RTLD_NOW | RTLD_LOCAL);
    else
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
    if (!r.lib) return r;
#define KG_SYM(F) r.F = reinterpret_cast<decltype(r.F)>(dlsym(r.lib, "nccl" #F))
    KG_SYM(GetUniqueId); KG_SYM(CommInitRank); KG_SYM(CommDestroy); KG_SYM(GroupStart); KG_SYM(GroupEnd); KG_SYM(Send); KG_SYM(Recv);
    KG_SYM(AllGather); KG_SYM(AllReduce); KG_SYM(GetErrorString);
#undef KG_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather && r.AllReduce && r.GetErrorString;
    return r;
}

// RCCL's two bootstrap calls (ncclGetUniqueId, ncclCommInitRank) open sockets and look for network interfaces, and have been seen not
// to come back on a box (round 5: one `--gpus 1` run of 200 sat in there for its caller's whole 300 s).  They run on a thread of their
// own; the caller waits KATGPU_COMM_INIT_TIMEOUT_S (default 120 s) and then reports an error instead of hanging -- the thread is left
// behind (it owns its state through the shared_ptr), the process is about to fail anyway.
static const double g_comm_init_timeout_s = getenv("KATGPU_COMM_INIT_TIMEOUT_S") ? std::max(1.0, atof(getenv("KATGPU_COMM_INIT_TIMEOUT_S"))) : 120.0;
struct BootCall { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclUniqueId id; ncclComm_t comm = nullptr; };
// false: the call has not returned within the limit
template <typename F>
static bool rccl_boot_call(int device, F&& f, std::shared_ptr<BootCall> st) {
    std::thread([device, f, st]() {
        (void)hipSetDevice(device);
        const ncclResult_t r = f(*st);
        std::lock_guard<std::mutex> lk(st->mu);
        st->r = r; st->done = true;
        st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->mu);
    return st->cv.wait_for(lk, std::chrono::duration<double>(g_comm_init_timeout_s), [&] { return st->done; });
}

// ---- the id ranks share: [magic | 16 bytes of token (names the /dev/shm objects) | has_rccl | ncclUniqueId] ----
constexpr uint32_t ID_MAGIC = 0x4B474331;      // "KGC1"
struct CommId { uint32_t magic; uint32_t has_rccl; char token[24]; ncclUniqueId nccl; };
static_assert(sizeof(CommId) <= KATGPU_COMM_ID_BYTES, "id");

// ---- rendezvous block in /dev/shm: a sense-reversing barrier and a small mailbox per rank ----
constexpr size_t MAILBOX = 64 * 1024;
struct ShmHeader {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;
    uint32_t world;
    std::atomic<uint32_t> aborted;   // a rank that fails inside a collective raises it: its peers leave their barriers with an error instead of waiting for ever
    uint32_t pad_[11];               // (the heartbeats start on a cache line of their own)
};
struct RankBeat { std::atomic<uint64_t> beat; std::atomic<uint32_t> gone; uint32_t pad_[13]; };   // one cache line per rank: its heartbeat; gone: it has left (katgpu_comm_free)
static_assert(sizeof(ShmHeader) == 64 && sizeof(RankBeat) == 64, "shared block layout");
// A wait for peers -- at a barrier, for a transfer -- ends with an error (through katgpu_last_error, not a hang) when a peer is DEAD:
// its heartbeat has not moved for this long.  Not a bound on how long a healthy peer may take to get there.
static const double g_comm_timeout_ms = 1e3 * (getenv("KATGPU_COMM_TIMEOUT_S") ? std::max(0.5, atof(getenv("KATGPU_COMM_TIMEOUT_S"))) : 60.0);
// optional: no single wait longer than this, whatever the heartbeats say (0: unbounded)
static const double g_comm_gone_grace_ms = 1e3 * (getenv("KATGPU_COMM_GONE_GRACE_S") ? std::max(0.0, atof(getenv("KATGPU_COMM_GONE_GRACE_S"))) : 5.0);
// (read at the first wait, not when the library is loaded: the host binary gives it a finite default in main(), host/kat_main.cc)
static double comm_max_wait_ms() { static const double v = 1e3 * (getenv("KATGPU_COMM_MAX_WAIT_S") ? std::max(0.0, atof(getenv("KATGPU_COMM_MAX_WAIT_S"))) : 0.0); return v; }
constexpr int BEAT_PERIOD_MS = 50;

struct Msg { int peer; void* dev; size_t bytes; };          // one side of a point-to-point transfer (device memory)

}  // namespace

struct katgpu_comm {
    katgpu_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    bool use_rccl = false;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;             // transport stream: chunk c travels while chunk c-1 is merged on the context's stream
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::string token;
    ShmHeader* hdr = nullptr; RankBeat* beats = nullptr; uint8_t* boxes = nullptr; size_t shm_bytes = 0;
    std::thread beat_thread; std::atomic<bool> beat_stop{false};
    int distinct_devices = 1;                 // how many different devices the ranks run on (1: they all share one)
    uint64_t seq = 0;                         // names the shm files of successive transfers
    double ms_exchange = 0, ms_merge = 0, ms_extract = 0, ms_allreduce = 0;
    uint64_t bytes_sent = 0, merge_launches = 0;
    uint64_t records_sent = 0, record_bytes_sent = 0;          // what katgpu_exchange_merge put on the wire as records (not the count matrices, not the all-reduce)
    std::vector<void*> pending;                                // Exchanges begun (katgpu_exchange_begin) and not yet finished, oldest first (at most two)
    bool wire_packed = false;                                  // the last exchange's records: 9 bytes (remainder + count) or 12 (key + count)
    std::string transport_note;
    uint8_t* host_stage = nullptr; size_t host_stage_bytes = 0;
};

namespace {

int comm_fail(katgpu_comm* m, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (m && m->ctx) m->ctx->err = buf;
    if (m && m->hdr) m->hdr->aborted.store(1, std::memory_order_release);      // the peers are, or will be, waiting for this rank
    return code;
}
#define NCCLCHK(m, expr)                                                                                               \
    do {                                                                                                               \
        ncclResult_t _r = (expr);                                                                                      \
        if (_r != ncclSuccess) return comm_fail((m), KATGPU_ERR_DEVICE, "%s: %s", #expr, rccl().GetErrorString(_r));  \
    } while (0)

std::string shm_name(const std::string& token, const char* what, uint64_t seq = 0, int a = 0, int b = 0, int idx = 0) {
    char buf[176];
    snprintf(buf, sizeof buf, "/dev/shm/katgpu-%s-%s-%llu-%d-%d-%d", token.c_str(), what, (unsigned long long)seq, a, b, idx);
    return buf;
}

// What a waiting rank knows of its peers' health: each peer's last heartbeat value and when it was last seen to move.
struct Liveness {
    katgpu_comm* m; std::vector<uint64_t> last; std::vector<double> moved, gone_at; double t0; char why[256];
    explicit Liveness(katgpu_comm* m_) : m(m_), last((size_t)m_->world, 0), moved((size_t)m_->world, now_ms()), gone_at((size_t)m_->world, 0.0), t0(now_ms()) {
        why[0] = 0;
        for (int r = 0; r < m->world; ++r) last[r] = m->beats[r].beat.load(std::memory_order_relaxed);
    }
    // false: this wait should end with an error (`why`): a peer failed, left, or its heartbeat stands still; or the optional wall-clock
    // bound.  The caller looks once more at what it waits for before it gives up (a peer may leave right after it did its part).
    bool ok() {
        if (m->hdr->aborted.load(std::memory_order_acquire)) { snprintf(why, sizeof why, "a peer rank failed (rank %d gives up)", m->rank); return false; }
        const double now = now_ms();
        for (int r = 0; r < m->world; ++r) {
            if (r == m->rank) continue;
            const uint64_t b = m->beats[r].beat.load(std::memory_order_relaxed);
            if (b != last[r]) { last[r] = b; moved[r] = now; continue; }
            // A peer that has LEFT (katgpu_comm_free) is not yet a failure: a rank that finished its side of the last collective may free
            // its communicator while a slower rank's transfer is still landing (ncclAllReduce returns per rank).  It becomes one when what
            // this rank waits for has not happened a grace period later (KATGPU_COMM_GONE_GRACE_S, 5 s).
            if (m->beats[r].gone.load(std::memory_order_acquire)) {
                if (gone_at[r] == 0.0) gone_at[r] = now;
                if (now - gone_at[r] > g_comm_gone_grace_ms) { snprintf(why, sizeof why, "rank %d has left the communicator while rank %d waits for it", r, m->rank); return false; }
                continue;
            }
            if (now - moved[r] > g_comm_timeout_ms) {
                snprintf(why, sizeof why, "no sign of life from rank %d for %.0f s (rank %d gives up; KATGPU_COMM_TIMEOUT_S)", r, (now - moved[r]) / 1e3, m->rank);
                return false;
            }
        }
        if (comm_max_wait_ms() > 0 && now - t0 > comm_max_wait_ms()) { snprintf(why, sizeof why, "rank %d waited %.0f s (KATGPU_COMM_MAX_WAIT_S)", m->rank, (now - t0) / 1e3); return false; }
        return true;
    }
};

// every rank of the communicator: wait until all have arrived -- or until a peer has failed (ShmHeader::aborted) or died (its heartbeat)
int shm_barrier(katgpu_comm* m) {
    if (m->world == 1) return KATGPU_OK;
    if (m->hdr->aborted.load(std::memory_order_acquire)) return comm_fail(m, KATGPU_ERR_DEVICE, "a peer rank failed (rank %d leaves the barrier)", m->rank);
    const uint32_t gen = m->hdr->generation.load(std::memory_order_acquire);
    if (m->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)m->world) {
        m->hdr->arrived.store(0, std::memory_order_relaxed);
        m->hdr->generation.store(gen + 1, std::memory_order_release);
    } else {
        Liveness live(m);
        for (uint32_t spins = 0; m->hdr->generation.load(std::memory_order_acquire) == gen; ++spins) {
            if (spins <= 1000) continue;
            if ((spins & 255) == 0 && !live.ok()) {
                if (m->hdr->generation.load(std::memory_order_acquire) != gen) break;          // everyone did arrive (and one has left since)
                return comm_fail(m, KATGPU_ERR_DEVICE, "barrier: %s", live.why);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return KATGPU_OK;
}
// wait for the transport stream (or an event on it) the same way: a collective whose peer never posts its side would sit in
// hipStreamSynchronize for ever
int comm_wait(katgpu_comm* m, hipEvent_t ev /* or null: the whole stream */, const char* what) {
    Liveness live(m);
    auto query = [&]() { return ev ? hipEventQuery(ev) : hipStreamQuery(m->stream); };
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = query();
        if (e == hipSuccess) return KATGPU_OK;
        if (e != hipErrorNotReady) return comm_fail(m, KATGPU_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
        if (spins < 2000) continue;
        if ((spins & 255) == 0 && m->hdr && !live.ok()) {
            if (query() == hipSuccess) return KATGPU_OK;
            return comm_fail(m, KATGPU_ERR_DEVICE, "%s: %s", what, live.why);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// small host values through the mailboxes: out[r * n .. ) = rank r's n bytes
int host_allgather(katgpu_comm* m, const void* mine, size_t n, void* out) {
    if (n > MAILBOX) return comm_fail(m, KATGPU_ERR_INVALID_ARG, "host_allgather: %zu bytes per rank", n);
    if (m->world == 1) { memcpy(out, mine, n); return KATGPU_OK; }
    memcpy(m->boxes + (size_t)m->rank * MAILBOX, mine, n);
    int rc = shm_barrier(m);
    if (rc) return rc;
    for (int r = 0; r < m->world; ++r) memcpy((uint8_t*)out + (size_t)r * n, m->boxes + (size_t)r * MAILBOX, n);
    return shm_barrier(m);
}

int ensure_host_stage(katgpu_comm* m, size_t bytes) {
    if (m->host_stage_bytes >= bytes) return KATGPU_OK;
    if (m->host_stage) hipHostFree(m->host_stage);
    m->host_stage = nullptr; m->host_stage_bytes = 0;
    if (hipHostMalloc((void**)&m->host_stage, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return comm_fail(m, KATGPU_ERR_NOMEM, "pinned staging of %zu bytes", bytes); }
    m->host_stage_bytes = bytes;
    return KATGPU_OK;
}

// A group of point-to-point transfers, all ranks together.  RCCL: asynchronous on m->stream, *ev is recorded behind it.  SHM:
// done when this returns (every send is a file in /dev/shm that its receiver reads and the sender removes).
int transfer(katgpu_comm* m, const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipEvent_t ev) {
    katgpu_ctx* c = m->ctx;
    if (m->use_rccl) {
        bool any = false;
        for (auto& s : sends) any = any || s.bytes;
        for (auto& r : recvs) any = any || r.bytes;
        if (any) {
            NCCLCHK(m, rccl().GroupStart());
            for (auto& s : sends) if (s.bytes) NCCLCHK(m, rccl().Send(s.dev, s.bytes, ncclUint8, s.peer, m->nccl, m->stream));
            for (auto& r : recvs) if (r.bytes) NCCLCHK(m, rccl().Recv(r.dev, r.bytes, ncclUint8, r.peer, m->nccl, m->stream));
            NCCLCHK(m, rccl().GroupEnd());
        }
        for (auto& s : sends) m->bytes_sent += s.bytes;
        if (ev) HIPCHK(c, hipEventRecord(ev, m->stream));
        return KATGPU_OK;
    }
    const uint64_t seq = m->seq++;
    size_t biggest = 0;
    for (auto& s : sends) biggest = std::max(biggest, s.bytes);
    for (auto& r : recvs) biggest = std::max(biggest, r.bytes);
    int rc = ensure_host_stage(m, std::max<size_t>(biggest, 4096));
    if (rc) return rc;
    // (a group may carry several messages for one peer -- keys, then counts: the n-th to a peer meets the n-th from it)
    std::vector<int> nth((size_t)m->world, 0);
    for (auto& s : sends) {
        const int idx = nth[s.peer]++;
        if (!s.bytes) continue;
        if (hipMemcpy(m->host_stage, s.dev, s.bytes, hipMemcpyDeviceToHost) != hipSuccess) return comm_fail(m, KATGPU_ERR_DEVICE, "staging a message for rank %d", s.peer);
        const std::string name = shm_name(m->token, "x", seq, m->rank, s.peer, idx);
        const int fd = ::open(name.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        if (fd < 0) return comm_fail(m, KATGPU_ERR_IO, "cannot create %s", name.c_str());
        size_t off = 0;
        while (off < s.bytes) { const ssize_t w = ::write(fd, m->host_stage + off, s.bytes - off); if (w <= 0) { ::close(fd); return comm_fail(m, KATGPU_ERR_IO, "short write to %s", name.c_str()); } off += (size_t)w; }
        ::close(fd);
        m->bytes_sent += s.bytes;
    }
    rc = shm_barrier(m);
    if (rc) return rc;
    std::fill(nth.begin(), nth.end(), 0);
    for (auto& r : recvs) {
        const int idx = nth[r.peer]++;
        if (!r.bytes) continue;
        const std::string name = shm_name(m->token, "x", seq, r.peer, m->rank, idx);
        const int fd = ::open(name.c_str(), O_RDONLY);
        if (fd < 0) return comm_fail(m, KATGPU_ERR_IO, "cannot open %s", name.c_str());
        size_t off = 0;
        while (off < r.bytes) { const ssize_t g = ::read(fd, m->host_stage + off, r.bytes - off); if (g <= 0) { ::close(fd); return comm_fail(m, KATGPU_ERR_IO, "short read from %s", name.c_str()); } off += (size_t)g; }
        ::close(fd);
        ::unlink(name.c_str());
        if (hipMemcpy(r.dev, m->host_stage, r.bytes, hipMemcpyHostToDevice) != hipSuccess) return comm_fail(m, KATGPU_ERR_DEVICE, "unstaging a message from rank %d", r.peer);
    }
    return shm_barrier(m);
}
int transfer_wait(katgpu_comm* m, hipEvent_t ev) {
    if (m->use_rccl && ev) return comm_wait(m, ev, "exchange");
    return KATGPU_OK;
}

// out[r * n ..) = rank r's n u64 (host arrays)
int allgather_u64(katgpu_comm* m, const uint64_t* mine, size_t n, uint64_t* out) {
    katgpu_ctx* c = m->ctx;
    if (m->world == 1) { memcpy(out, mine, n * 8); return KATGPU_OK; }
    if (!m->use_rccl && n * 8 <= MAILBOX) return host_allgather(m, mine, n * 8, out);
    uint64_t* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, (size_t)(m->world + 1) * n * 8));
    int rc = KATGPU_OK;
    if (hipMemcpy(d, mine, n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "allgather upload");
    if (!rc && m->use_rccl) {
        ncclResult_t r = rccl().AllGather(d, d + n, n, ncclUint64, m->nccl, m->stream);
        if (r != ncclSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "ncclAllGather: %s", rccl().GetErrorString(r));
        else rc = comm_wait(m, nullptr, "allgather");
    } else if (!rc) {
        std::vector<Msg> s, rv;
        for (int p = 0; p < m->world; ++p) {
            if (p == m->rank) { if (hipMemcpy(d + n + (size_t)p * n, d, n * 8, hipMemcpyDeviceToDevice) != hipSuccess) rc = KATGPU_ERR_DEVICE; continue; }
            s.push_back({p, d, n * 8});
            rv.push_back({p, d + n + (size_t)p * n, n * 8});
        }
        if (!rc) rc = transfer(m, s, rv, nullptr);
    }
    if (!rc && hipMemcpy(out, d + n, (size_t)m->world * n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "allgather download");
    hipFree(d);
    return rc;
}

uint64_t host_revcomp(uint64_t x, uint32_t k) {
    uint64_t r = 0;
    for (uint32_t i = 0; i < k; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
    return r;
}
uint32_t host_owner_of(uint64_t key, uint32_t k, uint32_t n_parts) {          // kg_device.hpp: owner_of
    const uint64_t rc = host_revcomp(key, k), cn = rc < key ? rc : key;
    return (uint32_t)(((unsigned __int128)mix64(cn ^ 0x9E3779B97F4A7C15ULL) * n_parts) >> 64);
}

double wall_ms() { return now_ms(); }

}  // namespace

// ------------------------------------------------------------------ the communicator ------------------

extern "C" int katgpu_comm_unique_id(void* id_out) {
    if (!id_out) return KATGPU_ERR_INVALID_ARG;
    CommId id{};
    id.magic = ID_MAGIC;
    FILE* f = fopen("/dev/urandom", "rb");
    uint8_t rnd[10] = {0};
    if (f) { if (fread(rnd, 1, sizeof rnd, f) != sizeof rnd) rnd[0] = (uint8_t)getpid(); fclose(f); }
    snprintf(id.token, sizeof id.token, "%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", rnd[0], rnd[1], rnd[2], rnd[3], rnd[4], rnd[5], rnd[6], rnd[7], rnd[8], rnd[9]);
    const char* tr = getenv("KATGPU_COMM_TRANSPORT");
    if (!(tr && !strcmp(tr, "shm")) && rccl().ok) {
        int dev = 0; (void)hipGetDevice(&dev);
        auto st = std::make_shared<BootCall>();
        if (!rccl_boot_call(dev, [](BootCall& b) { return rccl().GetUniqueId(&b.id); }, st)) {
            fprintf(stderr, "[katgpu] ncclGetUniqueId did not return within %.0f s (KATGPU_COMM_INIT_TIMEOUT_S)\n", g_comm_init_timeout_s);
            return KATGPU_ERR_DEVICE;
        }
        if (st->r == ncclSuccess) { id.nccl = st->id; id.has_rccl = 1; }
    }
    memset(id_out, 0, KATGPU_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return KATGPU_OK;
}

static void drop_pending(katgpu_comm* m);                        // (an exchange begun and never finished: its buffers go with the communicator)
extern "C" void katgpu_comm_free(katgpu_comm* m) {
    if (!m) return;
    if (m->ctx) hipSetDevice(m->ctx->device);
    drop_pending(m);
    if (m->nccl) rccl().CommDestroy(m->nccl);
    for (auto& e : m->ev) if (e) hipEventDestroy(e);
    if (m->stream) hipStreamDestroy(m->stream);
    if (m->host_stage) hipHostFree(m->host_stage);
    if (m->beat_thread.joinable()) { m->beat_stop.store(true); m->beat_thread.join(); }
    if (m->hdr) {
        if (m->beats) m->beats[m->rank].gone.store(1, std::memory_order_release);          // a peer still waiting for this rank learns it at once, not from a silent heartbeat
        const bool last = m->hdr->attached.fetch_sub(1) == 1;
        munmap((void*)m->hdr, m->shm_bytes);
        if (last || m->rank == 0) ::unlink(shm_name(m->token, "hdr").c_str());
    }
    delete m;
}

extern "C" int katgpu_comm_init(katgpu_ctx* c, int rank, int world, const void* id_in, katgpu_comm** out) {
    if (!c || !out || !id_in || world < 1 || rank < 0 || rank >= world || world > (int)MAX_EXCHANGE_PARTS_HOST) return KATGPU_ERR_INVALID_ARG;
    *out = nullptr;
    CommId id;
    memcpy(&id, id_in, sizeof id);
    if (id.magic != ID_MAGIC) return fail(c, KATGPU_ERR_INVALID_ARG, "katgpu_comm_init: not an id made by katgpu_comm_unique_id");
    HIPCHK(c, hipSetDevice(c->device));
    katgpu_comm* m = new katgpu_comm();
    m->ctx = c; m->rank = rank; m->world = world;
    id.token[sizeof id.token - 1] = 0;
    m->token = id.token;
    // the rendezvous block: every rank maps it (rank order does not matter: O_CREAT, then ftruncate to the same size)
    m->shm_bytes = sizeof(ShmHeader) + (size_t)world * sizeof(RankBeat) + (size_t)world * MAILBOX;
    const std::string hname = shm_name(m->token, "hdr");
    const int fd = ::open(hname.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)m->shm_bytes) != 0) { if (fd >= 0) ::close(fd); delete m; return fail(c, KATGPU_ERR_IO, "cannot create %s", hname.c_str()); }
    void* p = mmap(nullptr, m->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) { delete m; return fail(c, KATGPU_ERR_IO, "cannot map %s", hname.c_str()); }
    m->hdr = (ShmHeader*)p;                       // (a fresh file is zero-filled: counters start at 0)
    m->beats = (RankBeat*)((uint8_t*)p + sizeof(ShmHeader));
    m->boxes = (uint8_t*)p + sizeof(ShmHeader) + (size_t)world * sizeof(RankBeat);
    // this rank's heartbeat: alive as long as the process is, whatever the main thread is busy with (counting a .gz for minutes)
    m->beats[rank].beat.store(1, std::memory_order_relaxed);
    m->beat_thread = std::thread([m, rank]() {
        while (!m->beat_stop.load(std::memory_order_relaxed)) {
            m->beats[rank].beat.fetch_add(1, std::memory_order_relaxed);
            std::this_thread::sleep_for(std::chrono::milliseconds(BEAT_PERIOD_MS));
        }
    });
    m->hdr->attached.fetch_add(1);
    // wait for everyone (bounded: a rank that never shows up must not hang the others for ever)
    const double t0 = wall_ms();
    while (m->hdr->attached.load() < (uint32_t)world) {
        if (wall_ms() - t0 > 120e3) {
            const int seen = (int)m->hdr->attached.load();        // (before the block is unmapped)
            katgpu_comm_free(m);
            return fail(c, KATGPU_ERR_DEVICE, "katgpu_comm_init: %d rank(s) of %d showed up within 120 s", seen, world);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    {   // (every way out from here on gives the communicator back: katgpu_comm_free copes with a half-made one)
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        for (auto& ev : m->ev) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) { katgpu_comm_free(m); return fail(c, KATGPU_ERR_DEVICE, "katgpu_comm_init: %s", hipGetErrorString(e)); }
    }
    // which devices the ranks run on: ranks that share one cannot have RCCL (it refuses two ranks on a device) and stage through /dev/shm;
    // ranks on devices of their own must not end up there by accident
    int rc = KATGPU_OK;
    {
        char mine_id[64] = {0};
        if (hipDeviceGetPCIBusId(mine_id, (int)sizeof mine_id - 1, c->device) != hipSuccess) { (void)hipGetLastError(); snprintf(mine_id, sizeof mine_id, "device-%d", c->device); }
        std::vector<char> ids((size_t)world * sizeof mine_id);
        rc = host_allgather(m, mine_id, sizeof mine_id, ids.data());
        if (rc) { katgpu_comm_free(m); return rc; }
        std::vector<std::string> uniq;
        for (int r = 0; r < world; ++r) { std::string d(ids.data() + (size_t)r * sizeof mine_id); if (std::find(uniq.begin(), uniq.end(), d) == uniq.end()) uniq.push_back(d); }
        m->distinct_devices = (int)uniq.size();
    }
    // transport: RCCL when every rank can have it
    const char* tr = getenv("KATGPU_COMM_TRANSPORT");
    const bool asked_shm = tr && !strcmp(tr, "shm");
    const bool want_rccl = !asked_shm && id.has_rccl && rccl().ok;
    uint32_t mine = 0;
    if (want_rccl) {
        auto st = std::make_shared<BootCall>();
        st->id = id.nccl;
        if (!rccl_boot_call(c->device, [world, rank](BootCall& b) { return rccl().CommInitRank(&b.comm, world, b.id, rank); }, st)) {
            m->beats[rank].gone.store(1, std::memory_order_release);
            // (the communicator is NOT taken apart: the thread inside RCCL may still touch what it was given; the process is to end)
            return fail(c, KATGPU_ERR_DEVICE, "katgpu_comm_init: ncclCommInitRank did not return within %.0f s on rank %d of %d (KATGPU_COMM_INIT_TIMEOUT_S)", g_comm_init_timeout_s, rank, world);
        }
        const ncclResult_t r = st->r;
        if (r == ncclSuccess) { m->nccl = st->comm; mine = 1; }
        else { m->nccl = nullptr; m->transport_note = std::string("RCCL refused to initialise (") + rccl().GetErrorString(r) + ")"; (void)hipGetLastError(); }
    } else m->transport_note = asked_shm ? "KATGPU_COMM_TRANSPORT=shm" : (id.has_rccl ? "librccl could not be loaded here" : "no RCCL id (librccl missing where the id was made)");
    std::vector<uint32_t> all((size_t)world);
    rc = host_allgather(m, &mine, sizeof mine, all.data());
    if (rc) { katgpu_comm_free(m); return rc; }
    m->use_rccl = true;
    for (uint32_t v : all) m->use_rccl = m->use_rccl && v;
    if (!m->use_rccl && m->nccl) { rccl().CommDestroy(m->nccl); m->nccl = nullptr; if (m->transport_note.empty()) m->transport_note = "a peer could not initialise RCCL"; }
    if (tr && !strcmp(tr, "rccl") && !m->use_rccl) { std::string why = m->transport_note; katgpu_comm_free(m); return fail(c, KATGPU_ERR_DEVICE, "KATGPU_COMM_TRANSPORT=rccl: %s", why.c_str()); }
    // no silent fall-back between devices: /dev/shm staging is for ranks that share a device (or for whoever asked for it by name)
    if (!m->use_rccl && world > 1 && m->distinct_devices > 1 && !asked_shm && !getenv("KATGPU_COMM_ALLOW_SHM")) {
        std::string why = m->transport_note;
        const int nd = m->distinct_devices;
        katgpu_comm_free(m);
        return fail(c, KATGPU_ERR_DEVICE, "katgpu_comm_init: %d ranks on %d devices, and RCCL is not to be had (%s): refusing to stage the exchange through /dev/shm "
                    "(KATGPU_COMM_ALLOW_SHM=1 or KATGPU_COMM_TRANSPORT=shm to take it knowingly)", world, nd, why.c_str());
    }
    if (g_trace) fprintf(stderr, "[katgpu] comm: rank %d of %d, transport %s%s%s\n", rank, world, m->use_rccl ? "RCCL" : "SHM (staged through /dev/shm)", m->transport_note.empty() ? "" : ": ", m->transport_note.c_str());
    *out = m;
    return KATGPU_OK;
}

extern "C" int katgpu_comm_rank(const katgpu_comm* m) { return m ? m->rank : -1; }
extern "C" int katgpu_comm_world(const katgpu_comm* m) { return m ? m->world : 0; }
extern "C" const char* katgpu_comm_transport(const katgpu_comm* m) { return !m ? "" : m->use_rccl ? "rccl" : "shm"; }
extern "C" const char* katgpu_comm_transport_note(const katgpu_comm* m) { return m ? m->transport_note.c_str() : ""; }
extern "C" int katgpu_comm_distinct_devices(const katgpu_comm* m) { return m ? m->distinct_devices : 0; }

extern "C" int katgpu_comm_barrier(katgpu_comm* m) {
    if (!m) return KATGPU_ERR_INVALID_ARG;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    return shm_barrier(m);
}

extern "C" int katgpu_comm_wire(katgpu_comm* m, uint64_t* records_sent, uint64_t* record_bytes_sent, int* packed) {
    if (!m) return KATGPU_ERR_INVALID_ARG;
    if (records_sent) *records_sent = m->records_sent;
    if (record_bytes_sent) *record_bytes_sent = m->record_bytes_sent;
    if (packed) *packed = m->wire_packed ? 1 : 0;
    return KATGPU_OK;
}

extern "C" int katgpu_comm_stats(katgpu_comm* m, double* ms_extract, double* ms_exchange, double* ms_merge, double* ms_allreduce, uint64_t* bytes_sent, uint64_t* merge_launches) {
    if (!m) return KATGPU_ERR_INVALID_ARG;
    if (ms_extract) *ms_extract = m->ms_extract;
    if (ms_exchange) *ms_exchange = m->ms_exchange;
    if (ms_merge) *ms_merge = m->ms_merge;
    if (ms_allreduce) *ms_allreduce = m->ms_allreduce;
    if (bytes_sent) *bytes_sent = m->bytes_sent;
    if (merge_launches) *merge_launches = m->merge_launches;
    return KATGPU_OK;
}

// Sum of `n` u64 over all ranks, in place, every rank gets the result: the small results of the reducers (hist 80 KB, gcp 216 KB,
// comp 8 MB + counters) -- what mergeThreadedMatricies / ThreadedCompCounters::merge / Histogram::merge do for threads.
extern "C" int katgpu_allreduce_u64(katgpu_comm* m, uint64_t* buf, size_t n) {
    if (!m || (n && !buf)) return KATGPU_ERR_INVALID_ARG;
    if (m->world == 1 || n == 0) return KATGPU_OK;
    katgpu_ctx* c = m->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = wall_ms();
    int rc = KATGPU_OK;
    if (m->use_rccl) {
        uint64_t* d = nullptr;
        HIPCHK(c, hipMalloc((void**)&d, n * 8));
        if (hipMemcpy(d, buf, n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "allreduce upload");
        if (!rc) {
            ncclResult_t r = rccl().AllReduce(d, d, n, ncclUint64, ncclSum, m->nccl, m->stream);
            if (r != ncclSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "ncclAllReduce: %s", rccl().GetErrorString(r));
            else if ((rc = comm_wait(m, nullptr, "allreduce")) == KATGPU_OK && hipMemcpy(buf, d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = comm_fail(m, KATGPU_ERR_DEVICE, "allreduce");
        }
        hipFree(d);
    } else {
        // every rank writes its vector, reads all of them (host memory only: the vectors are small)
        const uint64_t seq = m->seq++;
        const std::string mine = shm_name(m->token, "r", seq, m->rank, 0);
        FILE* f = fopen(mine.c_str(), "wb");
        if (!f || fwrite(buf, 8, n, f) != n) rc = comm_fail(m, KATGPU_ERR_IO, "cannot write %s", mine.c_str());
        if (f) fclose(f);
        if (!rc) rc = shm_barrier(m);                            // (a rank that could not write has raised the abort flag: its peers leave here too)
        std::vector<uint64_t> other(n);
        for (int r = 0; r < m->world && !rc; ++r) {
            if (r == m->rank) continue;
            const std::string name = shm_name(m->token, "r", seq, r, 0);
            FILE* g = fopen(name.c_str(), "rb");
            if (!g || fread(other.data(), 8, n, g) != n) rc = comm_fail(m, KATGPU_ERR_IO, "cannot read %s", name.c_str());
            if (g) fclose(g);
            if (!rc) for (size_t i = 0; i < n; ++i) buf[i] += other[i];
        }
        if (!rc) rc = shm_barrier(m);
        ::unlink(mine.c_str());
    }
    m->ms_allreduce += wall_ms() - t0;
    return rc;
}

// ------------------------------------------------------------------ the exchange ----------------------

static size_t xalign(size_t n, size_t a = 256) { return (n + a - 1) / a * a; }
// (sized for key + count records; packed records -- 4 + 1 + 4 bytes -- are carved out of the same room)
static size_t exchange_bytes(uint64_t total_send, uint64_t set_records) {
    return xalign(8 * std::max<uint64_t>(total_send, 1)) + xalign(4 * std::max<uint64_t>(total_send, 1)) +
           2 * (xalign(8 * std::max<uint64_t>(set_records, 1)) + xalign(4 * std::max<uint64_t>(set_records, 1))) + 256;
}
static const bool g_comm_trace = getenv("KATGPU_COMM_TRACE") != nullptr;      // one stderr line per stage of katgpu_exchange_merge, per rank: where a run of many ranks stands
#define CTRACE(m, ...) do { if (g_comm_trace) { fprintf(stderr, "[katgpu comm %d/%d +%.0f ms] ", (m)->rank, (m)->world, wall_ms() - t_begin); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
static const bool g_wire_packed = !getenv("KATGPU_COMM_PACKED_RECORDS") || atoi(getenv("KATGPU_COMM_PACKED_RECORDS")) != 0;   // A/B + tests: 0 = key + count records (12 bytes) even between ranks that share the grid

// Wide tables (k > 32): the simple exchange -- records (hi, lo, count) grouped by owner, all to all, the table emptied and refilled
// with what arrived.  Not region-ordered: the wide table's hash is not one to one and its slots are 20 bytes in three arrays, which the
// LDS merge is not built for; wide tables count through the direct kernel and merge through it too (k_merge_w).  The table keeps its
// handle; it grows if its owner share is larger than what it held.
static int exchange_merge_wide(katgpu_comm* m, katgpu_table* t) {
    katgpu_ctx* c = m->ctx;
    const int world = m->world, rank = m->rank;
    const uint64_t mine[2] = {t->dev().k, t->dev().canonical};
    std::vector<uint64_t> all((size_t)world * 2);
    int rc = allgather_u64(m, mine, 2, all.data());
    if (rc) return rc;
    for (int s = 0; s < world; ++s)
        if (all[(size_t)s * 2] != mine[0] || all[(size_t)s * 2 + 1] != mine[1]) return fail(c, KATGPU_ERR_MISMATCH, "katgpu_exchange_merge: ranks disagree on k / canonical");
    double t0 = wall_ms();
    std::vector<uint64_t> sizes((size_t)world), s_all((size_t)world * world);
    rc = katgpu_table_partition_sizes(t, (uint32_t)world, sizes.data());
    if (rc) return rc;
    rc = allgather_u64(m, sizes.data(), (size_t)world, s_all.data());
    if (rc) return rc;
    std::vector<uint64_t> send_off((size_t)world + 1, 0), recv_off((size_t)world + 1, 0);
    for (int p = 0; p < world; ++p) { send_off[p + 1] = send_off[p] + sizes[p]; recv_off[p + 1] = recv_off[p] + s_all[(size_t)p * world + rank]; }
    const uint64_t ns = std::max<uint64_t>(send_off[world], 1), nr = std::max<uint64_t>(recv_off[world], 1);
    uint64_t* buf = nullptr;
    if (hipMalloc((void**)&buf, (ns + nr) * 24) != hipSuccess) { (void)hipGetLastError(); return fail(c, KATGPU_ERR_NOMEM, "wide exchange: %llu + %llu records of 24 bytes", (unsigned long long)ns, (unsigned long long)nr); }
    struct Free { void* p; ~Free() { hipFree(p); } } free_buf{buf};
    uint64_t* snd[3] = {buf, buf + ns, buf + 2 * ns};                      // hi | lo | counts, each grouped by owner
    uint64_t* rcv[3] = {buf + 3 * ns, buf + 3 * ns + nr, buf + 3 * ns + 2 * nr};
    rc = katgpu_table_partition_wide(t, (uint32_t)world, send_off.data(), snd[0], snd[1], snd[2]);      // (returns with the records written)
    if (rc) return rc;
    m->ms_extract += wall_ms() - t0;
    t0 = wall_ms();
    std::vector<Msg> sends, recvs;
    for (int p = 0; p < world; ++p) {
        const uint64_t n_out = sizes[p], n_in = recv_off[p + 1] - recv_off[p];
        for (int a = 0; a < 3; ++a) {
            if (p == rank) { if (n_out) HIPCHK(c, hipMemcpyAsync(rcv[a] + recv_off[p], snd[a] + send_off[p], n_out * 8, hipMemcpyDeviceToDevice, m->stream)); continue; }
            sends.push_back({p, snd[a] + send_off[p], (size_t)n_out * 8});
            recvs.push_back({p, rcv[a] + recv_off[p], (size_t)n_in * 8});
        }
    }
    rc = transfer(m, sends, recvs, m->ev[0]);
    if (!rc) rc = transfer_wait(m, m->ev[0]);
    if (rc) return rc;
    rc = comm_wait(m, nullptr, "exchange");
    if (rc) return rc;
    m->ms_exchange += wall_ms() - t0;
    t0 = wall_ms();
    rc = katgpu_table_clear(t);
    if (!rc) rc = katgpu_table_merge_device_wide(t, rcv[0], rcv[1], rcv[2], (size_t)recv_off[world]);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ++m->merge_launches;
    m->ms_merge += wall_ms() - t0;
    rc = shm_barrier(m);
    return rc ? rc : refresh_counters(t);
}

// Route every record of `t` to its owner rank, IN PLACE: on return the table holds exactly the k-mers this rank owns, their counts
// summed over all ranks.  It keeps its storage and its region grid (a second table created "like" the first still joins with it
// region by region).  Every rank of the communicator calls this, with tables of one k / one strand mode.  world == 1 runs the whole
// protocol on the rank's own send list (extraction, clear, region-by-region merge): the table comes back as it was.
//
// Two shapes of one protocol (struct Exchange):
//   pipelined (katgpu_exchange_merge): the send list and TWO receive sets in the context's arena; chunk c on the wire while chunk c - 1
//       is applied;
//   split (katgpu_exchange_begin ... katgpu_exchange_finish): the send list and a receive set PER CHUNK in a buffer of the exchange's own
//       (the arena is the counter's: the caller counts its next input in between), every chunk posted at once; finish waits chunk by
//       chunk and applies.  When a rank cannot have that buffer, all ranks run the pipelined shape inside begin and finish has nothing to do.
struct Exchange {
    katgpu_comm* m; katgpu_table* t; katgpu_ctx* c;
    int world, rank;
    double t_begin;
    katgpu_geometry geo{};
    std::vector<uint64_t> geos;
    uint32_t R = 0, R_min = 0;
    uint32_t* d_cnt = nullptr;                                    // [world x R] u32: records of region g owned by part p (small: outside the arena)
    std::vector<uint32_t*> d_rcnt;                                // per sender: the region counts of what it holds for me (d_rcnt[rank] points into d_cnt)
    std::vector<uint64_t> sizes, recv_from, part_base;
    uint64_t total_send = 0, recv_other = 0;
    std::vector<std::vector<uint64_t>> cnt_cum, rcnt_cum, bounds, recv_sz, send_off;
    uint32_t C = 1;
    bool split = false, packed = false, done = false;
    void* own_buf = nullptr;                                      // split: the exchange's own buffer
    uint64_t* skeys = nullptr; uint32_t* scounts = nullptr; uint32_t* srem_lo = nullptr; uint8_t* srem_hi = nullptr;
    struct Set { uint64_t* keys; uint32_t* rem_lo; uint8_t* rem_hi; uint32_t* counts; };
    std::vector<Set> sets;                                        // pipelined: 2 (chunk & 1); split: one per chunk
    std::vector<hipEvent_t> evs;                                  // split: one per chunk (pipelined: the communicator's two)
    static constexpr uint32_t BIG = 4200;
    std::vector<uint64_t> big_keys, big_counts;
    uint32_t n_big = 0;
    struct Layout { int s; uint64_t o, n; };
    std::vector<std::vector<Layout>> lay;

    Exchange(katgpu_comm* m_, katgpu_table* t_) : m(m_), t(t_), c(m_->ctx), world(m_->world), rank(m_->rank), t_begin(wall_ms()), big_keys(BIG), big_counts(BIG) {}
    ~Exchange() {
        for (size_t i = 0; i < d_rcnt.size(); ++i) if ((int)i != rank && d_rcnt[i]) hipFree(d_rcnt[i]);
        if (d_cnt) hipFree(d_cnt);
        if (own_buf) hipFree(own_buf);
        for (auto e : evs) if (e) hipEventDestroy(e);
    }
    uint32_t R_of(int s) const { return (uint32_t)geos[(size_t)s * 6 + 2]; }
    Set& set_of(uint32_t ch) { return sets[split ? ch : (ch & 1)]; }
    hipEvent_t ev_of(uint32_t ch) { return split ? evs[ch] : m->ev[ch & 1]; }
    int agree(bool mine, bool* all_agree) {                       // (what a rank can do depends on its own table / memory: all must agree before one acts)
        uint64_t v = mine ? 1 : 0;
        std::vector<uint64_t> all((size_t)world);
        int rc = allgather_u64(m, &v, 1, all.data());
        if (rc) return rc;
        bool ok = true;
        for (uint64_t x : all) ok = ok && x != 0;
        *all_agree = ok;
        return KATGPU_OK;
    }

    // geometry of every rank's table, how many records go where per region, the region counts of what I will receive
    int prepare() {
        int rc = katgpu_table_geometry(t, &geo);
        if (rc) return rc;
        const uint64_t g_mine[6] = {geo.k, geo.canonical, geo.n_regions, geo.region_slots, geo.p1, geo.p2};
        geos.resize((size_t)world * 6);
        rc = allgather_u64(m, g_mine, 6, geos.data());
        if (rc) return rc;
        CTRACE(m, "geometries known: R %u slots %u p1 %u p2 %u", geo.n_regions, geo.region_slots, geo.p1, geo.p2);
        for (int s = 0; s < world; ++s)
            if (geos[(size_t)s * 6] != geo.k || geos[(size_t)s * 6 + 1] != geo.canonical) return fail(c, KATGPU_ERR_MISMATCH, "katgpu_exchange_merge: ranks disagree on k / canonical");
        R = geo.n_regions;
        R_min = R;
        for (int s = 0; s < world; ++s) R_min = std::min(R_min, R_of(s));

        // ---- pass 1: how many records go where, per region ----
        HIPCHK(c, hipMalloc((void**)&d_cnt, (size_t)world * R * 4 + 64));
        sizes.resize((size_t)world);
        rc = katgpu_table_extract_sizes(t, (uint32_t)world, d_cnt, sizes.data());
        if (rc) return rc;
        for (uint64_t s : sizes) total_send += s;
        std::vector<uint64_t> s_all((size_t)world * world);
        rc = allgather_u64(m, sizes.data(), (size_t)world, s_all.data());
        if (rc) return rc;
        CTRACE(m, "sizes known: %llu records to send", (unsigned long long)total_send);
        recv_from.resize((size_t)world);
        for (int s = 0; s < world; ++s) recv_from[s] = s_all[(size_t)s * world + rank];         // what each peer holds for me
        // the region counts of what I will receive: row `rank` of every peer's matrix
        d_rcnt.assign((size_t)world, nullptr);
        {
            std::vector<Msg> sends, recvs;
            for (int p = 0; p < world; ++p) {
                if (p == rank) { d_rcnt[p] = d_cnt + (size_t)rank * R; continue; }
                HIPCHK(c, hipMalloc((void**)&d_rcnt[p], (size_t)R_of(p) * 4 + 64));
                sends.push_back({p, d_cnt + (size_t)p * R, (size_t)R * 4});
                recvs.push_back({p, d_rcnt[p], (size_t)R_of(p) * 4});
            }
            rc = transfer(m, sends, recvs, m->ev[0]);
            if (!rc) rc = transfer_wait(m, m->ev[0]);
            if (rc) return rc;
        }
        CTRACE(m, "region counts exchanged");
        // prefix sums on the host (cnt: mine, per owner; rcnt: per sender, of the records it holds for me)
        cnt_cum.resize((size_t)world); rcnt_cum.resize((size_t)world);
        {
            std::vector<uint32_t> h;
            for (int p = 0; p < world; ++p) {
                h.resize(R);
                HIPCHK(c, hipMemcpy(h.data(), d_cnt + (size_t)p * R, (size_t)R * 4, hipMemcpyDeviceToHost));
                cnt_cum[p].assign((size_t)R + 1, 0);
                for (uint32_t g = 0; g < R; ++g) cnt_cum[p][g + 1] = cnt_cum[p][g] + h[g];
                h.resize(R_of(p));
                HIPCHK(c, hipMemcpy(h.data(), d_rcnt[p], (size_t)R_of(p) * 4, hipMemcpyDeviceToHost));
                rcnt_cum[p].assign((size_t)R_of(p) + 1, 0);
                for (uint32_t g = 0; g < R_of(p); ++g) rcnt_cum[p][g + 1] = rcnt_cum[p][g] + h[g];
            }
        }
        part_base.assign((size_t)world + 1, 0);
        for (int p = 0; p < world; ++p) part_base[p + 1] = part_base[p] + sizes[p];
        for (int s = 0; s < world; ++s) if (s != rank) recv_other += recv_from[s];
        return KATGPU_OK;
    }
    void cut(uint32_t chunks) {                                   // chunks of consecutive regions, per sender's grid; what each brings me
        C = chunks;
        bounds.resize((size_t)world); recv_sz.resize((size_t)world);
        for (int s = 0; s < world; ++s) {
            bounds[s].resize((size_t)C + 1);
            for (uint32_t i = 0; i <= C; ++i) bounds[s][i] = (uint64_t)i * R_of(s) / C;
            recv_sz[s].resize(C);
            for (uint32_t i = 0; i < C; ++i) recv_sz[s][i] = rcnt_cum[s][bounds[s][i + 1]] - rcnt_cum[s][bounds[s][i]];
        }
    }
    uint64_t chunk_in(uint32_t i) const { uint64_t x = 0; for (int s = 0; s < world; ++s) if (s != rank) x += recv_sz[s][i]; return x; }

    // the buffers, the send list; the emptied table becomes the owner table
    int plan(bool want_split) {
        int rc;
        const uint32_t C0 = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)hook_u64("KATGPU_TEST_EXCHANGE_CHUNKS", 4), R_min));
        uint8_t* a = nullptr;
        if (want_split) {                                         // a buffer of the exchange's own: the send list + a receive set per chunk
            cut(C0);
            size_t bytes = xalign(8 * std::max<uint64_t>(total_send, 1)) + xalign(4 * std::max<uint64_t>(total_send, 1)) + 256;
            for (uint32_t i = 0; i < C; ++i) bytes += xalign(8 * std::max<uint64_t>(chunk_in(i), 1)) + xalign(4 * std::max<uint64_t>(chunk_in(i), 1));
            const bool got = !hook("KATGPU_TEST_EXCHANGE_NO_SPLIT") && hipMalloc(&own_buf, bytes) == hipSuccess;
            if (!got) { (void)hipGetLastError(); own_buf = nullptr; }
            bool all = false;
            rc = agree(got, &all);
            if (rc) return rc;
            if (!all && own_buf) { hipFree(own_buf); own_buf = nullptr; }
            split = all;
            if (split) {
                a = (uint8_t*)own_buf;
                evs.assign(C, nullptr);
                for (auto& e : evs) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            } else if (g_trace || g_comm_trace) fprintf(stderr, "[katgpu comm %d/%d] no buffer of %.1f GB for a split exchange on some rank: the pipelined one, now\n", rank, world, bytes / 1e9);
        }
        uint64_t set_records = 1;
        if (!split) {                                             // the arena: as few chunks as it allows (>= 4 for the overlap)
            void* arena = nullptr; size_t cap = 0;
            {
                const size_t want = exchange_bytes(total_send, (recv_other + C0 - 1) / C0 * 5 / 4);
                rc = katgpu_scratch_acquire(c, 0, &arena, &cap);
                if (!rc && cap < want && katgpu_scratch_acquire(c, want, &arena, &cap) != KATGPU_OK) rc = katgpu_scratch_acquire(c, 0, &arena, &cap);   // (keeps the old arena when the larger one cannot be had)
                if (rc) return rc;
            }
            uint32_t chunks = C0;
            for (;;) {
                cut(chunks);
                set_records = 1;
                for (uint32_t i = 0; i < C; ++i) set_records = std::max(set_records, chunk_in(i));
                bool ok = false;
                rc = agree(exchange_bytes(total_send, set_records) <= cap, &ok);
                if (rc) return rc;
                if (ok) break;
                if (C >= R_min) return fail(c, KATGPU_ERR_NOMEM, "katgpu_exchange_merge: the send list and one region's receive buffers do not fit the exchange scratch");
                chunks = std::min(C * 2, R_min);
            }
            a = (uint8_t*)arena;
        }
        send_off.resize((size_t)world);                           // [owner][chunk boundary]: index into the send list
        for (int p = 0; p < world; ++p) {
            send_off[p].resize((size_t)C + 1);
            for (uint32_t i = 0; i <= C; ++i) send_off[p][i] = part_base[p] + cnt_cum[p][bounds[rank][i]];
        }

        // ---- pass 2: the send list ----
        // Records: key + count (12 bytes) -- or, when EVERY rank's table has this one's grid and can give them, what a slot holds of the k-mer
        // + count (4 + 1 + 4 = 9 bytes: katgpu_table_extract_packed); the region a record lies in says the rest, and the chunks are region ranges.
        bool mine = g_wire_packed && katgpu_table_packed_records(t) != 0;
        for (int s = 0; s < world; ++s) mine = mine && geos[(size_t)s * 6 + 4] == geo.p1 && geos[(size_t)s * 6 + 5] == geo.p2 && R_of(s) == R;
        rc = agree(mine, &packed);
        if (rc) return rc;
        m->wire_packed = packed;
        CTRACE(m, "%u chunks (%s), records of %d bytes", C, split ? "all on the wire at once" : "one travels while one is applied", packed ? 9 : 12);
        skeys = (uint64_t*)a;           a += xalign(8 * std::max<uint64_t>(total_send, 1));     // (packed: the low words, then the high bytes, in the same room)
        scounts = (uint32_t*)a;         a += xalign(4 * std::max<uint64_t>(total_send, 1));
        srem_lo = (uint32_t*)skeys;
        srem_hi = (uint8_t*)skeys + xalign(4 * std::max<uint64_t>(total_send, 1));
        sets.resize(split ? C : 2);
        for (size_t i = 0; i < sets.size(); ++i) {
            const uint64_t n = split ? std::max<uint64_t>(chunk_in((uint32_t)i), 1) : set_records;
            sets[i].keys = (uint64_t*)a; sets[i].rem_lo = (uint32_t*)a; sets[i].rem_hi = a + xalign(4 * n); a += xalign(8 * n);
            sets[i].counts = (uint32_t*)a; a += xalign(4 * n);
        }
        rc = packed ? katgpu_table_extract_packed(t, (uint32_t)world, d_cnt, srem_lo, srem_hi, scounts, big_keys.data(), big_counts.data(), BIG, &n_big)
                    : katgpu_table_extract(t, (uint32_t)world, d_cnt, skeys, scounts, big_keys.data(), big_counts.data(), BIG, &n_big);
        if (rc) return rc;
        rc = katgpu_table_clear(t);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));               // the send list is complete before the transport stream reads it
        m->ms_extract += wall_ms() - t_begin;
        CTRACE(m, "send list written, table emptied");
        lay.resize(split ? C : 2);
        return KATGPU_OK;
    }
    std::vector<Layout>& lay_of(uint32_t ch) { return lay[split ? ch : (ch & 1)]; }

    int post(uint32_t ch) {
        std::vector<Msg> sends, recvs;
        std::vector<Layout>& layout = lay_of(ch);
        Set& rs = set_of(ch);
        uint64_t o = 0;
        layout.clear();
        for (int s = 0; s < world; ++s) {
            if (s == rank) continue;
            const uint64_t a0 = send_off[s][ch], n_out = send_off[s][ch + 1] - a0;
            if (n_out) {
                if (packed) { sends.push_back({s, srem_lo + a0, (size_t)n_out * 4}); sends.push_back({s, srem_hi + a0, (size_t)n_out}); }
                else sends.push_back({s, skeys + a0, (size_t)n_out * 8});
                sends.push_back({s, scounts + a0, (size_t)n_out * 4});
                m->records_sent += n_out; m->record_bytes_sent += n_out * (packed ? 9 : 12);
            }
            const uint64_t n_in = recv_sz[s][ch];
            if (n_in) {
                if (packed) { recvs.push_back({s, rs.rem_lo + o, (size_t)n_in * 4}); recvs.push_back({s, rs.rem_hi + o, (size_t)n_in}); }
                else recvs.push_back({s, rs.keys + o, (size_t)n_in * 8});
                recvs.push_back({s, rs.counts + o, (size_t)n_in * 4});
            }
            layout.push_back({s, o, n_in});
            o += n_in;
        }
        return transfer(m, sends, recvs, ev_of(ch));
    }
    int merge(uint32_t ch) {
        const std::vector<Layout>& layout = lay_of(ch);
        Set& rs = set_of(ch);
        const uint32_t my_lo = (uint32_t)bounds[rank][ch], my_hi = (uint32_t)bounds[rank][ch + 1];
        const uint64_t a0 = send_off[rank][ch], n_own = send_off[rank][ch + 1] - a0;
        if (packed) {                                             // (every rank has this table's grid -- as it was when the exchange began: the sources say so)
            std::vector<katgpu_merge_source_packed> src;
            if (n_own) src.push_back({srem_lo + a0, srem_hi + a0, scounts + a0, d_rcnt[rank] + my_lo, n_own, geo.p1, geo.p2});
            for (auto& l : layout) if (l.n) src.push_back({rs.rem_lo + l.o, rs.rem_hi + l.o, rs.counts + l.o, d_rcnt[l.s] + my_lo, l.n, geo.p1, geo.p2});
            if (src.empty()) return KATGPU_OK;
            ++m->merge_launches;
            return katgpu_table_merge_regions_packed(t, my_lo, my_hi, (uint32_t)src.size(), src.data());
        }
        std::vector<katgpu_merge_source> src;
        if (n_own) src.push_back({skeys + a0, scounts + a0, d_rcnt[rank] + my_lo, n_own, geo.p1, geo.p2});
        for (auto& l : layout) {
            if (!l.n) continue;
            const bool same_regions = bounds[l.s][ch] == my_lo && bounds[l.s][ch + 1] == my_hi && geos[(size_t)l.s * 6 + 4] == geo.p1 && geos[(size_t)l.s * 6 + 5] == geo.p2;
            src.push_back({rs.keys + l.o, rs.counts + l.o, same_regions ? d_rcnt[l.s] + my_lo : nullptr, l.n, (uint32_t)geos[(size_t)l.s * 6 + 4], (uint32_t)geos[(size_t)l.s * 6 + 5]});
        }
        if (src.empty()) return KATGPU_OK;
        ++m->merge_launches;
        return katgpu_table_merge_regions(t, my_lo, my_hi, (uint32_t)src.size(), src.data());
    }
    int arrived_then_merge(uint32_t ch) {
        double t0 = wall_ms();
        int rc = transfer_wait(m, ev_of(ch));
        if (rc) return rc;
        m->ms_exchange += wall_ms() - t0;
        t0 = wall_ms();
        CTRACE(m, "chunk %u arrived", ch);
        rc = merge(ch);
        if (rc) return rc;
        m->ms_merge += wall_ms() - t0;
        CTRACE(m, "chunk %u merged", ch);
        return KATGPU_OK;
    }
    // out-of-band records: counts beyond the record's field and the all-ones k-mer (a handful)
    int tail() {
        std::vector<uint64_t> mine((size_t)1 + 2 * BIG, 0), everyone((size_t)world * (1 + 2 * BIG));
        mine[0] = n_big;
        for (uint32_t i = 0; i < n_big; ++i) { mine[1 + i] = big_keys[i]; mine[1 + BIG + i] = big_counts[i]; }
        int rc = allgather_u64(m, mine.data(), mine.size(), everyone.data());
        if (rc) return rc;
        std::vector<uint64_t> ok_keys, ok_counts;
        for (int s = 0; s < world; ++s) {
            const uint64_t* e = everyone.data() + (size_t)s * (1 + 2 * BIG);
            for (uint64_t i = 0; i < e[0] && i < BIG; ++i)
                if (host_owner_of(e[1 + i], geo.k, (uint32_t)world) == (uint32_t)rank) { ok_keys.push_back(e[1 + i]); ok_counts.push_back(e[1 + BIG + i]); }
        }
        if (!ok_keys.empty()) { rc = katgpu_table_merge_host(t, ok_keys.data(), ok_counts.data(), ok_keys.size()); if (rc) return rc; }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        CTRACE(m, "out-of-band records done");
        rc = shm_barrier(m);                                      // nobody reuses its buffers while a peer may still be reading from them
        done = true;
        return rc ? rc : refresh_counters(t);
    }
    int pipelined() {
        for (uint32_t ch = 0; ch <= C; ++ch) {
            if (ch < C) {                                         // chunk ch goes on the wire ...
                const double t0 = wall_ms();
                int rc = post(ch);
                if (rc) return rc;
                m->ms_exchange += wall_ms() - t0;
            }
            if (ch > 0) { int rc = arrived_then_merge(ch - 1); if (rc) return rc; }     // ... while chunk ch - 1 is applied
        }
        return tail();
    }
    int begin(bool want_split) {
        int rc = prepare();
        if (!rc) rc = plan(want_split);
        if (rc) return rc;
        if (!split) return pipelined();
        const double t0 = wall_ms();
        for (uint32_t ch = 0; ch < C && !rc; ++ch) rc = post(ch);
        m->ms_exchange += wall_ms() - t0;
        return rc;
    }
    int finish() {
        if (done) return KATGPU_OK;
        t_begin = wall_ms();
        for (uint32_t ch = 0; ch < C; ++ch) { int rc = arrived_then_merge(ch); if (rc) return rc; }
        return tail();
    }
};

static void drop_pending(katgpu_comm* m) { for (void* p : m->pending) delete (Exchange*)p; m->pending.clear(); }

extern "C" int katgpu_exchange_merge(katgpu_comm* m, katgpu_table* t) {
    if (!m || !t || t->ctx != m->ctx) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = m->ctx;
    if (!m->pending.empty()) return fail(c, KATGPU_ERR_INVALID_ARG, "katgpu_exchange_merge: an exchange begun with katgpu_exchange_begin has not been finished");
    HIPCHK(c, hipSetDevice(c->device));
    if (t->dev().keys_b) return exchange_merge_wide(m, t);
    Exchange x(m, t);
    return x.begin(false);
}

// The same exchange in two calls, so that the caller's next input is counted while this table's records travel: begin extracts, empties
// the table and puts every chunk on the wire (from and into a buffer of the exchange's own -- the arena stays the counter's); finish waits
// for the chunks and applies them.  Between the two the table is not to be touched.  TWO exchanges may be under way: the second table's
// begin (its collectives queue behind the first one's transfers, long landed by then) before the first one's finish, so that the second
// table's records travel while the first one's are applied; they are finished in the order they were begun.
// Wide tables (k > 32) and ranks without room for the buffer do the whole exchange in begin.
extern "C" int katgpu_exchange_begin(katgpu_comm* m, katgpu_table* t) {
    if (!m || !t || t->ctx != m->ctx) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = m->ctx;
    if (m->pending.size() >= 2) return fail(c, KATGPU_ERR_INVALID_ARG, "katgpu_exchange_begin: two exchanges are under way already: finish the older one first");
    for (void* p : m->pending) if (((Exchange*)p)->t == t) return fail(c, KATGPU_ERR_INVALID_ARG, "katgpu_exchange_begin: this table's exchange has not been finished");
    HIPCHK(c, hipSetDevice(c->device));
    if (t->dev().keys_b) return exchange_merge_wide(m, t);
    Exchange* x = new Exchange(m, t);
    const int rc = x->begin(true);
    if (rc || x->done) { delete x; return rc; }
    m->pending.push_back(x);
    return KATGPU_OK;
}
extern "C" int katgpu_exchange_finish(katgpu_comm* m, katgpu_table* t) {
    if (!m || !t || t->ctx != m->ctx) return KATGPU_ERR_INVALID_ARG;
    katgpu_ctx* c = m->ctx;
    // (every rank finishes its exchanges in the order it began them: the tails are collectives)
    if (m->pending.empty() || ((Exchange*)m->pending.front())->t != t) {
        for (void* p : m->pending) if (((Exchange*)p)->t == t) return fail(c, KATGPU_ERR_INVALID_ARG, "katgpu_exchange_finish: an older exchange has to be finished first");
        return KATGPU_OK;                                          // (begin did it all)
    }
    Exchange* x = (Exchange*)m->pending.front();
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = x->finish();
    m->pending.erase(m->pending.begin());
    delete x;
    return rc;
}
