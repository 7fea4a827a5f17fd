// kg_scan.hip -- katgpu_count_files' fast path for large plain FASTQ / FASTA files: the host moves raw file bytes, the device parses.
//
//   reader threads: pread 8 MiB segments of the file into a PINNED segment of their own, each followed by its own H2D copy on the
//                   thread's stream (reading and copying overlap across the threads; a dozen copies are in flight)
//   the caller's thread, per batch of 512 MiB of file: find the record-aligned cut at the batch's end (host: two memchr), run the
//                   record scan on the device (kg_scan.hpp), read three words back, and hand the compacted base stream -- a
//                   resident buffer like any other -- to count_resident, i.e. to the partitioned counter.
// While a batch is scanned and counted the readers fill the other batch buffer.  The host parses nothing: the 16-thread parser team
// (kg_ingest.cpp: parse_file_parallel) delivered 22-35 GB/s of FASTQ and one feeder thread copied its output into pinned memory at
// 8.7; here the bound is pread + PCIe.
// A batch the device cannot vouch for (kg_scan.hpp lists what) is parsed by the host state machine from the batch's first byte,
// which the batch before it has proven to be a record start -- and so is the rest of the file: the stream stays the streaming
// parser's whatever the file looks like.
//
// FASTQ, round 5: ONLY THE BASES CROSS PCIe ("host strip", the default for FASTQ files here).  A FASTQ file is 47 % sequence; the rest --
// headers, '+' lines, qualities -- went to the device only to be scanned over and dropped there, by kernels that shared the GPU with the
// counter (a 47.7 GB file: 1.3 s, of it 0.97 s in the scan's launches and their synchronisation).  Now every reader thread takes an 8 MiB
// segment of the file, finds the record starts at its two ends (kg_ingest: find_record_start -- the neighbour computes the same cuts from
// the same bytes), checks that what lies between is plain four-line FASTQ and copies the sequence lines, each followed by 'N', into a
// pinned buffer of its own (kg_ingest: strip_fastq_records: four memchr and one memcpy per record); the caller's thread takes the
// segments IN FILE ORDER, checks that each begins where the one before ended, and issues its copy straight into the accumulation buffer the
// counter will be handed.  No scan kernel runs and half the bytes travel.  A segment that is not plain four-line FASTQ (multi-line
// records, odd quality lengths, a last line without its newline, a record longer than the look-ahead) stops the fast path there: what was
// committed is counted, and the file's rest goes through the host state machine from the last proven record start -- as above.
#include "kg_host.hpp"
#include "kg_ingest.hpp"
#include "kg_scan.hpp"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>

static const size_t g_scan_batch = (size_t)std::max<uint64_t>(1, getenv("KATGPU_SCAN_BATCH_MB") ? strtoull(getenv("KATGPU_SCAN_BATCH_MB"), nullptr, 10) : 512) << 20;
static const size_t g_scan_segment = (size_t)std::max<uint64_t>(1, getenv("KATGPU_SCAN_SEGMENT_MB") ? strtoull(getenv("KATGPU_SCAN_SEGMENT_MB"), nullptr, 10) : 8) << 20;
static const size_t g_scan_acc = (size_t)std::max<uint64_t>(64, getenv("KATGPU_SCAN_ACC_MB") ? strtoull(getenv("KATGPU_SCAN_ACC_MB"), nullptr, 10) : 4096) << 20;   // base stream counted per call
static const size_t g_scan_overlap_dflt = (size_t)1 << 20;        // how far past a batch's nominal end its cut may lie (a record, a FASTA line)
static const unsigned g_scan_threads = (unsigned)std::max<uint64_t>(1, getenv("KATGPU_SCAN_THREADS") ? strtoull(getenv("KATGPU_SCAN_THREADS"), nullptr, 10) : 16);
static const uint64_t g_scan_min_bytes = getenv("KATGPU_SCAN_MIN_BYTES") ? strtoull(getenv("KATGPU_SCAN_MIN_BYTES"), nullptr, 10) : ((uint64_t)64 << 20);
static const bool g_scan_off = getenv("KATGPU_DEVICE_SCAN") && atoi(getenv("KATGPU_DEVICE_SCAN")) == 0;
// How the readers get a file's bytes into their pinned segments: pread, or -- files on tmpfs (/dev/shm) -- a memcpy out of a mapping of
// the file.  Measured on the MI355X boxes' hosts (tools/reader_bench.hip, 16 threads, the FIRST read of a freshly written file, which
// is what a run pays): page-cache files pread at 230-330 GB/s, but tmpfs files at 16-27 GB/s (and no faster with more threads: every
// page's first pread moves it between the kernel's shared-memory LRU lists, under one lock), where the same bytes come out of a
// mapping at 120 GB/s.  KATGPU_SCAN_MMAP=0 / 1 forces one or the other.
static const bool g_scan_populate = getenv("KATGPU_SCAN_POPULATE") ? atoi(getenv("KATGPU_SCAN_POPULATE")) != 0 : false;   // strip readers: MADV_POPULATE_READ on a segment before it is looked at
static const int g_scan_mmap = getenv("KATGPU_SCAN_MMAP") ? atoi(getenv("KATGPU_SCAN_MMAP")) : -1;
// tests: small batches / overlaps (bytes) so that little files cross many cuts; force the host fall-back from batch N on
static const size_t g_test_scan_batch = (size_t)hook_u64("KATGPU_TEST_SCAN_BATCH", 0), g_test_scan_overlap = (size_t)hook_u64("KATGPU_TEST_SCAN_OVERLAP", 0);
static const size_t g_test_scan_segment = (size_t)hook_u64("KATGPU_TEST_SCAN_SEGMENT", 0);
static const uint64_t g_test_scan_fail_at = hook_u64("KATGPU_TEST_SCAN_FAIL_AT", ~0ULL);
// FASTQ files: the readers strip to the sequence lines on the host (above); KATGPU_FASTQ_STRIP=0: the device scan as for FASTA
static const bool g_fastq_strip = !(getenv("KATGPU_FASTQ_STRIP") && atoi(getenv("KATGPU_FASTQ_STRIP")) == 0);
// tests: tiny segments / look-ahead for the strip path (little files cross many cuts); force the hand-over to the host parser at a segment
static const size_t g_test_strip_segment = (size_t)hook_u64("KATGPU_TEST_STRIP_SEGMENT", 0), g_test_strip_overlap = (size_t)hook_u64("KATGPU_TEST_STRIP_OVERLAP", 0);
static const uint64_t g_test_strip_fail_at = hook_u64("KATGPU_TEST_STRIP_FAIL_AT", ~0ULL);
static const size_t g_strip_acc = (size_t)3 << 30;                // strip path: base stream per count call (one partition round of a 32 GB arena)

bool device_scan_applies(const char* path, uint32_t trim5p, uint64_t* size_out, uint8_t* first_byte) {
    if (g_scan_off || trim5p) return false;                       // (a 5' trim swallows line starts the way is.ignore does: the host machine's job)
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if ((uint64_t)st.st_size < (g_test_scan_batch || g_test_strip_segment ? 1 : g_scan_min_bytes)) return false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t head[2] = {0, 0};
    const bool got = pread(fd, head, 2, 0) == 2;
    ::close(fd);
    if (!got || (head[0] == 0x1f && head[1] == 0x8b) || (head[0] != '>' && head[0] != '@')) return false;    // gzip streams; other formats: the streaming path words the error
    if (size_out) *size_out = (uint64_t)st.st_size;
    if (first_byte) *first_byte = head[0];
    return true;
}

// the partition arena a file of this kind will ask for through this path (katgpu_count allocates it beside the feeders)
size_t scan_arena_bytes(uint8_t first_byte) { return first_byte == '@' && g_fastq_strip ? (size_t)32 << 30 : (size_t)16 << 30; }

namespace {

struct RawFeeder {
    katgpu_table* t; katgpu_ctx* c;
    const char* path; int fd = -1;
    uint64_t size = 0;
    ScanType type = SCAN_FASTQ;
    size_t batch = 0, segment = 0, overlap = 0, buf_bytes = 0;
    // Sharding over the ranks of a multi-GPU run (FASTQ only): rank r takes batches r, r + world, ...  A chunk's two ends are found
    // from the file bytes around its batch's nominal ends by ONE function (kg_ingest: find_record_start) -- the rank that owns the
    // batch before computes the same cut from the same bytes -- so the chunks tile the file without talking to each other.
    int shard_rank = 0, shard_world = 1;
    std::vector<uint64_t> my;                                     // the batches this feeder processes, in order
    static constexpr size_t PRE = 64;                             // bytes read in front of a batch: a record that starts exactly at the nominal cut needs its '\n'
    static constexpr size_t HEAD = 64;                            // carry area in front of an output buffer (k - 1 <= 62 bytes), keeps the payload 16-byte aligned
    // two batch buffers on the device: raw bytes, output; on the host only the bytes around a batch's two ends (where the cuts are looked for)
    std::vector<uint8_t> edge_lo[2], edge_hi[2];                  // file bytes [base, base + PRE + overlap) and [nominal end - 1, hi_read) of the batch in buffer i
    uint8_t* raw[2] = {nullptr, nullptr};
    uint8_t* acc[2] = {nullptr, nullptr};                         // accumulation buffers of base stream: HEAD + acc_bytes each
    size_t acc_bytes = 0, acc_fill = 0; int acc_cur = 0;
    uint8_t* raw_al = nullptr;                                    // a chunk starts at a record, i.e. anywhere: the scan reads it from a 16-byte aligned copy (one D2D copy, ~0.5 ms per GiB)
    // the scan's arrays (one set: scans of successive batches are serial on the compute stream)
    uint32_t *tile_cnt = nullptr, *NL = nullptr, *len_off = nullptr, *line_tile_sum = nullptr;
    uint64_t *tile_off = nullptr, *line_tile_off = nullptr;
    unsigned long long* flags = nullptr;
    uint64_t cap_lines = 0;
    // readers
    std::vector<std::thread> readers;
    std::mutex mu; std::condition_variable cv;
    uint64_t next_seg = 0;                                        // segment counter over my batches (my[j] = segments [j * spb, (j + 1) * spb))
    uint64_t n_batches = 0, spb = 0;                              // batches of the file; segments per batch (a batch's first one also reads PRE, its last one the overlap)
    std::vector<uint32_t> done_segs;                              // per my[j]: segments read and enqueued
    uint64_t consumed = 0;                                        // of my batches, how many the main thread is through with: my[j] may be read when j < consumed + 2
    bool stop = false, io_error = false;
    size_t n_readers = 0;
    std::atomic<uint64_t> us_pread{0}, us_h2d{0};                 // summed over the reader threads: in pread / in their H2D copy (enqueue + landing)
    const uint8_t* map = nullptr;                                 // the file, mapped (tmpfs: see g_scan_mmap); null: pread
    // ---- host strip (FASTQ): segments of the file stripped to their sequence lines by the readers, committed in file order ----
    bool strip = false;
    struct StripSeg { int state = 0 /* 0: not yet, 1: stripped, 2: not plain FASTQ / no certain cut, 3: read error */; uint64_t lo = 0, hi = 0; uint8_t* pin = nullptr; size_t out_n = 0; unsigned reader = 0; int half = 0; };
    std::vector<StripSeg> ssegs;                                  // one per segment of `my_segs`
    std::vector<uint64_t> my_segs;                                // the file segments this feeder takes (all of them; a rank's share when sharded)
    uint64_t n_fsegs = 0, committed = 0, next_sseg = 0;           // segments of the file; of mine, how many the caller's thread has taken / the readers have claimed
    std::vector<int> half_state;                                  // [2 * reader + half]: 0 free, 1 stripped and waiting for its copy to be issued, 2 copy issued (event recorded)
    std::vector<hipEvent_t> half_ev;

    RawFeeder(katgpu_table* t_, const char* p) : t(t_), c(t_->ctx), path(p) {}
    ~RawFeeder() { shutdown(); release(); }

    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& th : readers) if (th.joinable()) th.join();
        readers.clear();
    }
    void release() {                                              // (the buffers stay with the context: scan_cache_release)
        for (auto st : c->scan.seg_stream) hipStreamSynchronize(st);
        for (auto e : half_ev) if (e) hipEventDestroy(e);
        half_ev.clear();
        if (map) { munmap(const_cast<uint8_t*>(map), (size_t)size); map = nullptr; }     // (cheap: the readers dropped their page-table entries as they went)
        if (fd >= 0) { ::close(fd); fd = -1; }
    }
    // the context's cached buffers, (re)made when this file wants bigger ones
    int acquire(int nb, unsigned threads) {
        katgpu_ctx::ScanCache& sc = c->scan;
        const size_t seg_bytes = PRE + segment + overlap + 64;
        // part A: what every mode needs -- the accumulation buffers, two pinned segments and a stream per reader
        if (sc.pin_seg_bytes < seg_bytes || sc.pin_seg.size() < 2 * (size_t)threads || sc.acc_bytes < acc_bytes) {
            scan_cache_release(c);
            sc.pin_seg_bytes = seg_bytes; sc.acc_bytes = acc_bytes;
            for (int i = 0; i < 2; ++i) HIPCHK(c, hipMalloc((void**)&sc.acc[i], HEAD + acc_bytes + 256));      // (+ 256 bytes behind the buffer that nothing counts: the reader streams' warm-up writes land there)
            for (unsigned i = 0; i < threads; ++i) {              // two pinned segments and one stream per reader; the readers pin their own
                hipStream_t st = nullptr;                         // segments when they start (read_loop): 32 x 8 MiB pinned one after the other cost 80 ms
                sc.pin_seg.push_back(nullptr); sc.pin_seg.push_back(nullptr);
                HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                sc.seg_stream.push_back(st);
            }
        }
        // part B: the device scan's own -- raw batch buffers and the scan's arrays (a FASTQ file stripped on the host needs none of them)
        if (nb && (sc.buf_bytes < buf_bytes || sc.n_buf < nb)) {
            for (int i = 0; i < 2; ++i) { hipFree(sc.raw[i]); sc.raw[i] = nullptr; }
            hipFree(sc.raw_al); hipFree(sc.tile_cnt); hipFree(sc.NL); hipFree(sc.len_off); hipFree(sc.line_tile_sum); hipFree(sc.tile_off); hipFree(sc.line_tile_off); hipFree(sc.flags);
            sc.raw_al = nullptr; sc.tile_cnt = sc.NL = sc.len_off = sc.line_tile_sum = nullptr; sc.tile_off = sc.line_tile_off = nullptr; sc.flags = nullptr;
            sc.buf_bytes = 0; sc.n_buf = 0;
            for (int i = 0; i < nb; ++i) {
                HIPCHK(c, hipMalloc((void**)&sc.raw[i], buf_bytes));
            }
            HIPCHK(c, hipMalloc((void**)&sc.raw_al, buf_bytes));
            const uint64_t n_tiles = (buf_bytes + SC_TILE - 1) / SC_TILE;
            sc.cap_lines = buf_bytes / 16 + 4096;
            HIPCHK(c, hipMalloc((void**)&sc.tile_cnt, (n_tiles + 1) * 4));
            HIPCHK(c, hipMalloc((void**)&sc.tile_off, (n_tiles + 2) * 8));
            HIPCHK(c, hipMalloc((void**)&sc.NL, sc.cap_lines * 4));
            HIPCHK(c, hipMalloc((void**)&sc.len_off, sc.cap_lines * 4));
            HIPCHK(c, hipMalloc((void**)&sc.line_tile_sum, (sc.cap_lines / SC_BLOCK + 2) * 4));
            HIPCHK(c, hipMalloc((void**)&sc.line_tile_off, (sc.cap_lines / SC_BLOCK + 3) * 8));
            HIPCHK(c, hipMalloc((void**)&sc.flags, SCF_WORDS * 8));
            sc.buf_bytes = buf_bytes; sc.n_buf = nb;
        }
        for (int i = 0; i < 2; ++i) { raw[i] = sc.raw[i]; acc[i] = sc.acc[i]; }
        acc_bytes = sc.acc_bytes;
        raw_al = sc.raw_al; tile_cnt = sc.tile_cnt; tile_off = sc.tile_off; NL = sc.NL; len_off = sc.len_off;
        line_tile_sum = sc.line_tile_sum; line_tile_off = sc.line_tile_off; flags = sc.flags; cap_lines = sc.cap_lines;
        if (nb) buf_bytes = sc.buf_bytes;                         // (possibly larger than asked for: only ever a bound)
        return KATGPU_OK;
    }

    double setup_ms = 0;                                  // open + map + device / pinned buffers: before the file's pass (in its timing line)
    int setup(uint64_t file_size, uint8_t first, int rank, int world) {
        const double t_setup = now_ms();
        struct SetupTime { RawFeeder* f; double t0; ~SetupTime() { f->setup_ms = now_ms() - t0; } } setup_time{this, t_setup};
        size = file_size;
        type = first == '@' ? SCAN_FASTQ : SCAN_FASTA;
        shard_rank = rank; shard_world = world;
        // FASTQ: stripped on the host (the old test hooks keep the device scan they were written for)
        strip = type == SCAN_FASTQ && g_fastq_strip && (!g_test_scan_batch || g_test_strip_segment);
        batch = g_test_scan_batch ? g_test_scan_batch : g_scan_batch;
        segment = g_test_scan_segment ? g_test_scan_segment : std::min(g_scan_segment, batch);
        overlap = g_test_scan_overlap ? g_test_scan_overlap : g_scan_overlap_dflt;
        if (strip && g_test_strip_segment) { segment = g_test_strip_segment; batch = std::max<size_t>(batch == g_scan_batch ? 4 * segment : batch, segment); overlap = g_test_strip_overlap ? g_test_strip_overlap : overlap; }
        batch = std::max<size_t>(batch, 64);
        segment = std::max<size_t>(16, std::min(segment, batch));
        batch = (batch + segment - 1) / segment * segment;
        batch = std::min<size_t>(batch, (size_t)((size + segment - 1) / segment * segment));       // a small file: one batch of its own size
        spb = batch / segment;
        n_batches = (size + batch - 1) / batch;
        for (uint64_t b = (uint64_t)rank; b < n_batches; b += (uint64_t)world) my.push_back(b);
        buf_bytes = PRE + batch + overlap + 64;
        done_segs.assign(my.size(), 0);
        if (my.empty()) return KATGPU_OK;
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(c, KATGPU_ERR_IO, "Could not find input file at: %s", path);
        {
            struct statfs sfs;
            const bool tmpfs = fstatfs(fd, &sfs) == 0 && (unsigned long)sfs.f_type == 0x01021994UL /* TMPFS_MAGIC */;
            if (g_scan_mmap == 1 || (g_scan_mmap < 0 && tmpfs)) {
                void* p = mmap(nullptr, (size_t)size, PROT_READ, MAP_SHARED, fd, 0);
                if (p != MAP_FAILED) { map = (const uint8_t*)p; madvise(p, (size_t)size, MADV_SEQUENTIAL); }
            }
        }
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(g_scan_threads, (size / world + segment - 1) / segment));
        // accumulation buffers: what the file will give (FASTQ: the sequence lines, a bit under half of it), within KATGPU_SCAN_ACC_MB each
        acc_bytes = std::max<size_t>(buf_bytes, std::min<size_t>(g_scan_acc, (size_t)((double)size / world * (type == SCAN_FASTQ ? 0.6 : 1.02)) + ((size_t)1 << 20)));
        if (g_test_scan_batch) acc_bytes = std::max<size_t>(buf_bytes, (size_t)hook_u64("KATGPU_TEST_SCAN_ACC", 3 * buf_bytes));
        if (strip) {                                              // the file's segments, dealt to the ranks a batch's worth at a time (as the batches are)
            n_fsegs = (size + segment - 1) / segment;
            for (uint64_t sg = 0; sg < n_fsegs; ++sg) if ((sg / spb) % (uint64_t)world == (uint64_t)rank) my_segs.push_back(sg);
            ssegs.assign(my_segs.size(), StripSeg{});
            const size_t seg_out = (segment + overlap) / 2 + 64;          // a segment's sequence lines: at most half of its bytes
            acc_bytes = std::max<size_t>(2 * seg_out, std::min<size_t>(getenv("KATGPU_SCAN_ACC_MB") ? g_scan_acc : g_strip_acc, (size_t)((double)size / world * 0.6) + ((size_t)1 << 20)));
            if (g_test_strip_segment) acc_bytes = std::max<size_t>(2 * seg_out, (size_t)hook_u64("KATGPU_TEST_SCAN_ACC", 3 * seg_out));
        }
        const double t_acq = now_ms();
        { int rc = acquire(strip ? 0 : (my.size() > 1 ? 2 : 1), T); c->scan_waiting.store(0, std::memory_order_release); if (rc) { scan_cache_release(c); return rc; } }   // (a half-made cache must not pass for a whole one)
        if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] device scan buffers: %.0f ms\n", since_load(), now_ms() - t_acq);
        n_readers = T;
        if (strip) {
            half_state.assign(2 * (size_t)T, 0);
            half_ev.assign(2 * (size_t)T, nullptr);
            for (auto& e : half_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (unsigned i = 0; i < T; ++i) readers.emplace_back([this, i] { read_loop_strip(i); });
            return KATGPU_OK;
        }
        for (unsigned i = 0; i < T; ++i) readers.emplace_back([this, i] { read_loop(i); });
        return KATGPU_OK;
    }

    // ---- host strip: a reader takes the next segment of the file, strips it into one of its two pinned halves and reports; the
    // caller's thread (run_strip) issues the copies in file order.  A half is the reader's again when its copy has landed. ----
    void read_loop_strip(unsigned me) {
        hipSetDevice(c->device);
        bool ok_pin = true;
        for (int h = 0; h < 2; ++h)
            if (!c->scan.pin_seg[2 * me + h] &&
                hipHostMalloc((void**)&c->scan.pin_seg[2 * me + h], c->scan.pin_seg_bytes, hipHostMallocNonCoherent) != hipSuccess) { c->scan.pin_seg[2 * me + h] = nullptr; ok_pin = false; }
        // The stream's first submission makes its hardware queue (~10 ms): paid here, by all readers at once, and not by the caller's
        // thread on its first sixteen copies one after the other (measured: 180 ms of a first file's pass).
        // (the write goes BEHIND the accumulation buffer, into bytes of its own: the buffer's HEAD is written by the copy stream -- the carry of the
        // stretch before, up to k - 1 = 62 bytes -- and a warm-up that lands late must not race with it)
        if (c->scan.acc[0] && hipMemsetAsync(c->scan.acc[0] + HEAD + c->scan.acc_bytes, 'N', 16, c->scan.seg_stream[me]) == hipSuccess) hipStreamSynchronize(c->scan.seg_stream[me]);
        std::vector<uint8_t> tmp;                                 // pread mode: the segment's bytes (a mapping is read in place)
        for (int h = 0;; h ^= 1) {
            // this half free again?  (1: stripped, the caller's thread has not taken it yet; 2: its copy is on its way)
            int hs;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || io_error || half_state[2 * me + h] != 1; });
                if (stop || io_error) break;
                hs = half_state[2 * me + h];
            }
            if (hs == 2) {                                        // (only this thread moves a half out of state 2)
                const double t0 = now_ms();
                const bool landed = hipEventSynchronize(half_ev[2 * me + h]) == hipSuccess;
                us_h2d += (uint64_t)((now_ms() - t0) * 1e3);
                std::lock_guard<std::mutex> lk(mu);
                if (!landed) { io_error = true; cv.notify_all(); break; }
                half_state[2 * me + h] = 0;
            }
            uint64_t j;
            {
                std::unique_lock<std::mutex> lk(mu);
                // (not too far ahead of the caller's thread: it takes the segments in order, and every reader holds at most two)
                cv.wait(lk, [&] { return stop || io_error || next_sseg >= my_segs.size() || next_sseg < committed + 4 * n_readers; });
                if (stop || io_error || next_sseg >= my_segs.size()) break;
                j = next_sseg++;
            }
            StripSeg sg;
            sg.reader = me; sg.half = h; sg.pin = c->scan.pin_seg[2 * me + h];
            const uint64_t f0 = my_segs[j] * (uint64_t)segment, f1 = std::min<uint64_t>(size, f0 + segment);
            const uint64_t w0 = f0 ? f0 - 1 : 0, w1 = std::min<uint64_t>(size, f1 + overlap);      // the bytes the two cuts are looked for in ('\n' before a record start included)
            const double ta = now_ms();
            const uint8_t* buf = nullptr;
            sg.state = 1;
            if (!ok_pin) sg.state = 3;
            else if (map) {
                struct stat st0;                                  // (a file cut short under the run: KATGPU_ERR_IO between segments, SIGBUS inside one -- INTEGRATION.md)
                if (fstat(fd, &st0) != 0 || (uint64_t)st0.st_size < w1) sg.state = 3; else buf = map + w0;
                // the segment's page-table entries in ONE call instead of a fault per sixteen pages (the faults of sixteen readers into one address
                // space were two thirds of a reader's time: DESIGN.md section 4)
                if (sg.state == 1 && g_scan_populate) { const uint64_t a0 = w0 & ~4095ULL; (void)madvise(const_cast<uint8_t*>(map) + a0, (size_t)(w1 - a0), 22 /* MADV_POPULATE_READ, Linux 5.14 (an older kernel says EINVAL: the faults do it) */); }
            } else {
                tmp.resize((size_t)(w1 - w0));
                uint64_t got = 0;
                while (got < w1 - w0) {
                    const ssize_t r = pread(fd, tmp.data() + got, (size_t)std::min<uint64_t>(w1 - w0 - got, (uint64_t)1 << 30), (off_t)(w0 + got));
                    if (r <= 0) { sg.state = 3; break; }
                    got += (uint64_t)r;
                }
                buf = tmp.data();
            }
            if (sg.state == 1) {
                const int64_t lo = f0 == 0 ? 0 : kg::find_record_start(kg::ParseState::FASTQ, buf, (int64_t)w0, (int64_t)(w1 - w0), (int64_t)f0);
                const int64_t hi = f1 >= size ? (int64_t)size : kg::find_record_start(kg::ParseState::FASTQ, buf, (int64_t)w0, (int64_t)(w1 - w0), (int64_t)f1);
                if (lo < 0 || hi < 0 || lo > hi || (uint64_t)(hi - lo) / 2 + 1 > c->scan.pin_seg_bytes) sg.state = 2;
                else {
                    sg.lo = (uint64_t)lo; sg.hi = (uint64_t)hi;
                    if (!kg::strip_fastq_records(buf + (sg.lo - w0), (size_t)(sg.hi - sg.lo), sg.pin, &sg.out_n)) sg.state = 2;
                }
                if (map) {
                    struct stat st0;
                    if (fstat(fd, &st0) != 0 || (uint64_t)st0.st_size < w1) sg.state = 3;
                    // this reader is through with these pages: their page-table entries go now (see read_loop)
                    const uint64_t a0 = (f0 + 4095) & ~4095ULL, a1 = f1 & ~4095ULL;
                    if (a1 > a0) madvise(const_cast<uint8_t*>(map) + a0, (size_t)(a1 - a0), MADV_DONTNEED);
                }
            }
            us_pread += (uint64_t)((now_ms() - ta) * 1e3);
            {
                std::lock_guard<std::mutex> lk(mu);
                ssegs[j] = sg;
                half_state[2 * me + h] = 1;
            }
            cv.notify_all();
        }
        // (copies still in flight from this reader's halves: the caller's thread synchronises the readers' streams before it moves on)
    }

    // file range of batch b as read: [base, hi_read) = [b * batch - PRE, min(size, (b + 1) * batch + overlap)); buffer byte 0 = file byte base
    uint64_t batch_base(uint64_t b) const { return b ? b * (uint64_t)batch - PRE : 0; }
    uint64_t batch_hi_read(uint64_t b) const { return std::min<uint64_t>(size, (b + 1) * (uint64_t)batch + overlap); }

    // A reader alternates between its two pinned segments: while the copy of one is on its way to the device the thread is already
    // in pread for the other (one copy and one pread in flight per reader); a segment counts as done -- its pinned memory free again,
    // its bytes the main thread's to scan -- when its copy has landed, which the thread learns one segment later.
    void read_loop(unsigned me) {
        hipSetDevice(c->device);
        bool ev_ok = true;
        for (int h = 0; h < 2; ++h)                               // (this reader's slots of the context's cache: nobody else touches them)
            if (!c->scan.pin_seg[2 * me + h] &&                   // written by the CPU, read by the copy engine: the coarse-grained kind copies a quarter faster (tools/reader_bench.hip)
                hipHostMalloc((void**)&c->scan.pin_seg[2 * me + h], c->scan.pin_seg_bytes, hipHostMallocNonCoherent) != hipSuccess) { c->scan.pin_seg[2 * me + h] = nullptr; ev_ok = false; }
        uint8_t* const seg_mem[2] = {c->scan.pin_seg[2 * me], c->scan.pin_seg[2 * me + 1]};
        const hipStream_t st = c->scan.seg_stream[me];
        hipEvent_t landed[2] = {nullptr, nullptr};
        ev_ok = ev_ok && hipEventCreateWithFlags(&landed[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&landed[1], hipEventDisableTiming) == hipSuccess;
        int64_t in_flight[2] = {-1, -1};                          // my-batch index j of the copy in flight from each segment
        auto retire = [&](int h, bool ok) {                        // the copy from segment h has landed (or failed)
            if (in_flight[h] < 0) return;
            const double t0 = now_ms();
            if (ok && hipEventSynchronize(landed[h]) != hipSuccess) ok = false;
            us_h2d += (uint64_t)((now_ms() - t0) * 1e3);
            { std::lock_guard<std::mutex> lk(mu); if (!ok) io_error = true; ++done_segs[(size_t)in_flight[h]]; }
            in_flight[h] = -1;
            cv.notify_all();
        };
        for (int h = 0;; h ^= 1) {
            uint64_t seg;
            {
                std::unique_lock<std::mutex> lk(mu);
                // (a reader that would have to wait for the main thread first lets its copy in flight land: the main thread may be waiting for exactly that segment)
                while (!(stop || io_error || next_seg >= my.size() * spb || next_seg / spb < consumed + 2)) {
                    if (in_flight[0] >= 0 || in_flight[1] >= 0) { lk.unlock(); retire(h, ev_ok); retire(h ^ 1, ev_ok); lk.lock(); continue; }
                    cv.wait(lk);
                }
                if (stop || io_error || next_seg >= my.size() * spb) break;
                seg = next_seg++;
            }
            retire(h, ev_ok);                                       // this segment's previous copy: long landed, normally
            uint8_t* const mine = seg_mem[h];
            const uint64_t j = seg / spb, s = seg % spb, b = my[j];
            const int buf = (int)(j & 1);
            // segment s of batch b: [b * batch + s * segment, ... + segment); the first one also reads PRE, the last one the overlap
            const uint64_t f0 = s ? b * (uint64_t)batch + s * (uint64_t)segment : batch_base(b);
            const uint64_t f1 = s + 1 == spb ? batch_hi_read(b) : std::min<uint64_t>(size, b * (uint64_t)batch + (s + 1) * (uint64_t)segment);
            bool ok = ev_ok;
            if (ok && f0 < f1) {
                const double ta = now_ms();
                uint64_t got = 0;
                if (map) {
                    // (a mapping of a file that SHRINKS under the run faults with SIGBUS where pread returns short: the file's size is looked at
                    // before and after every segment, so a truncation between segments is an I/O error like on the pread path; one that lands
                    // inside the copy of a segment still kills the process -- INTEGRATION.md section 5 says so, KATGPU_SCAN_MMAP=0 avoids it)
                    struct stat st0;
                    if (fstat(fd, &st0) != 0 || (uint64_t)st0.st_size < f1) { ok = false; }
                    else {
                        memcpy(mine, map + f0, (size_t)(f1 - f0)); got = f1 - f0;
                        if (fstat(fd, &st0) != 0 || (uint64_t)st0.st_size < f1) ok = false;
                    }
                    if (!ok) got = f1 - f0;                       // (no pread retry of a file that is being cut)
                    // this reader is through with these pages: their page-table entries go now, a segment at a time and on every reader
                    // at once (the pages stay in the page cache) -- left in place, the 8 M entries of a 32 GB run are taken down by one
                    // thread when the mapping or the process ends: 0.2 s per file under the mmap lock, or 0.6 s at exit (measured)
                    const uint64_t a0 = (f0 + 4095) & ~4095ULL, a1 = f1 & ~4095ULL;
                    if (a1 > a0) madvise(const_cast<uint8_t*>(map) + a0, (size_t)(a1 - a0), MADV_DONTNEED);
                }
                while (got < f1 - f0) {
                    const ssize_t r = pread(fd, mine + got, (size_t)std::min<uint64_t>(f1 - f0 - got, (uint64_t)1 << 30), (off_t)(f0 + got));
                    if (r <= 0) { ok = false; break; }
                    got += (uint64_t)r;
                }
                const double tb = now_ms();
                us_pread += (uint64_t)((tb - ta) * 1e3);
                if (ok && (hipMemcpyAsync(raw[buf] + (f0 - batch_base(b)), mine, (size_t)(f1 - f0), hipMemcpyHostToDevice, st) != hipSuccess ||
                           hipEventRecord(landed[h], st) != hipSuccess)) ok = false;
                us_h2d += (uint64_t)((now_ms() - tb) * 1e3);
                if (ok) { in_flight[h] = (int64_t)j; continue; }
            }
            {                                                       // an empty segment, or an error: nothing in flight from it
                std::lock_guard<std::mutex> lk(mu);
                if (!ok) io_error = true;
                ++done_segs[j];
            }
            cv.notify_all();
        }
        retire(0, ev_ok); retire(1, ev_ok);
        for (int h = 0; h < 2; ++h) if (landed[h]) hipEventDestroy(landed[h]);
    }

    // wait until every segment of batch b is in pinned memory and its copy enqueued, then until the copies have landed
    int wait_batch(uint64_t j) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return io_error || done_segs[j] == spb; });
            if (io_error) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
        }
        // the bytes around the batch's two ends, for the cuts (the page cache has just served them to the readers)
        const uint64_t b = my[j], base = batch_base(b), hi_read = batch_hi_read(b);
        const int buf = (int)(j & 1);
        auto grab = [&](std::vector<uint8_t>& v, uint64_t f0, uint64_t f1) -> bool {
            v.resize((size_t)(f1 > f0 ? f1 - f0 : 0));
            uint64_t got = 0;
            while (got < v.size()) { const ssize_t r = pread(fd, v.data() + got, v.size() - got, (off_t)(f0 + got)); if (r <= 0) return false; got += (uint64_t)r; }
            return true;
        };
        const uint64_t nominal = std::min<uint64_t>(size, (b + 1) * (uint64_t)batch);
        if (!grab(edge_lo[buf], base, std::min<uint64_t>(hi_read, base + PRE + overlap)) || !grab(edge_hi[buf], nominal ? nominal - 1 : 0, hi_read))
            return fail(c, KATGPU_ERR_IO, "read error on %s", path);
        return KATGPU_OK;
    }
    void batch_consumed() {
        { std::lock_guard<std::mutex> lk(mu); ++consumed; }
        cv.notify_all();
    }

    // The record scan of raw[buf][lo, hi), on the context's second stream (the counting worker owns the first).  Phase 1 (measure): lines,
    // structure checks, output size.  *valid: the device vouches for the chunk; *out_n: bytes of base stream it will give.
    struct Measured { const uint8_t* src; uint64_t n, n_lines; uint32_t n_tiles; };
    int scan_measure(int buf, uint64_t lo, uint64_t hi, bool* valid, uint64_t* out_n, Measured* ms) {
        const hipStream_t st = c->copy_stream;
        const uint8_t* src = raw[buf] + lo;
        const uint64_t n = hi - lo;
        *valid = false; *out_n = 0;
        ms->src = src; ms->n = n; ms->n_lines = 0; ms->n_tiles = 0;
        if (n == 0) { *valid = true; return KATGPU_OK; }
        if (reinterpret_cast<uintptr_t>(src) & 15) {
            HIPCHK(c, hipMemcpyAsync(raw_al, src, n, hipMemcpyDeviceToDevice, st));
            src = raw_al;
        }
        const uint32_t n_tiles = (uint32_t)((n + SC_TILE - 1) / SC_TILE);
        const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)c->n_cu * 16);
        { std::lock_guard<std::mutex> lk(c->prof_mu); c->prof_launches[KATGPU_K_SCAN] += 1; c->prof_units[KATGPU_K_SCAN] += n; }
        HIPCHK(c, hipMemsetAsync(flags, 0, SCF_WORDS * 8, st));
        hipLaunchKernelGGL(k_nl_count, dim3(grid), dim3(SC_BLOCK), 0, st, src, n, n_tiles, tile_cnt, flags);
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)tile_cnt, n_tiles, (uint64_t)n_tiles, (const uint64_t*)nullptr, tile_off,
                           (uint64_t)(n_tiles + 1), 1, (unsigned long long*)&flags[SCF_LINES]);
        unsigned long long h[SCF_WORDS];
        HIPCHK(c, hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        const uint64_t n_lines = h[SCF_LINES];
        if (h[SCF_BAD] || n_lines == 0 || n_lines > cap_lines || (type == SCAN_FASTQ && (n_lines & 3))) return KATGPU_OK;     // the host's
        const uint64_t n_ltiles = (n_lines + SC_BLOCK - 1) / SC_BLOCK;
        const int lgrid = (int)std::min<uint64_t>(n_ltiles, (uint64_t)c->n_cu * 16);
        hipLaunchKernelGGL(k_nl_write, dim3(grid), dim3(SC_BLOCK), 0, st, src, n, n_tiles, (const uint64_t*)tile_off, NL, cap_lines);
        if (type == SCAN_FASTQ) hipLaunchKernelGGL(k_line_len<SCAN_FASTQ>, dim3(lgrid), dim3(SC_BLOCK), 0, st, src, (const uint32_t*)NL, n_lines, len_off, line_tile_sum, flags);
        else hipLaunchKernelGGL(k_line_len<SCAN_FASTA>, dim3(lgrid), dim3(SC_BLOCK), 0, st, src, (const uint32_t*)NL, n_lines, len_off, line_tile_sum, flags);
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)line_tile_sum, (uint32_t)n_ltiles, (uint64_t)n_ltiles, (const uint64_t*)nullptr,
                           line_tile_off, (uint64_t)(n_ltiles + 1), 1, (unsigned long long*)&flags[SCF_OUT]);
        HIPCHK(c, hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (h[SCF_BAD] || h[SCF_OUT] > buf_bytes) return KATGPU_OK;
        hipLaunchKernelGGL(k_line_off, dim3(lgrid), dim3(SC_BLOCK), 0, st, n_lines, len_off, (const uint64_t*)line_tile_off);
        ms->src = src; ms->n_lines = n_lines; ms->n_tiles = n_tiles;
        *valid = true; *out_n = h[SCF_OUT];
        return KATGPU_OK;
    }
    // Phase 2 (emit): the chunk's base stream to dst (asynchronous on the second stream)
    int scan_emit(const Measured& ms, uint8_t* dst) {
        if (!ms.n_lines) return KATGPU_OK;
        const hipStream_t st = c->copy_stream;
        const int grid = (int)std::min<uint64_t>(ms.n_tiles, (uint64_t)c->n_cu * 16);
        if (type == SCAN_FASTQ) hipLaunchKernelGGL(k_emit<SCAN_FASTQ>, dim3(grid), dim3(SC_BLOCK), 0, st, ms.src, ms.n, ms.n_tiles, (const uint64_t*)tile_off, (const uint32_t*)NL,
                                                   (const uint32_t*)len_off, ms.n_lines, dst);
        else hipLaunchKernelGGL(k_emit<SCAN_FASTA>, dim3(grid), dim3(SC_BLOCK), 0, st, ms.src, ms.n, ms.n_tiles, (const uint64_t*)tile_off, (const uint32_t*)NL,
                                (const uint32_t*)len_off, ms.n_lines, dst);
        HIPCHK(c, hipGetLastError());
        return KATGPU_OK;
    }

    // ---- the counting worker: a full accumulation buffer is a resident stretch of the base stream; it is counted (partition rounds
    // sized like a resident input's: the table is swept once per GBs of stream, not once per batch) while the next one fills ----
    std::thread worker;
    std::mutex wmu; std::condition_variable wcv;
    struct Job { int a; size_t n; };
    std::deque<Job> jobs;
    bool acc_busy[2] = {false, false}, wstop = false;
    int worker_rc = KATGPU_OK; std::string worker_err;
    double worker_ms = 0;
    void work() {
        hipSetDevice(c->device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return wstop || !jobs.empty(); });
                if (jobs.empty()) return;
                j = jobs.front(); jobs.pop_front();
            }
            const double t0 = now_ms();
            int rc = worker_rc ? worker_rc : table_wait(t);                                // (katgpu_count allocates the table beside us)
            if (!rc) rc = count_resident(t, acc[j.a], j.n);                                // (after an error the queued buffers are dropped)
            {
                std::lock_guard<std::mutex> lk(wmu);
                if (rc && !worker_rc) { worker_rc = rc; worker_err = c->err; }
                acc_busy[j.a] = false;
                worker_ms += now_ms() - t0;
            }
            wcv.notify_all();
        }
    }
    void stop_worker() {
        if (!worker.joinable()) return;
        { std::lock_guard<std::mutex> lk(wmu); wstop = true; }
        wcv.notify_all();
        worker.join();
    }
    // hand acc[acc_cur] (HEAD + acc_fill bytes) to the worker and open the other buffer with the stream's last k-1 bytes in its head
    int submit(bool last) {
        const uint32_t k = t->dv.k;
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));          // every chunk emitted into it has landed
        const int other = acc_cur ^ 1;
        {
            std::unique_lock<std::mutex> lk(wmu);
            if (acc_fill) { acc_busy[acc_cur] = true; jobs.push_back({acc_cur, HEAD + acc_fill}); }
            wcv.notify_all();
            wcv.wait(lk, [&] { return last ? (!acc_busy[0] && !acc_busy[1] && jobs.empty()) : !acc_busy[other]; });
            if (worker_rc) return fail(c, worker_rc, "%s", worker_err.c_str());
        }
        if (last) return KATGPU_OK;
        HIPCHK(c, hipMemsetAsync(acc[other], 'N', HEAD, c->copy_stream));
        if (acc_fill) {
            const uint32_t cn = (uint32_t)std::min<uint64_t>(acc_fill, k - 1);
            HIPCHK(c, hipMemcpyAsync(acc[other] + HEAD - cn, acc[acc_cur] + HEAD + acc_fill - cn, cn, hipMemcpyDeviceToDevice, c->copy_stream));
        }
        acc_cur = other; acc_fill = 0;
        return KATGPU_OK;
    }

    // the rest of the file from offset `from` (a record start: FASTQ; a line start: FASTA) through the host state machine
    int host_rest(uint64_t from, const uint8_t* carry, uint32_t carry_n) {
        kg::ParseState ps;
        ps.begin(type == SCAN_FASTQ ? '@' : '>');
        uint8_t first = 0;
        if (from < size && pread(fd, &first, 1, (off_t)from) != 1) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
        if (type == SCAN_FASTA && first != '>') ps.st = kg::ParseState::LOOP_CHECK;        // in the middle of a record's sequence lines
        std::vector<uint8_t> rawb((size_t)16 << 20), outb;
        outb.reserve(rawb.size() + 64);
        outb.insert(outb.end(), carry, carry + carry_n);
        int rc = KATGPU_OK;
        for (uint64_t off = from; off < size && rc == KATGPU_OK;) {
            const ssize_t r = pread(fd, rawb.data(), rawb.size(), (off_t)off);
            if (r <= 0) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
            bool bad = false;
            ps.consume(rawb.data(), (size_t)r, outb, &bad);
            if (bad) return fail(c, KATGPU_ERR_FASTQ, "Invalid fastq sequence");
            off += (uint64_t)r;
            if (outb.size() >= ((size_t)64 << 20) || off >= size) {
                // (katgpu_count_bases_host would reset the table's carry: feed through the same rings by hand)
                rc = count_host_stream(outb.data(), outb.size());
                const size_t keep = std::min<size_t>(outb.size(), t->dv.k - 1);
                std::vector<uint8_t> tailb(outb.end() - keep, outb.end());
                outb.assign(tailb.begin(), tailb.end());
            }
        }
        if (rc == KATGPU_OK && !ps.end_ok()) return fail(c, KATGPU_ERR_FASTQ, "Invalid fastq sequence");
        return rc;
    }
    // a host piece of the base stream -> acc[0] -> count_resident (the fall-back is rare: no pipelining; the worker is idle by then)
    int count_host_stream(const uint8_t* p, size_t n) {
        size_t pos = 0;
        const uint32_t k = t->dv.k;
        while (pos < n && n - pos >= k) {
            const size_t take = std::min(n - pos, acc_bytes);
            uint8_t head[HEAD];
            memset(head, 'N', HEAD);
            HIPCHK(c, hipMemcpyAsync(acc[0], head, HEAD, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(acc[0] + HEAD, p + pos, take, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            int rc = count_resident(t, acc[0], HEAD + take);
            if (rc) return rc;
            if (take == n - pos) break;
            pos += take - (k - 1);
        }
        return KATGPU_OK;
    }

    // first record start at or after file offset `nominal` (FASTQ), as every rank computes it: from the bytes [nominal - 1, nominal + overlap)
    int64_t fastq_cut(const uint8_t* buf, uint64_t base, uint64_t hi_read, uint64_t nominal) const {
        const uint64_t lim = std::min<uint64_t>(hi_read, nominal + overlap);
        return kg::find_record_start(kg::ParseState::FASTQ, buf, (int64_t)base, (int64_t)(lim - base), (int64_t)nominal);
    }

    // every copy the readers' streams carry has landed (before an accumulation buffer changes hands, before the fall-back)
    int flush_copies() {
        for (auto st : c->scan.seg_stream) HIPCHK(c, hipStreamSynchronize(st));
        return KATGPU_OK;
    }

    // The caller's thread of the host-strip path: the readers' segments in file order -> the accumulation buffers -> the counting worker.
    int run_strip() {
        const bool sharded = shard_world > 1;
        double ms_wait = 0, ms_issue = 0, ms_count = 0;
        const double t_run = now_ms();
        struct Report { RawFeeder* f; double *w, *s, *n, t0; ~Report() {
            if (g_timing) fprintf(stderr, "katgpu_timing {\"file\": \"%s\", \"bytes\": %llu, \"setup_ms\": %.1f, \"wall_ms\": %.1f, \"reader_wait_ms\": %.1f, \"scan_ms\": %.1f, \"counter_wait_ms\": %.1f, \"counting_ms\": %.1f, "
                                  "\"reader_threads\": %u, \"pread_ms_per_thread\": %.1f, \"h2d_ms_per_thread\": %.1f, \"segment_MiB\": %zu, \"read_by\": \"%s\"}\n",
                                  json_escaped(f->path).c_str(), (unsigned long long)f->size, f->setup_ms, now_ms() - t0, *w, *s, *n, f->worker_ms, (unsigned)f->n_readers,
                                  f->us_pread.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->us_h2d.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->segment >> 20,
                                  f->map ? "sequence lines stripped out of a mapping (tmpfs) on the host: only bases cross PCIe" : "pread, sequence lines stripped on the host: only bases cross PCIe");
            if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] host strip of %s: %.1f GB in %.0f ms (%.1f GB/s): waiting for the readers %.0f ms, issuing copies %.0f ms, waiting for the counter %.0f ms (it counted for %.0f ms); %u reader threads (read + strip %.0f ms, waiting for their copies %.0f ms each), %zu MiB segments, %zu MiB accumulated per count\n",
                                 since_load(), f->path, f->size / 1e9, now_ms() - t0, f->size / 1e6 / std::max(1.0, now_ms() - t0), *w, *s, *n, f->worker_ms, (unsigned)f->n_readers,
                                 f->us_pread.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->us_h2d.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->segment >> 20, f->acc_bytes >> 20); } } report{this, &ms_wait, &ms_issue, &ms_count, t_run};
        struct StopWorker { RawFeeder* f; ~StopWorker() { f->stop_worker(); } } stop_w{this};
        worker = std::thread([this] { work(); });
        struct Limit { katgpu_ctx* c; size_t old; ~Limit() { c->arena_limit = old; } } limit{c, c->arena_limit};
        c->arena_limit = (size_t)32 << 30;                     // one round per accumulation buffer (3 GiB of base stream: 2.6 G k-mers)
        HIPCHK(c, hipMemsetAsync(acc[0], 'N', HEAD, c->copy_stream));
        acc_cur = 0; acc_fill = 0;
        uint64_t prev_hi = 0;                                     // where the segment before ended: a proven record start
        for (uint64_t j = 0; j < my_segs.size(); ++j) {
            double t0 = now_ms();
            StripSeg sg;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return io_error || ssegs[j].state != 0; });
                if (io_error && ssegs[j].state == 0) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
                sg = ssegs[j];
            }
            ms_wait += now_ms() - t0;
            if (sg.state == 3) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
            const bool follows = j > 0 && my_segs[j] == my_segs[j - 1] + 1;
            const bool plain = sg.state == 1 && (!follows || sg.lo == prev_hi) && my_segs[j] < g_test_strip_fail_at;
            if (!plain && sharded)
                return fail(c, KATGPU_ERR_FASTQ, "%s: the bytes around offset %llu are not plain four-line FASTQ: such a file cannot be cut between GPUs -- run it on one", path,
                            (unsigned long long)(my_segs[j] * (uint64_t)segment));
            if (!plain) {
                // not plain four-line FASTQ from here on (or no certain cut): what has been committed is counted, the rest of the file goes
                // through the host state machine from the last proven record start
                const uint64_t from = j == 0 ? 0 : prev_hi;
                if (g_trace) fprintf(stderr, "[katgpu] host strip: %s is not plain four-line FASTQ around offset %llu: the host parser takes the file from offset %llu\n", path,
                                     (unsigned long long)(my_segs[j] * (uint64_t)segment), (unsigned long long)from);
                shutdown();
                int rc = flush_copies();
                if (rc) return rc;
                rc = submit(true);
                if (rc) return rc;
                stop_worker();
                rc = table_wait(t);
                if (rc) return rc;
                return host_rest(from, nullptr, 0);              // (the stream so far ended on a record's 'N': nothing to carry over)
            }
            t0 = now_ms();
            if (acc_fill + sg.out_n > acc_bytes) {               // this segment opens the other buffer
                int rc = flush_copies();
                if (!rc) rc = submit(false);
                if (rc) return rc;
            }
            ms_count += now_ms() - t0;
            t0 = now_ms();
            const hipStream_t st = c->scan.seg_stream[sg.reader];
            if (sg.out_n) HIPCHK(c, hipMemcpyAsync(acc[acc_cur] + HEAD + acc_fill, sg.pin, sg.out_n, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipEventRecord(half_ev[2 * sg.reader + sg.half], st));
            acc_fill += sg.out_n;
            prev_hi = sg.hi;
            {
                std::lock_guard<std::mutex> lk(mu);
                half_state[2 * sg.reader + sg.half] = 2;
                ++committed;
            }
            cv.notify_all();
            ms_issue += now_ms() - t0;
        }
        const double t0 = now_ms();
        int rc = flush_copies();
        if (!rc) rc = submit(true);
        ms_count += now_ms() - t0;
        return rc;
    }

    int run() {
        if (strip) return run_strip();
        const uint32_t k = t->dv.k;
        const bool sharded = shard_world > 1;
        uint64_t cut_lo = 0;                                      // file offset where the next chunk starts: a proven record / line start
        uint8_t carry[HEAD]; uint32_t carry_n = 0;                // last k-1 bytes of the base stream so far (host copy, for the fall-back)
        double ms_wait = 0, ms_scan = 0, ms_count = 0;
        const double t_run = now_ms();
        struct Report { RawFeeder* f; double *w, *s, *n, t0; ~Report() {
            if (g_timing) fprintf(stderr, "katgpu_timing {\"file\": \"%s\", \"bytes\": %llu, \"setup_ms\": %.1f, \"wall_ms\": %.1f, \"reader_wait_ms\": %.1f, \"scan_ms\": %.1f, \"counter_wait_ms\": %.1f, \"counting_ms\": %.1f, "
                                  "\"reader_threads\": %u, \"pread_ms_per_thread\": %.1f, \"h2d_ms_per_thread\": %.1f, \"segment_MiB\": %zu, \"read_by\": \"%s\"}\n",
                                  json_escaped(f->path).c_str(), (unsigned long long)f->size, f->setup_ms, now_ms() - t0, *w, *s, *n, f->worker_ms, (unsigned)f->n_readers,
                                  f->us_pread.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->us_h2d.load() / 1e3 / std::max<size_t>(1, f->n_readers), f->segment >> 20,
                                  f->map ? "memcpy out of a mapping (tmpfs)" : "pread");
            if (g_trace) fprintf(stderr, "[katgpu +%.0f ms] device scan of %s: %.1f GB in %.0f ms (%.1f GB/s): waiting for readers + H2D %.0f ms, scan %.0f ms, waiting for the counter %.0f ms (it counted for %.0f ms); %u reader threads, %zu MiB segments, %zu MiB accumulated per count\n",
                                 since_load(), f->path, f->size / 1e9, now_ms() - t0, f->size / 1e6 / std::max(1.0, now_ms() - t0), *w, *s, *n, f->worker_ms, (unsigned)f->n_readers, f->segment >> 20, f->acc_bytes >> 20); } } report{this, &ms_wait, &ms_scan, &ms_count, t_run};
        struct StopWorker { RawFeeder* f; ~StopWorker() { f->stop_worker(); } } stop_w{this};
        worker = std::thread([this] { work(); });
        struct Limit { katgpu_ctx* c; size_t old; ~Limit() { c->arena_limit = old; } } limit{c, c->arena_limit};
        c->arena_limit = (size_t)16 << 30;                     // rounds of what an accumulation buffer holds: a modest arena does (allocating 100 GB costs seconds)
        HIPCHK(c, hipMemsetAsync(acc[0], 'N', HEAD, c->copy_stream));
        acc_cur = 0; acc_fill = 0;
        for (uint64_t j = 0; j < my.size(); ++j) {
            double t0 = now_ms();
            int rc = wait_batch(j);
            if (rc) return rc;
            ms_wait += now_ms() - t0;
            const uint64_t b = my[j];
            const int buf = (int)(j & 1);
            const uint64_t lo = batch_base(b), hi_read = batch_hi_read(b);
            bool cut_ok = true;
            if (sharded) {                                        // this chunk's start, found the way the owner of batch b - 1 finds its chunk's end
                cut_lo = 0;
                if (b) { const int64_t f = fastq_cut(edge_lo[buf].data(), lo, lo + edge_lo[buf].size(), b * (uint64_t)batch); if (f < 0) cut_ok = false; else cut_lo = (uint64_t)f; }
            }
            // the chunk's end: the first record start (FASTQ) / line start (FASTA) at or after the nominal end, inside what was read
            uint64_t cut_hi = size;
            if (b + 1 < n_batches) {
                const uint64_t nominal = (b + 1) * (uint64_t)batch;
                if (type == SCAN_FASTQ) {
                    const int64_t f = fastq_cut(edge_hi[buf].data(), nominal - 1, hi_read, nominal);
                    if (f < 0) cut_ok = false; else cut_hi = (uint64_t)f;
                } else {
                    const uint8_t* nl = (const uint8_t*)memchr(edge_hi[buf].data(), '\n', edge_hi[buf].size());
                    if (!nl) cut_ok = false; else cut_hi = (nominal - 1) + (uint64_t)(nl - edge_hi[buf].data()) + 1;
                }
            } else if (size && (edge_hi[buf].empty() || edge_hi[buf].back() != '\n')) cut_ok = false;              // a last line without its newline: the host machine knows what to do
            bool valid = false; uint64_t out_n = 0;
            Measured msd{};
            t0 = now_ms();
            if (cut_ok && cut_lo <= cut_hi && b < g_test_scan_fail_at) {
                rc = scan_measure(buf, cut_lo - lo, cut_hi - lo, &valid, &out_n, &msd);
                if (rc) return rc;
            }
            ms_scan += now_ms() - t0;
            if (!valid && sharded)
                return fail(c, KATGPU_ERR_FASTQ, "%s: batch %llu is not plain four-line FASTQ (or holds a '\\r'): such a file cannot be cut between GPUs -- run it on one", path, (unsigned long long)b);
            if (!valid) {
                if (g_trace) fprintf(stderr, "[katgpu] device scan: batch %llu of %s goes to the host parser (and the rest of the file with it)\n", (unsigned long long)b, path);
                if (acc_fill) {                                   // what the stream ended on, for the windows across the hand-over
                    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
                    carry_n = (uint32_t)std::min<uint64_t>(acc_fill, k - 1);
                    HIPCHK(c, hipMemcpy(carry, acc[acc_cur] + HEAD + acc_fill - carry_n, carry_n, hipMemcpyDeviceToHost));
                } else {                                          // ... which may lie in the head of a freshly opened buffer
                    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
                    carry_n = k - 1;
                    HIPCHK(c, hipMemcpy(carry, acc[acc_cur] + HEAD - carry_n, carry_n, hipMemcpyDeviceToHost));
                }
                shutdown();
                rc = submit(true);                                // what the device has parsed is counted first; the worker is idle afterwards
                if (rc) return rc;
                stop_worker();
                rc = table_wait(t);
                if (rc) return rc;
                return host_rest(cut_lo, carry, carry_n);
            }
            t0 = now_ms();
            if (acc_fill + out_n > acc_bytes) { rc = submit(false); if (rc) return rc; }      // this chunk opens the other buffer
            ms_count += now_ms() - t0;
            t0 = now_ms();
            rc = scan_emit(msd, acc[acc_cur] + HEAD + acc_fill);
            if (rc) return rc;
            acc_fill += out_n;
            HIPCHK(c, hipStreamSynchronize(c->copy_stream));      // the scan's kernels are through with raw[buf]
            ms_scan += now_ms() - t0;
            cut_lo = cut_hi;
            batch_consumed();                                     // raw[buf] may take my batch after next
        }
        const double t0 = now_ms();
        int rc = submit(true);
        ms_count += now_ms() - t0;
        return rc;
    }
};

}  // namespace

void scan_cache_release(katgpu_ctx* c) {
    katgpu_ctx::ScanCache& sc = c->scan;
    for (auto st : sc.seg_stream) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (auto p : sc.pin_seg) if (p) hipHostFree(p);
    for (int i = 0; i < 2; ++i) { hipFree(sc.raw[i]); hipFree(sc.acc[i]); }
    hipFree(sc.raw_al); hipFree(sc.tile_cnt); hipFree(sc.NL); hipFree(sc.len_off); hipFree(sc.line_tile_sum); hipFree(sc.tile_off); hipFree(sc.line_tile_off); hipFree(sc.flags);
    sc = katgpu_ctx::ScanCache{};
}

// One large plain file through the device scan.  *took = false: not a file for this path (nothing was counted).
int count_file_device_scan(katgpu_table* t, const char* path, uint32_t trim5p, bool* took, int rank, int world) {
    *took = false;
    uint64_t size = 0; uint8_t first = 0;
    if (!device_scan_applies(path, trim5p, &size, &first)) { t->ctx->scan_waiting.store(0, std::memory_order_release); return KATGPU_OK; }
    katgpu_ctx* c = t->ctx;
    struct Lower { katgpu_ctx* c; ~Lower() { c->scan_waiting.store(0, std::memory_order_release); } } lower{c};      // (whatever way this call ends, nobody keeps waiting for its buffers)
    if (!t->alloc_thread.joinable() && c->arena && !c->arena_busy && !c->arena_borrowed) {      // (not while katgpu_count's allocation thread is at work)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < ((size_t)6 << 30)) release_arena(c);    // the cached arena holds most of the free HBM: the batch buffers come first
    }
    RawFeeder f(t, path);
    if (world > 1 && first != '@') return KATGPU_OK;              // only FASTQ is cut between ranks (the caller deals whole files otherwise)
    int rc = f.setup(size, first, rank, world);
    if (rc == KATGPU_ERR_NOMEM) { (void)hipGetLastError(); return KATGPU_OK; }      // no room for the batch buffers: the streaming path and its smaller rings
    if (rc == KATGPU_OK) { *took = true; rc = f.run(); }
    if (rc == KATGPU_OK) rc = table_wait(t);
    if (rc == KATGPU_OK) rc = refresh_counters(t);
    return rc;
}
