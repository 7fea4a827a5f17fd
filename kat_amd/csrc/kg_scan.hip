// kg_scan.hip -- katgpu_count_files' fast path for large plain FASTQ / FASTA files: the host moves raw file bytes, the device parses.
//
//   reader threads: pread 32 MiB segments of the file straight into pre-faulted PINNED memory, each followed by its own H2D copy
//                   (so reading and copying overlap inside a batch, and several copies are in flight)
//   the caller's thread, per batch of 1 GiB of file: find the record-aligned cut at the batch's end (host: two memchr), run the
//                   record scan on the device (kg_scan.hpp), read three words back, and hand the compacted base stream -- a
//                   resident buffer like any other -- to count_resident, i.e. to the partitioned counter.
// While a batch is scanned and counted the readers fill the other batch buffer.  The host parses nothing: the 16-thread parser team
// (kg_ingest.cpp: parse_file_parallel) delivered 22-35 GB/s of FASTQ and one feeder thread copied its output into pinned memory at
// 8.7; here the bound is pread + PCIe.
// A batch the device cannot vouch for (kg_scan.hpp lists what) is parsed by the host state machine from the batch's first byte,
// which the batch before it has proven to be a record start -- and so is the rest of the file: the stream stays the streaming
// parser's whatever the file looks like.
#include "kg_host.hpp"
#include "kg_ingest.hpp"
#include "kg_scan.hpp"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

static const size_t g_scan_batch = (size_t)std::max<uint64_t>(1, getenv("KATGPU_SCAN_BATCH_MB") ? strtoull(getenv("KATGPU_SCAN_BATCH_MB"), nullptr, 10) : 1024) << 20;
static const size_t g_scan_segment = (size_t)std::max<uint64_t>(1, getenv("KATGPU_SCAN_SEGMENT_MB") ? strtoull(getenv("KATGPU_SCAN_SEGMENT_MB"), nullptr, 10) : 32) << 20;
static const size_t g_scan_overlap_dflt = (size_t)1 << 20;        // how far past a batch's nominal end its cut may lie (a record, a FASTA line)
static const unsigned g_scan_threads = (unsigned)std::max<uint64_t>(1, getenv("KATGPU_SCAN_THREADS") ? strtoull(getenv("KATGPU_SCAN_THREADS"), nullptr, 10) : 12);
static const uint64_t g_scan_min_bytes = getenv("KATGPU_SCAN_MIN_BYTES") ? strtoull(getenv("KATGPU_SCAN_MIN_BYTES"), nullptr, 10) : ((uint64_t)64 << 20);
static const bool g_scan_off = getenv("KATGPU_DEVICE_SCAN") && atoi(getenv("KATGPU_DEVICE_SCAN")) == 0;
// tests: small batches / overlaps (bytes) so that little files cross many cuts; force the host fall-back from batch N on
static const size_t g_test_scan_batch = (size_t)hook_u64("KATGPU_TEST_SCAN_BATCH", 0), g_test_scan_overlap = (size_t)hook_u64("KATGPU_TEST_SCAN_OVERLAP", 0);
static const size_t g_test_scan_segment = (size_t)hook_u64("KATGPU_TEST_SCAN_SEGMENT", 0);
static const uint64_t g_test_scan_fail_at = hook_u64("KATGPU_TEST_SCAN_FAIL_AT", ~0ULL);

bool device_scan_applies(const char* path, uint32_t trim5p, uint64_t* size_out, uint8_t* first_byte) {
    if (g_scan_off || trim5p) return false;                       // (a 5' trim swallows line starts the way is.ignore does: the host machine's job)
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if ((uint64_t)st.st_size < (g_test_scan_batch ? 1 : g_scan_min_bytes)) return false;
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
    // two batch buffers: pinned host, raw device, output device
    uint8_t* pin[2] = {nullptr, nullptr};
    uint8_t* raw[2] = {nullptr, nullptr};
    uint8_t* out[2] = {nullptr, nullptr};
    uint8_t* raw_al = nullptr;                                    // a chunk starts at a record, i.e. anywhere: the scan reads it from a 16-byte aligned copy (one D2D copy, ~0.5 ms per GiB)
    // the scan's arrays (one set: scans of successive batches are serial on the compute stream)
    uint32_t *tile_cnt = nullptr, *NL = nullptr, *len_off = nullptr, *line_tile_sum = nullptr;
    uint64_t *tile_off = nullptr, *line_tile_off = nullptr;
    unsigned long long* flags = nullptr;
    uint64_t cap_lines = 0;
    hipStream_t up[2] = {nullptr, nullptr};                       // one upload stream per batch buffer
    // readers
    std::vector<std::thread> readers;
    std::mutex mu; std::condition_variable cv;
    uint64_t next_seg = 0;                                        // segment counter over my batches (my[j] = segments [j * spb, (j + 1) * spb))
    uint64_t n_batches = 0, spb = 0;                              // batches of the file; segments per batch (a batch's first one also reads PRE, its last one the overlap)
    std::vector<uint32_t> done_segs;                              // per my[j]: segments read and enqueued
    uint64_t consumed = 0;                                        // of my batches, how many the main thread is through with: my[j] may be read when j < consumed + 2
    bool stop = false, io_error = false;

    RawFeeder(katgpu_table* t_, const char* p) : t(t_), c(t_->ctx), path(p) {}
    ~RawFeeder() { shutdown(); release(); }

    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& th : readers) if (th.joinable()) th.join();
        readers.clear();
    }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (up[i]) { hipStreamSynchronize(up[i]); hipStreamDestroy(up[i]); up[i] = nullptr; }
            if (pin[i]) { hipHostFree(pin[i]); pin[i] = nullptr; }
            if (raw[i]) { pool_release(c, raw[i]); raw[i] = nullptr; }
            if (out[i]) { pool_release(c, out[i]); out[i] = nullptr; }
        }
        if (raw_al) { pool_release(c, raw_al); raw_al = nullptr; }
        hipFree(tile_cnt); hipFree(NL); hipFree(len_off); hipFree(line_tile_sum); hipFree(tile_off); hipFree(line_tile_off); hipFree(flags);
        tile_cnt = NL = len_off = line_tile_sum = nullptr; tile_off = line_tile_off = nullptr; flags = nullptr;
        if (fd >= 0) { ::close(fd); fd = -1; }
    }

    int setup(uint64_t file_size, uint8_t first, int rank, int world) {
        size = file_size;
        type = first == '@' ? SCAN_FASTQ : SCAN_FASTA;
        shard_rank = rank; shard_world = world;
        batch = g_test_scan_batch ? g_test_scan_batch : g_scan_batch;
        segment = g_test_scan_segment ? g_test_scan_segment : std::min(g_scan_segment, batch);
        overlap = g_test_scan_overlap ? g_test_scan_overlap : g_scan_overlap_dflt;
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
        const int nb = my.size() > 1 ? 2 : 1;
        for (int i = 0; i < nb; ++i) {
            HIPCHK(c, hipHostMalloc((void**)&pin[i], buf_bytes, hipHostMallocDefault));
            HIPCHK(c, pool_alloc(c, (void**)&raw[i], buf_bytes));
            HIPCHK(c, pool_alloc(c, (void**)&out[i], HEAD + buf_bytes));
            HIPCHK(c, hipStreamCreateWithFlags(&up[i], hipStreamNonBlocking));
        }
        HIPCHK(c, pool_alloc(c, (void**)&raw_al, buf_bytes));
        const uint64_t n_tiles = (buf_bytes + SC_TILE - 1) / SC_TILE;
        cap_lines = buf_bytes / 16 + 4096;
        HIPCHK(c, hipMalloc((void**)&tile_cnt, (n_tiles + 1) * 4));
        HIPCHK(c, hipMalloc((void**)&tile_off, (n_tiles + 2) * 8));
        HIPCHK(c, hipMalloc((void**)&NL, cap_lines * 4));
        HIPCHK(c, hipMalloc((void**)&len_off, cap_lines * 4));
        HIPCHK(c, hipMalloc((void**)&line_tile_sum, (cap_lines / SC_BLOCK + 2) * 4));
        HIPCHK(c, hipMalloc((void**)&line_tile_off, (cap_lines / SC_BLOCK + 3) * 8));
        HIPCHK(c, hipMalloc((void**)&flags, SCF_WORDS * 8));
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(g_scan_threads, (size + segment - 1) / segment));
        for (unsigned i = 0; i < T; ++i) readers.emplace_back([this] { read_loop(); });
        return KATGPU_OK;
    }

    // file range of batch b as read: [base, hi_read) = [b * batch - PRE, min(size, (b + 1) * batch + overlap)); buffer byte 0 = file byte base
    uint64_t batch_base(uint64_t b) const { return b ? b * (uint64_t)batch - PRE : 0; }
    uint64_t batch_hi_read(uint64_t b) const { return std::min<uint64_t>(size, (b + 1) * (uint64_t)batch + overlap); }

    void read_loop() {
        hipSetDevice(c->device);
        for (;;) {
            uint64_t seg;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || io_error || next_seg >= my.size() * spb || next_seg / spb < consumed + 2; });
                if (stop || io_error || next_seg >= my.size() * spb) return;
                seg = next_seg++;
            }
            const uint64_t j = seg / spb, s = seg % spb, b = my[j];
            const int buf = (int)(j & 1);
            // segment s of batch b: [b * batch + s * segment, ... + segment); the first one also reads PRE, the last one the overlap
            const uint64_t f0 = s ? b * (uint64_t)batch + s * (uint64_t)segment : batch_base(b);
            const uint64_t f1 = s + 1 == spb ? batch_hi_read(b) : std::min<uint64_t>(size, b * (uint64_t)batch + (s + 1) * (uint64_t)segment);
            bool ok = true;
            if (f0 < f1) {
                uint8_t* dst = pin[buf] + (f0 - batch_base(b));
                uint64_t got = 0;
                while (got < f1 - f0) {
                    const ssize_t r = pread(fd, dst + got, (size_t)std::min<uint64_t>(f1 - f0 - got, (uint64_t)1 << 30), (off_t)(f0 + got));
                    if (r <= 0) { ok = false; break; }
                    got += (uint64_t)r;
                }
                if (ok && hipMemcpyAsync(raw[buf] + (f0 - batch_base(b)), dst, (size_t)(f1 - f0), hipMemcpyHostToDevice, up[buf]) != hipSuccess) ok = false;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok) io_error = true;
                ++done_segs[j];
            }
            cv.notify_all();
        }
    }

    // wait until every segment of batch b is in pinned memory and its copy enqueued, then until the copies have landed
    int wait_batch(uint64_t j) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return io_error || done_segs[j] == spb; });
            if (io_error) return fail(c, KATGPU_ERR_IO, "read error on %s", path);
        }
        HIPCHK(c, hipStreamSynchronize(up[j & 1]));
        return KATGPU_OK;
    }
    void batch_consumed() {
        { std::lock_guard<std::mutex> lk(mu); ++consumed; }
        cv.notify_all();
    }

    // The record scan of raw[buf][lo, hi) into out[buf] + HEAD.  *valid: the device vouches for the chunk; *out_n: bytes of base stream.
    int scan(int buf, uint64_t lo, uint64_t hi, bool* valid, uint64_t* out_n) {
        const uint8_t* src = raw[buf] + lo;
        const uint64_t n = hi - lo;
        *valid = false; *out_n = 0;
        if (n == 0) { *valid = true; return KATGPU_OK; }
        if (reinterpret_cast<uintptr_t>(src) & 15) {
            HIPCHK(c, hipMemcpyAsync(raw_al, src, n, hipMemcpyDeviceToDevice, c->stream));
            src = raw_al;
        }
        const uint32_t n_tiles = (uint32_t)((n + SC_TILE - 1) / SC_TILE);
        const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)c->n_cu * 16);
        ScopedTimer tm(c, KATGPU_K_SCAN, n);
        HIPCHK(c, hipMemsetAsync(flags, 0, SCF_WORDS * 8, c->stream));
        hipLaunchKernelGGL(k_nl_count, dim3(grid), dim3(SC_BLOCK), 0, c->stream, src, n, n_tiles, tile_cnt, flags);
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)tile_cnt, n_tiles, (uint64_t)n_tiles, (const uint64_t*)nullptr, tile_off,
                           (uint64_t)(n_tiles + 1), 1, (unsigned long long*)&flags[SCF_LINES]);
        unsigned long long h[SCF_WORDS];
        HIPCHK(c, hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint64_t n_lines = h[SCF_LINES];
        if (h[SCF_BAD] || n_lines == 0 || n_lines > cap_lines || (type == SCAN_FASTQ && (n_lines & 3))) return KATGPU_OK;     // the host's
        const uint64_t n_ltiles = (n_lines + SC_BLOCK - 1) / SC_BLOCK;
        const int lgrid = (int)std::min<uint64_t>(n_ltiles, (uint64_t)c->n_cu * 16);
        hipLaunchKernelGGL(k_nl_write, dim3(grid), dim3(SC_BLOCK), 0, c->stream, src, n, n_tiles, (const uint64_t*)tile_off, NL, cap_lines);
        if (type == SCAN_FASTQ) hipLaunchKernelGGL(k_line_len<SCAN_FASTQ>, dim3(lgrid), dim3(SC_BLOCK), 0, c->stream, src, (const uint32_t*)NL, n_lines, len_off, line_tile_sum, flags);
        else hipLaunchKernelGGL(k_line_len<SCAN_FASTA>, dim3(lgrid), dim3(SC_BLOCK), 0, c->stream, src, (const uint32_t*)NL, n_lines, len_off, line_tile_sum, flags);
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)line_tile_sum, (uint32_t)n_ltiles, (uint64_t)n_ltiles, (const uint64_t*)nullptr,
                           line_tile_off, (uint64_t)(n_ltiles + 1), 1, (unsigned long long*)&flags[SCF_OUT]);
        HIPCHK(c, hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h[SCF_BAD] || h[SCF_OUT] > buf_bytes) return KATGPU_OK;
        hipLaunchKernelGGL(k_line_off, dim3(lgrid), dim3(SC_BLOCK), 0, c->stream, n_lines, len_off, (const uint64_t*)line_tile_off);
        uint8_t* dst = out[buf] + HEAD;
        if (type == SCAN_FASTQ) hipLaunchKernelGGL(k_emit<SCAN_FASTQ>, dim3(grid), dim3(SC_BLOCK), 0, c->stream, src, n, n_tiles, (const uint64_t*)tile_off, (const uint32_t*)NL,
                                                   (const uint32_t*)len_off, n_lines, dst);
        else hipLaunchKernelGGL(k_emit<SCAN_FASTA>, dim3(grid), dim3(SC_BLOCK), 0, c->stream, src, n, n_tiles, (const uint64_t*)tile_off, (const uint32_t*)NL,
                                (const uint32_t*)len_off, n_lines, dst);
        HIPCHK(c, hipGetLastError());
        *valid = true; *out_n = h[SCF_OUT];
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
                const size_t keep = std::min<size_t>(outb.size(), t->d.k - 1);
                std::vector<uint8_t> tailb(outb.end() - keep, outb.end());
                outb.assign(tailb.begin(), tailb.end());
            }
        }
        if (rc == KATGPU_OK && !ps.end_ok()) return fail(c, KATGPU_ERR_FASTQ, "Invalid fastq sequence");
        return rc;
    }
    // a host piece of the base stream -> out[0] -> count_resident (the fall-back is rare: no pipelining)
    int count_host_stream(const uint8_t* p, size_t n) {
        size_t pos = 0;
        const uint32_t k = t->d.k;
        while (pos < n && n - pos >= k) {
            const size_t take = std::min(n - pos, buf_bytes);
            uint8_t head[HEAD];
            memset(head, 'N', HEAD);
            HIPCHK(c, hipMemcpyAsync(out[0], head, HEAD, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(out[0] + HEAD, p + pos, take, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            int rc = count_resident(t, out[0], HEAD + take);
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

    int run() {
        const uint32_t k = t->d.k;
        const bool sharded = shard_world > 1;
        uint64_t cut_lo = 0;                                      // file offset where the next chunk starts: a proven record / line start
        uint8_t carry[HEAD]; uint32_t carry_n = 0;                // last k-1 bytes of the base stream so far (host copy, for the fall-back)
        int prev_buf = -1; uint64_t prev_out_n = 0;
        for (uint64_t j = 0; j < my.size(); ++j) {
            int rc = wait_batch(j);
            if (rc) return rc;
            const uint64_t b = my[j];
            const int buf = (int)(j & 1);
            const uint64_t lo = batch_base(b), hi_read = batch_hi_read(b);
            bool cut_ok = true;
            if (sharded) {                                        // this chunk's start, found the way the owner of batch b - 1 finds its chunk's end
                cut_lo = 0;
                if (b) { const int64_t f = fastq_cut(pin[buf], lo, hi_read, b * (uint64_t)batch); if (f < 0) cut_ok = false; else cut_lo = (uint64_t)f; }
                prev_buf = -1; prev_out_n = 0;                    // (FASTQ chunks end on a record's 'N': nothing to carry between them)
            }
            // the chunk's end: the first record start (FASTQ) / line start (FASTA) at or after the nominal end, inside what was read
            uint64_t cut_hi = size;
            if (b + 1 < n_batches) {
                const uint64_t nominal = (b + 1) * (uint64_t)batch;
                if (type == SCAN_FASTQ) {
                    const int64_t f = fastq_cut(pin[buf], lo, hi_read, nominal);
                    if (f < 0) cut_ok = false; else cut_hi = (uint64_t)f;
                } else {
                    const uint8_t* nl = (const uint8_t*)memchr(pin[buf] + (nominal - 1 - lo), '\n', (size_t)(hi_read - (nominal - 1)));
                    if (!nl) cut_ok = false; else cut_hi = lo + (uint64_t)(nl - pin[buf]) + 1;
                }
            } else if (size && pin[buf][size - 1 - lo] != '\n') cut_ok = false;              // a last line without its newline: the host machine knows what to do
            bool valid = false; uint64_t out_n = 0;
            if (cut_ok && cut_lo <= cut_hi && b < g_test_scan_fail_at) {
                rc = scan(buf, cut_lo - lo, cut_hi - lo, &valid, &out_n);
                if (rc) return rc;
            }
            if (!valid && sharded)
                return fail(c, KATGPU_ERR_FASTQ, "%s: batch %llu is not plain four-line FASTQ (or holds a '\\r'): such a file cannot be cut between GPUs -- run it on one", path, (unsigned long long)b);
            if (!valid) {
                if (g_trace) fprintf(stderr, "[katgpu] device scan: batch %llu of %s goes to the host parser (and the rest of the file with it)\n", (unsigned long long)b, path);
                if (prev_buf >= 0 && carry_n == 0 && prev_out_n) {                           // what the stream ended on, for the windows across the hand-over
                    carry_n = (uint32_t)std::min<uint64_t>(prev_out_n, k - 1);
                    HIPCHK(c, hipMemcpy(carry, out[prev_buf] + HEAD + prev_out_n - carry_n, carry_n, hipMemcpyDeviceToHost));
                }
                shutdown();
                return host_rest(cut_lo, carry, carry_n);
            }
            // the carry: the previous chunk's last k-1 output bytes, right in front of this chunk's (FASTA chunks cut a record's sequence anywhere between two lines)
            uint8_t* o = out[buf];
            HIPCHK(c, hipMemsetAsync(o, 'N', HEAD, c->stream));
            if (prev_buf >= 0 && prev_out_n) {
                const uint32_t cn = (uint32_t)std::min<uint64_t>(prev_out_n, k - 1);
                HIPCHK(c, hipMemcpyAsync(o + HEAD - cn, out[prev_buf] + HEAD + prev_out_n - cn, cn, hipMemcpyDeviceToDevice, c->stream));
            }
            if (out_n) {
                rc = count_resident(t, o, HEAD + out_n);         // (ends synchronised with the stream)
                if (rc) return rc;
                prev_buf = buf; prev_out_n = out_n;
            } else HIPCHK(c, hipStreamSynchronize(c->stream));    // the scan's kernels are through with raw[buf]
            cut_lo = cut_hi;
            batch_consumed();                                     // raw[buf] / pin[buf] may take my batch after next (out[buf] is rewritten by ITS scan, on this stream)
        }
        return KATGPU_OK;
    }
};

}  // namespace

// One large plain file through the device scan.  *took = false: not a file for this path (nothing was counted).
int count_file_device_scan(katgpu_table* t, const char* path, uint32_t trim5p, bool* took, int rank, int world) {
    *took = false;
    uint64_t size = 0; uint8_t first = 0;
    if (!device_scan_applies(path, trim5p, &size, &first)) return KATGPU_OK;
    katgpu_ctx* c = t->ctx;
    if (c->arena && !c->arena_busy && !c->arena_borrowed) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < ((size_t)6 << 30)) release_arena(c);    // the cached arena holds most of the free HBM: the batch buffers come first
    }
    RawFeeder f(t, path);
    if (world > 1 && first != '@') return KATGPU_OK;              // only FASTQ is cut between ranks (the caller deals whole files otherwise)
    int rc = f.setup(size, first, rank, world);
    if (rc == KATGPU_ERR_NOMEM) { (void)hipGetLastError(); return KATGPU_OK; }      // no room for the batch buffers: the streaming path and its smaller rings
    if (rc == KATGPU_OK) { *took = true; rc = f.run(); }
    if (rc == KATGPU_OK) rc = refresh_counters(t);
    return rc;
}
