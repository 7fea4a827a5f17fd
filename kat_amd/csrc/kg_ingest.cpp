// kg_ingest.cpp -- see kg_ingest.hpp.  Pure host code (no HIP), compiled into libkatgpu.so.
#include "kg_ingest.hpp"

#include "../../include/katgpu.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

namespace kg {

uint64_t file_size_or_zero(const char* path) {
    struct stat st;
    return (stat(path, &st) == 0 && S_ISREG(st.st_mode)) ? (uint64_t)st.st_size : 0;
}

SeqFileParser::SeqFileParser() = default;
SeqFileParser::~SeqFileParser() {
    if (gz_) gzclose((gzFile)gz_);
}

int SeqFileParser::open(const char* path, uint32_t trim5p, std::string* err, size_t raw_bytes) {
    path_ = path;
    ps_.trim5p = trim5p;
    // the reference opens every input through a gzip-aware stream (stream_manager.hpp:133-145); zlib passes plain files through
    gz_ = gzopen(path, "rb");
    if (!gz_) {
        *err = "Could not find input file at: " + path_ + "; please check the path and try again.";   // input_handler.cc:119-122
        return KATGPU_ERR_IO;
    }
    gzbuffer((gzFile)gz_, 1 << 20);
    raw_.resize(std::max<size_t>(raw_bytes, 1));
    out_.reserve(raw_.size() + 16);
    return KATGPU_OK;
}

void ParseState::after_header() {
    if (trim5p > 0) st = TRIM_SKIPNL;            // read_sequence: skip_newlines(); is.ignore(trim5p)   (parser.hpp:250-253)
    else st = LOOP_CHECK;
}

bool ParseState::begin(uint8_t first_byte) {       // open_next_file: dispatch on the first byte (parser.hpp:171-185)
    if (first_byte == '>') type = FASTA;
    else if (first_byte == '@') type = FASTQ;
    else return false;
    st = HEADER; seq_len = 0;
    return true;
}

// Truncated record?  (Deviation, documented in DESIGN.md: a last quality line of exactly the right length but without '\n'
// is accepted; the reference throws there and its pool swallows the exception.)
bool ParseState::end_ok() const {
    if (type != FASTQ) return true;
    if (st == QUAL_IGNORE) return quals + got == read_len;
    if (st == QUAL_SKIPNL) return false;
    if (st == PLUS_LINE) return seq_len + trim5p == 0;
    return true;
}

void ParseState::consume(const uint8_t* d, size_t n, std::vector<uint8_t>& out, bool* bad_fastq) {
    const uint8_t stop = type == FASTA ? '>' : '+';
    size_t i = 0;
    while (i < n) {
        switch (st) {
        case HEADER:
        case PLUS_LINE: {                                   // ignore_line (parser.hpp:264-266)
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', n - i);
            if (!nl) { i = n; break; }
            i = (size_t)(nl - d) + 1;
            if (st == HEADER) after_header();
            else {                                           // skip_quals(seq_len + trim5p)   (parser.hpp:237,274-289)
                read_len = seq_len + trim5p; quals = 0;
                st = read_len ? QUAL_SKIPNL : QUAL_DONE_SKIPNL;
            }
            break;
        }
        case TRIM_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            trim_left = trim5p; st = TRIM_IGNORE;
            break;
        case TRIM_IGNORE: {
            size_t take = (size_t)std::min<uint64_t>(trim_left, n - i);
            i += take; trim_left -= take;
            if (!trim_left) st = LOOP_CHECK;
            break;
        }
        case LOOP_CHECK:                                     // "while(... && is.peek() != stop)"   (parser.hpp:254)
            if (d[i] == stop) {
                if (type == FASTA) { out.push_back('N'); st = HEADER; }        // 'N' between records (parser.hpp:202)
                else st = PLUS_LINE;
            } else if (d[i] == '\n') st = FORCED_SKIPNL;     // blank line right after a header: the next line is read unconditionally
            else st = SEQ_LINE;
            break;
        case FORCED_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            st = SEQ_LINE;
            break;
        case SEQ_LINE: {                                     // is.get(): the rest of the line, verbatim   (parser.hpp:257)
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', n - i);
            size_t e = nl ? (size_t)(nl - d) : n;
            out.insert(out.end(), d + i, d + e);
            seq_len += e - i;
            i = e;
            if (nl) st = SEQ_SKIPNL;
            break;
        }
        case SEQ_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            st = LOOP_CHECK;
            break;
        case QUAL_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            want = read_len - quals + 1; got = 0; st = QUAL_IGNORE;
            break;
        case QUAL_IGNORE: {                                  // is.ignore(read_len - quals + 1, '\n')   (parser.hpp:280)
            size_t lim = (size_t)std::min<uint64_t>(want - got, n - i);
            const uint8_t* nl = (const uint8_t*)memchr(d + i, '\n', lim);
            size_t take = nl ? (size_t)(nl - (d + i)) + 1 : lim;
            i += take; got += take;
            if (nl || got == want) {
                quals += got; ++read_len;                    // "if(is) ++read_len"
                st = quals < read_len ? QUAL_SKIPNL : QUAL_DONE_SKIPNL;
            }
            break;
        }
        case QUAL_DONE_SKIPNL:
            if (d[i] == '\n') { ++i; break; }
            if (d[i] == '@') { out.push_back('N'); seq_len = 0; st = HEADER; }      // parser.hpp:285-286,238-242
            else { *bad_fastq = true; return; }
            break;
        }
    }
}

int SeqFileParser::next(const uint8_t** p, size_t* n, std::string* err) {
    *p = nullptr; *n = 0;
    out_.clear();
    while (!eof_ && out_.empty()) {
        int r = gzread((gzFile)gz_, raw_.data(), (unsigned)raw_.size());
        if (r < 0) { *err = "read error on " + path_; return KATGPU_ERR_IO; }
        if (r == 0) {
            eof_ = true;
            if (!ps_.end_ok()) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
            break;
        }
        if (ps_.type == ParseState::NONE && !ps_.begin(raw_[0])) { *err = "Unsupported format"; return KATGPU_ERR_FORMAT; }
        bool bad = false;
        ps_.consume(raw_.data(), (size_t)r, out_, &bad);
        if (bad) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
    }
    *p = out_.data(); *n = out_.size();
    return KATGPU_OK;
}

// ------------------------------------------------------------------ parallel front end ------------------------------

namespace {

uint64_t env_u64(const char* name, uint64_t def) {
    const char* v = getenv(name);
    return v && *v ? strtoull(v, nullptr, 10) : def;
}

// First guessed record start at file offset >= from, looking only at buf (which holds file bytes [buf_off, buf_off + len)).
// Returns the file offset, or -1 when none is certain inside the buffer.
int64_t find_cut(ParseState::Type type, const uint8_t* buf, int64_t buf_off, int64_t len, int64_t from) {
    int64_t p = from - buf_off;
    if (p < 1) p = 1;                                        // a cut needs the '\n' before it
    const uint8_t mark = type == ParseState::FASTA ? '>' : '@';
    while (p < len) {
        const uint8_t* nl = (const uint8_t*)memchr(buf + p - 1, '\n', (size_t)(len - (p - 1)));
        if (!nl) return -1;
        p = (nl - buf) + 1;                                  // first byte of the next line
        if (p >= len) return -1;
        if (buf[p] != mark) { ++p; continue; }
        if (type == ParseState::FASTA) return buf_off + p;
        const uint8_t* l1 = (const uint8_t*)memchr(buf + p, '\n', (size_t)(len - p));                 // end of the header line
        if (!l1) return -1;
        const uint8_t* l2 = (const uint8_t*)memchr(l1 + 1, '\n', (size_t)(len - (l1 + 1 - buf)));      // end of the sequence line
        if (!l2 || l2 + 1 >= buf + len) return -1;
        if (l2[1] == '+') return buf_off + p;
        ++p;
    }
    return -1;
}

struct Segment {
    int64_t s = -1, e = -1;          // accepted range [s, e) in file offsets; -1: no certain cut
    std::vector<uint8_t> out;
    ParseState end;
    bool bad = false, io_error = false;
};

}  // namespace

int64_t find_record_start(ParseState::Type type, const uint8_t* buf, int64_t buf_off, int64_t len, int64_t from) { return find_cut(type, buf, buf_off, len, from); }

bool strip_fastq_records(const uint8_t* p, size_t n, uint8_t* out, size_t* out_n) {
    const uint8_t* const end = p + n;
    uint8_t* o = out;
    while (p < end) {
        if (*p != '@') return false;
        const uint8_t* h = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));                      // end of the header line
        if (!h) return false;
        const uint8_t* s = h + 1;
        const uint8_t* se = s < end ? (const uint8_t*)memchr(s, '\n', (size_t)(end - s)) : nullptr;  // end of the sequence line
        if (!se) return false;
        const size_t len = (size_t)(se - s);
        if (len == 0 || *s == '+') return false;                                                     // (a record without bases, or a "sequence" line that the reference's loop takes for the '+' line of one -- it stops reading bases at the first line that begins with '+', mer_overlap_sequence_parser.hpp:226-233 --: the state machine's to say, it words the reference's error)
        const uint8_t* pl = se + 1;
        if (pl >= end || *pl != '+') return false;                                                   // (a second sequence line: not plain)
        const uint8_t* ple = (const uint8_t*)memchr(pl, '\n', (size_t)(end - pl));                   // end of the '+' line
        if (!ple) return false;
        const uint8_t* q = ple + 1;
        if ((size_t)(end - q) < len + 1 || q[len] != '\n') return false;                             // the quality line: exactly the sequence's length ...
        if (memchr(q, '\n', len)) return false;                                                    // ... in ONE line
        memcpy(o, s, len);
        o += len;
        *o++ = 'N';
        p = q + len + 1;
    }
    *out_n = (size_t)(o - out);
    return true;
}

// The conditions under which the team takes a file; *size_out / *first_byte for the caller that goes on.
static bool team_applies_impl(const char* path, uint32_t trim5p, int64_t* size_out, uint8_t* first_byte) {
    if (trim5p) return false;                                // is.ignore(trim5p) swallows line starts: keep that case on the streaming path
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if ((uint64_t)st.st_size < env_u64("KATGPU_INGEST_MIN_BYTES", (uint64_t)256 << 20)) return false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t head[2] = {0, 0};
    const bool got = pread(fd, head, 2, 0) == 2;
    ::close(fd);
    if (!got) return false;
    if (head[0] == 0x1f && head[1] == 0x8b) return false;    // gzip: one serial stream
    if (head[0] != '>' && head[0] != '@') return false;      // the streaming path words the error
    if (size_out) *size_out = (int64_t)st.st_size;
    if (first_byte) *first_byte = head[0];
    return true;
}

bool team_applies(const char* path, uint32_t trim5p) { return team_applies_impl(path, trim5p, nullptr, nullptr); }

int parse_file_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    int64_t size = 0;
    uint8_t first = 0;
    if (!team_applies_impl(path, trim5p, &size, &first)) return -1;
    const int64_t seg = (int64_t)std::max<uint64_t>(16, env_u64("KATGPU_INGEST_SEGMENT", (uint64_t)16 << 20));
    const int64_t margin = (int64_t)std::max<uint64_t>(16, env_u64("KATGPU_INGEST_MARGIN", (uint64_t)4 << 20));
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    // 16: measured on the MI355X host (2 x EPYC 9575F, tools/bench_ingest_host.sh): 35 GB/s of FASTQ into a null sink with 16
    // threads, 23 with 32, 11 with 64 -- every wave's fresh buffers are first touched by all threads at once, and page faults of
    // one process serialise on its address-space lock.  The feeder behind it takes ~22 GB/s.
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(env_u64("KATGPU_INGEST_THREADS", std::min(hw, 16u)), 256));
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    ParseState state;                                        // the machine's TRUE state at file offset `pos`
    state.begin(first);
    const ParseState::Type type = state.type;
    int64_t pos = 0;

    // one wave = T consecutive nominal segments starting at `from`; every thread reads its own bytes (parallel I/O too).
    // Raw and output buffers live in two sets that successive waves alternate between (a wave is parsed while the previous
    // one is handed over): fresh 64 MB vectors per piece would spend more time in page faults than in parsing.
    std::vector<std::vector<uint8_t>> raws(2 * (size_t)T);
    std::vector<std::shared_ptr<std::vector<Segment>>> sets = {std::make_shared<std::vector<Segment>>(T), std::make_shared<std::vector<Segment>>(T)};
    unsigned parity = 0;
    auto parse_wave = [&](int64_t from, const ParseState& first_state, unsigned set) {
        auto segs = sets[set];
        std::vector<std::thread> team;
        for (unsigned i = 0; i < T; ++i) {
            team.emplace_back([&, i, from, first_state]() {
                Segment& g = (*segs)[i];
                g.s = g.e = -1; g.bad = g.io_error = false; g.out.clear();
                const int64_t ns = from + (int64_t)i * seg, ne = std::min(size, ns + seg);
                if (ns >= size) { g.s = g.e = size; g.end = first_state; return; }
                const int64_t b0 = i == 0 ? ns : ns - 1, b1 = std::min(size, ne + margin);
                std::vector<uint8_t>& raw = raws[(size_t)set * T + i];
                if (raw.size() < (size_t)(b1 - b0)) raw.resize((size_t)(b1 - b0));
                int64_t got = 0;
                while (got < b1 - b0) {
                    ssize_t r = pread(fd, raw.data() + got, (size_t)(b1 - b0 - got), b0 + got);
                    if (r <= 0) { g.io_error = true; return; }
                    got += r;
                }
                g.s = i == 0 ? ns : find_cut(type, raw.data(), b0, b1 - b0, ns);
                if (g.s < 0 && b1 == size) g.s = size;       // no record starts in the rest of the file
                g.e = ne >= size ? size : find_cut(type, raw.data(), b0, b1 - b0, ne);
                if (g.e < 0 && b1 == size) g.e = size;
                if (g.s < 0 || g.e < 0 || g.e < g.s) { g.s = g.e = -1; return; }
                ParseState ps = first_state;
                if (i > 0) {                                 // assumed: a record starts here, i.e. the piece before ended at a boundary
                    ps = ParseState(); ps.type = type;
                    ps.st = type == ParseState::FASTA ? ParseState::LOOP_CHECK : ParseState::QUAL_DONE_SKIPNL;
                }
                g.out.reserve((size_t)(g.e - g.s));
                ps.consume(raw.data() + (g.s - b0), (size_t)(g.e - g.s), g.out, &g.bad);
                g.end = ps;
            });
        }
        for (auto& th : team) th.join();
        return segs;
    };

    auto serial_rest = [&]() -> int {                        // from `pos`, in `state`, to the end of the file
        std::vector<uint8_t> raw((size_t)16 << 20), out;
        while (pos < size) {
            ssize_t r = pread(fd, raw.data(), (size_t)std::min<int64_t>((int64_t)raw.size(), size - pos), pos);
            if (r <= 0) { *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
            out.clear();
            bool bad = false;
            state.consume(raw.data(), (size_t)r, out, &bad);
            if (bad) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
            pos += r;
            if (!out.empty()) { int rc = sink(out.data(), out.size()); if (rc) return rc; }
        }
        return KATGPU_OK;
    };

    const bool trace = getenv("KATGPU_TRACE") != nullptr;
    size_t pieces = 0;
    auto wave = parse_wave(0, state, parity);
    while (pos < size) {
        // accept the longest prefix of the wave whose assumptions hold
        size_t ok = 0;
        ParseState st_after = state;
        int64_t p_after = pos;
        for (; ok < wave->size(); ++ok) {
            const Segment& g = (*wave)[ok];
            if (g.io_error) { *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
            if (g.s != p_after || g.e < 0) break;
            if (ok > 0 && g.s < size && !st_after.at_record_boundary()) break;      // the guess at g.s was wrong
            if (g.bad) {
                if (ok == 0 || st_after.at_record_boundary()) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
                break;
            }
            st_after = g.s < size ? g.end : st_after;
            p_after = g.e;
        }
        // the next wave is parsed while this one is handed over
        std::future<std::shared_ptr<std::vector<Segment>>> next;
        const bool whole = ok == wave->size();
        if (whole && p_after < size) { parity ^= 1; next = std::async(std::launch::async, parse_wave, p_after, st_after, parity); }
        for (size_t i = 0; i < ok; ++i) {
            const Segment& g = (*wave)[i];
            if (!g.out.empty()) { int rc = sink(g.out.data(), g.out.size()); if (rc) { if (next.valid()) next.wait(); return rc; } }
        }
        pos = p_after; state = st_after;
        pieces += ok;
        if (!whole) {
            if (trace) fprintf(stderr, "[katgpu] ingest %s: %zu pieces by the team, streaming from offset %lld of %lld\n", path, pieces, (long long)pos, (long long)size);
            int rc = serial_rest(); if (rc) return rc;
            pieces = 0;
            break;
        }
        if (pos < size) wave = next.get();
    }
    if (trace && pieces) fprintf(stderr, "[katgpu] ingest %s: %zu pieces by the team (%u threads), all accepted\n", path, pieces, T);
    if (!state.end_ok()) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
    return KATGPU_OK;
}

// ------------------------------------------------------------------ BGZF ---------------------------------------------

namespace {

// Whole size of the BGZF member whose header starts at p (RFC 1952 header with FLG == FEXTRA and a 'B','C' subfield of two
// bytes holding size - 1; SAM specification section 4.1); 0 when the bytes are not such a header, 1 when `avail` bytes are too
// few to tell.  *hdr: bytes before the deflate data.
uint32_t bgzf_member_size(const uint8_t* p, size_t avail, uint32_t* hdr) {
    static const uint8_t lead[4] = {0x1f, 0x8b, 8, 4};
    for (size_t i = 0; i < 4 && i < avail; ++i) if (p[i] != lead[i]) return 0;
    if (avail < 12) return 1;
    const uint32_t xlen = p[10] | ((uint32_t)p[11] << 8);
    if (avail < 12 + (size_t)xlen) return 1;
    for (uint32_t o = 0; o + 4 <= xlen;) {
        const uint8_t* f = p + 12 + o;
        const uint32_t slen = f[2] | ((uint32_t)f[3] << 8);
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && o + 6 <= xlen) {
            const uint32_t size = (f[4] | ((uint32_t)f[5] << 8)) + 1;
            *hdr = 12 + xlen;
            return size >= *hdr + 8 ? size : 0;
        }
        o += 4 + slen;
    }
    return 0;
}

struct BgzfMember { uint32_t in_off, size, hdr, isize; uint64_t out_off; };
struct BgzfWindow {
    std::vector<uint8_t> comp, raw;
    std::vector<BgzfMember> members;
    int64_t next = 0;              // file offset after the last member taken
    bool foreign = false;          // the bytes at `next` are not a BGZF member (and `next` < file size)
    bool io_error = false, bad_data = false;
};

}  // namespace

bool bgzf_applies(const char* path) {
    if (env_u64("KATGPU_BGZF", 1) == 0) return false;
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if ((uint64_t)st.st_size < env_u64("KATGPU_BGZF_MIN_BYTES", (uint64_t)1 << 20)) return false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t head[64];
    const ssize_t got = pread(fd, head, sizeof head, 0);
    ::close(fd);
    uint32_t hdr;
    return got >= 18 && bgzf_member_size(head, (size_t)got, &hdr) >= 2;
}

int parse_bgzf_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    if (!bgzf_applies(path)) return -1;
    struct stat st;
    if (stat(path, &st) != 0) return -1;
    const int64_t size = (int64_t)st.st_size;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(env_u64("KATGPU_BGZF_THREADS", std::min(hw, 16u)), 256));
    const int64_t window = (int64_t)std::max<uint64_t>((uint64_t)1 << 17, env_u64("KATGPU_BGZF_WINDOW", (uint64_t)16 << 20));   // compressed bytes per round

    // one round: read up to `window` compressed bytes at `from`, find the members, inflate them in parallel
    BgzfWindow sets[2];
    auto inflate_window = [&](int64_t from, BgzfWindow* w) {
        w->members.clear(); w->foreign = w->io_error = w->bad_data = false; w->next = from;
        const int64_t want = std::min(window, size - from);
        if (want <= 0) return;
        if (w->comp.size() < (size_t)want) w->comp.resize((size_t)want);
        int64_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, w->comp.data() + got, (size_t)(want - got), from + got);
            if (r <= 0) { w->io_error = true; return; }
            got += r;
        }
        uint64_t out = 0;
        int64_t p = 0;
        while (p < want) {
            uint32_t hdr = 0;
            const uint32_t msize = bgzf_member_size(w->comp.data() + p, (size_t)(want - p), &hdr);
            if (msize == 1 && from + want < size && !w->members.empty()) break;   // a header cut by the window's end: next round
            if (msize < 2) { w->foreign = true; break; }                         // not a BGZF member (or a truncated header at the end)
            if (p + msize > want) {                                        // the member's tail lies beyond the window
                if (from + p + msize > size) w->bad_data = true;           // ... or beyond the file: truncated
                break;
            }
            const uint8_t* tail = w->comp.data() + p + msize - 8;
            const uint32_t isize = tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
            if (isize > 65536) { w->bad_data = true; break; }
            w->members.push_back({(uint32_t)p, msize, hdr, isize, out});
            out += isize;
            p += msize;
        }
        w->next = from + p;
        if (w->bad_data) return;
        if (w->raw.size() < out + 1) w->raw.resize(out + 1);
        std::atomic<size_t> cursor{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
            for (;;) {
                const size_t i = cursor.fetch_add(1);
                if (i >= w->members.size() || bad) break;
                const BgzfMember& m = w->members[i];
                if (m.isize == 0) continue;                                // the end-of-file marker (and any other empty member)
                inflateReset(&zs);
                zs.next_in = w->comp.data() + m.in_off + m.hdr;
                zs.avail_in = m.size - m.hdr - 8;
                zs.next_out = w->raw.data() + m.out_off;
                zs.avail_out = m.isize;
                const int zr = inflate(&zs, Z_FINISH);
                const uint8_t* tail = w->comp.data() + m.in_off + m.size - 8;
                const uint32_t crc = tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
                if (zr != Z_STREAM_END || zs.total_out != m.isize ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), w->raw.data() + m.out_off, m.isize) != crc) bad = true;
            }
            inflateEnd(&zs);
        };
        const unsigned nt = (unsigned)std::min<size_t>(T, std::max<size_t>(1, w->members.size() / 4));
        std::vector<std::thread> team;
        for (unsigned i = 1; i < nt; ++i) team.emplace_back(work);
        work();
        for (auto& th : team) th.join();
        if (bad) w->bad_data = true;
    };

    ParseState ps;
    ps.trim5p = trim5p;
    std::vector<uint8_t> out;
    auto feed = [&](const uint8_t* d, size_t n) -> int {       // inflated bytes -> state machine -> sink
        if (!n) return KATGPU_OK;
        if (ps.type == ParseState::NONE && !ps.begin(d[0])) { *err = "Unsupported format"; return KATGPU_ERR_FORMAT; }
        out.clear();
        bool bad = false;
        ps.consume(d, n, out, &bad);
        if (bad) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
        if (!out.empty()) { const int rc = sink(out.data(), out.size()); if (rc) { err->clear(); return rc; } }
        return KATGPU_OK;
    };

    int64_t pos = 0;
    unsigned parity = 0;
    size_t n_members = 0;
    inflate_window(0, &sets[0]);
    for (;;) {
        BgzfWindow* w = &sets[parity];
        if (w->io_error || w->bad_data) { *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
        std::future<void> next;
        const bool more = !w->foreign && w->next < size && !w->members.empty();
        if (more) { parity ^= 1; next = std::async(std::launch::async, inflate_window, w->next, &sets[parity]); }   // inflate ahead
        const uint64_t raw_bytes = w->members.empty() ? 0 : w->members.back().out_off + w->members.back().isize;
        const int rc = feed(w->raw.data(), (size_t)raw_bytes);
        n_members += w->members.size();
        pos = w->next;
        if (rc) { if (next.valid()) next.wait(); return rc; }
        if (!more) {
            if (!w->foreign && w->next < size && w->members.empty()) { *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }   // no progress
            break;
        }
        next.get();
    }
    if (getenv("KATGPU_TRACE")) fprintf(stderr, "[katgpu] ingest %s: %zu BGZF members by the team (%u threads)%s\n", path, n_members, T,
                                        pos < size ? ", the rest through zlib" : "");
    if (pos < size) {
        // not a BGZF member here.  gzip magic: the rest goes through zlib from this offset (members are self-contained);
        // anything else is trailing garbage, which zlib ignores after a complete member
        uint8_t magic[2] = {0, 0};
        if (pread(fd, magic, 2, pos) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            const int fd2 = ::open(path, O_RDONLY);
            if (fd2 < 0 || lseek(fd2, (off_t)pos, SEEK_SET) < 0) { if (fd2 >= 0) ::close(fd2); *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
            gzFile gz = gzdopen(fd2, "rb");
            if (!gz) { ::close(fd2); *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
            gzbuffer(gz, 1 << 20);
            std::vector<uint8_t> raw((size_t)4 << 20);
            int rc = KATGPU_OK;
            for (;;) {
                const int r = gzread(gz, raw.data(), (unsigned)raw.size());
                if (r < 0) { *err = std::string("read error on ") + path; rc = KATGPU_ERR_IO; break; }
                if (r == 0) break;
                rc = feed(raw.data(), (size_t)r);
                if (rc) break;
            }
            gzclose(gz);
            if (rc) return rc;
        }
    }
    if (!ps.end_ok()) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
    return KATGPU_OK;
}

// ------------------------------------------------------------------ the input group ---------------------------------

namespace {

// Files [lo, hi) of the group, all of the streaming kind, read by `readers` threads at once (see stream_group in the header).
int stream_run_concurrent(const char* const* paths, const uint16_t* trim5p, size_t lo, size_t hi, uint32_t k, unsigned readers,
                          const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    struct Block { size_t file = 0; unsigned reader = 0; std::vector<uint8_t> data; };
    const size_t raw_bytes = (size_t)std::max<uint64_t>(1, env_u64("KATGPU_INGEST_BLOCK", (uint64_t)4 << 20));
    const size_t depth = 3;                                  // blocks one reader may have waiting

    std::mutex mu;
    std::condition_variable cv_data, cv_room;
    std::deque<Block> ready;                                 // every reader's blocks, in arrival order
    std::vector<size_t> waiting(readers, 0);                 // per reader: how many of them are its own
    std::vector<std::vector<uint8_t>> spare;                 // emptied buffers go back to the readers
    unsigned live = readers;
    size_t bad_file = SIZE_MAX; int bad_rc = 0; std::string bad_msg;      // the lowest-numbered failing file so far
    bool abort_all = false;                                  // the sink failed: nothing more is wanted
    size_t next_file = lo;

    auto reader = [&](unsigned me) {
        for (;;) {
            size_t f;
            {
                std::lock_guard<std::mutex> g(mu);
                // files are handed out in order, so when file f fails every lower-numbered file is already open or done
                if (abort_all || next_file >= hi || next_file > bad_file) break;
                f = next_file++;
            }
            SeqFileParser parser;
            std::string msg;
            int rc = parser.open(paths[f], trim5p ? trim5p[f] : 0, &msg, raw_bytes);
            bool stopped = false;
            while (!rc && !stopped) {
                const uint8_t* p; size_t n;
                rc = parser.next(&p, &n, &msg);
                if (rc || !n) break;
                Block b; b.file = f; b.reader = me;
                std::unique_lock<std::mutex> g(mu);
                cv_room.wait(g, [&] { return waiting[me] < depth || abort_all || f > bad_file; });
                if (abort_all || f > bad_file) { stopped = true; break; }
                if (!spare.empty()) { b.data = std::move(spare.back()); spare.pop_back(); }
                g.unlock();
                b.data.assign(p, p + n);
                g.lock();
                ready.push_back(std::move(b)); ++waiting[me];
                cv_data.notify_one();
            }
            if (rc) {
                std::lock_guard<std::mutex> g(mu);
                if (f < bad_file) { bad_file = f; bad_rc = rc; bad_msg = msg; }
                cv_room.notify_all();                        // higher-numbered files stop; lower ones run on (they may fail too)
            }
        }
        std::lock_guard<std::mutex> g(mu);
        --live;
        cv_data.notify_one();
    };
    std::vector<std::thread> team;
    for (unsigned i = 0; i < readers; ++i) team.emplace_back(reader, i);

    // the consumer: whatever arrives first goes out; 'N' + the file's last k-1 bytes wherever the source changes
    std::vector<std::vector<uint8_t>> tails(hi - lo);
    const size_t keep = k > 0 ? k - 1 : 0;
    const uint8_t sep = 'N';
    size_t last_file = SIZE_MAX;
    int sink_rc = 0;
    for (;;) {
        Block b;
        bool drop;
        {
            std::unique_lock<std::mutex> g(mu);
            cv_data.wait(g, [&] { return !ready.empty() || live == 0; });
            if (ready.empty()) break;
            b = std::move(ready.front()); ready.pop_front();
            drop = abort_all || bad_file != SIZE_MAX;        // after an error the result is discarded anyway
        }
        if (!drop) {
            std::vector<uint8_t>& tail = tails[b.file - lo];
            if (b.file != last_file) {
                if (last_file != SIZE_MAX) sink_rc = sink(&sep, 1);
                if (!sink_rc && !tail.empty()) sink_rc = sink(tail.data(), tail.size());
                last_file = b.file;
            }
            if (!sink_rc) sink_rc = sink(b.data.data(), b.data.size());
            if (b.data.size() >= keep) tail.assign(b.data.end() - keep, b.data.end());
            else {
                tail.insert(tail.end(), b.data.begin(), b.data.end());
                if (tail.size() > keep) tail.erase(tail.begin(), tail.end() - keep);
            }
        }
        std::lock_guard<std::mutex> g(mu);
        if (sink_rc) abort_all = true;
        --waiting[b.reader];
        spare.push_back(std::move(b.data));
        cv_room.notify_all();
    }
    for (auto& th : team) th.join();
    if (sink_rc) { err->clear(); return sink_rc; }
    if (bad_file != SIZE_MAX) { *err = bad_msg; return bad_rc; }
    return last_file != SIZE_MAX ? sink(&sep, 1) : KATGPU_OK;            // the run ends as a file does
}

int stream_one(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    SeqFileParser parser;
    int rc = parser.open(path, trim5p, err);
    while (!rc) {
        const uint8_t* p; size_t n;
        rc = parser.next(&p, &n, err);
        if (rc || !n) break;
        rc = sink(p, n);
        if (rc) { err->clear(); return rc; }
    }
    return rc;
}

}  // namespace

int stream_group(const char* const* paths, size_t n_paths, const uint16_t* trim5p, uint32_t k,
                 const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    static const uint8_t sep = 'N';
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned max_readers = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(env_u64("KATGPU_INGEST_FILES", std::min(hw, 8u)), 64));
    size_t i = 0;
    while (i < n_paths) {
        const uint32_t trim = trim5p ? trim5p[i] : 0;
        if (team_applies(paths[i], trim) || bgzf_applies(paths[i]) || pgz_applies(paths[i], trim)) {
            // a large plain file, a BGZF file or one gzip stream of size: its thread team (same bytes out as the streaming parser)
            int rc = parse_file_parallel(paths[i], trim, sink, err);
            if (rc < 0) rc = parse_bgzf_parallel(paths[i], trim, sink, err);
            if (rc < 0) rc = parse_gz_parallel(paths[i], trim, sink, err);
            if (rc > 0) return rc;
            if (rc < 0) { rc = stream_one(paths[i], trim, sink, err); if (rc) return rc; }    // it changed under us: stream it
            rc = sink(&sep, 1);
            if (rc) { err->clear(); return rc; }
            ++i;
            continue;
        }
        size_t j = i + 1;                                    // the run of streaming files that starts here
        while (j < n_paths && !team_applies(paths[j], trim5p ? trim5p[j] : 0) && !bgzf_applies(paths[j]) && !pgz_applies(paths[j], trim5p ? trim5p[j] : 0)) ++j;
        const unsigned readers = (unsigned)std::min<size_t>(max_readers, j - i);
        int rc;
        if (readers <= 1) {
            rc = KATGPU_OK;
            for (size_t f = i; f < j && !rc; ++f) {
                rc = stream_one(paths[f], trim5p ? trim5p[f] : 0, sink, err);
                if (!rc) { rc = sink(&sep, 1); if (rc) err->clear(); }
            }
        } else rc = stream_run_concurrent(paths, trim5p, i, j, k, readers, sink, err);
        if (rc) return rc;
        i = j;
    }
    return KATGPU_OK;
}

}  // namespace kg

extern "C" int katgpu_strip_fastq(const uint8_t* fastq, size_t n, uint8_t* out, size_t* out_n) {
    if ((n && !fastq) || !out || !out_n) return KATGPU_ERR_INVALID_ARG;
    *out_n = 0;
    return kg::strip_fastq_records(fastq, n, out, out_n) ? KATGPU_OK : KATGPU_ERR_FASTQ;
}

extern "C" int katgpu_parse_file(const char* path, uint32_t trim5p, uint8_t** bases, size_t* n, const char** err_msg) {
    static thread_local std::string last;
    if (!path || !bases || !n) return KATGPU_ERR_INVALID_ARG;
    *bases = nullptr; *n = 0;
    std::vector<uint8_t> all;
    // large plain files: the thread team (same bytes out); everything else, and whatever the team declines: the streaming parser
    auto collect = [&](const uint8_t* p, size_t got) { all.insert(all.end(), p, p + got); return 0; };
    int rc = kg::parse_file_parallel(path, trim5p, collect, &last);
    if (rc < 0) rc = kg::parse_bgzf_parallel(path, trim5p, collect, &last);
    if (rc < 0) rc = kg::parse_gz_parallel(path, trim5p, collect, &last);
    if (rc < 0) {
        all.clear();
        kg::SeqFileParser parser;
        rc = parser.open(path, trim5p, &last);
        while (!rc) {
            const uint8_t* p; size_t got;
            rc = parser.next(&p, &got, &last);
            if (rc || !got) break;
            all.insert(all.end(), p, p + got);
        }
    }
    if (rc) { if (err_msg) *err_msg = last.c_str(); return rc; }
    *bases = (uint8_t*)malloc(all.size() ? all.size() : 1);
    if (!*bases) return KATGPU_ERR_NOMEM;
    memcpy(*bases, all.data(), all.size());
    *n = all.size();
    return KATGPU_OK;
}

extern "C" int katgpu_parse_files(const char* const* paths, size_t n_paths, const uint16_t* trim5p, uint32_t k,
                                  uint8_t** bases, size_t* n, const char** err_msg) {
    static thread_local std::string last;
    if (!paths || !bases || !n) return KATGPU_ERR_INVALID_ARG;
    *bases = nullptr; *n = 0;
    std::vector<uint8_t> all;
    last.clear();
    int rc = kg::stream_group(paths, n_paths, trim5p, k, [&](const uint8_t* p, size_t got) { all.insert(all.end(), p, p + got); return 0; }, &last);
    if (rc) { if (err_msg) *err_msg = last.c_str(); return rc; }
    *bases = (uint8_t*)malloc(all.size() ? all.size() : 1);
    if (!*bases) return KATGPU_ERR_NOMEM;
    memcpy(*bases, all.data(), all.size());
    *n = all.size();
    return KATGPU_OK;
}

extern "C" void katgpu_free_host(void* p) { free(p); }
