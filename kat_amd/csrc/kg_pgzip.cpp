// kg_pgzip.cpp -- ONE ordinary gzip stream inflated by a thread team (see kg_ingest.hpp: parse_gz_parallel).  Pure host code.
//
// The reference reads every input through zlib (deps/jellyfish-2.2.0/include/jellyfish/stream_manager.hpp:41-51,133-145, gzstream.hpp:121):
// one inflate per file, 0.15-0.3 GB/s of FASTQ -- with the counter on a GPU that is the whole run.  A deflate stream has no index, but it
// can be entered anywhere a block starts, at the price of not knowing the 32 KiB of history the block may copy from:
//   * the compressed file is cut into chunks of CB bytes; the worker that takes chunk j > 0 looks, bit by bit from the chunk's first byte,
//     for the start of a dynamic-Huffman block (a header whose three codes are complete prefix codes, a whole block of text behind it,
//     a plausible header after that): its `sync`;
//   * it decodes from there into 16-bit symbols: a byte, or 256 + i for "byte i of the 32 KiB before my first byte" (copies of
//     unknown bytes copy the markers), until a block ends exactly on the sync of a later chunk -- which makes both syncs true block
//     starts, whatever the search believed: a chunk whose sync is passed over without being hit is dropped, the decoder before it just
//     goes on through it.  The bytes that come out are therefore those a sequential inflate produces, for any input;
//   * the consumer walks the live chunks in order: it resolves each chunk's last 32 KiB with the window it has (the next chunk's
//     history), hands the chunk to a second job -- markers -> bytes, CRC-32, the FASTA / FASTQ state machine from a guessed record
//     start to the last one, as parse_file_parallel does with file pieces -- and passes the pieces between the chunks' cuts through
//     the state machine itself, which is also what checks every guess.
// Members: a gzip file may hold several; each member's CRC-32 and ISIZE are checked as zlib does (an error is "read error on <path>").
// The decoder is this file's own (zlib cannot leave markers): canonical Huffman tables, 11 / 8 root bits, a 64-bit bit buffer.
#include "kg_ingest.hpp"

#include "../../include/katgpu.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

namespace kg {
namespace {

uint64_t penv(const char* name, uint64_t def) {
    const char* v = getenv(name);
    if (!v || !*v) return def;
    char* e = nullptr;
    const unsigned long long x = strtoull(v, &e, 10);
    return e && *e == 0 ? (uint64_t)x : def;
}

// CPUs this process may really use: the hardware's, cut down to the scheduler affinity and to the cgroup's CPU quota (a container
// that shows 256 CPUs and is throttled to 16 CPU-seconds per second runs a team of 64 slower than a team of 12 -- and everything
// beside the team slower still: measured, profiles/r06_pgz_host.txt)
unsigned effective_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, (unsigned)c); }
    auto quota = [&](const char* path_max, const char* path_q, const char* path_p) {
        double q = -1, per = 100000;
        if (FILE* f = fopen(path_max, "r")) {                   // cgroup v2: "max 100000" | "1600000 100000"
            char a[64] = {0};
            if (fscanf(f, "%63s %lf", a, &per) >= 1 && strcmp(a, "max") != 0) q = atof(a);
            fclose(f);
        } else if (FILE* f1 = fopen(path_q, "r")) {             // cgroup v1
            if (fscanf(f1, "%lf", &q) != 1) q = -1;
            fclose(f1);
            if (FILE* f2 = fopen(path_p, "r")) { if (fscanf(f2, "%lf", &per) != 1) per = 100000; fclose(f2); }
        }
        if (q > 0 && per > 0) n = std::min(n, (unsigned)std::max(1.0, q / per + 0.5));
    };
    quota("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    return n;
}

// ------------------------------------------------------------------ bits ------------------------------------------------
struct Bits {
    const uint8_t* base = nullptr; const uint8_t* p = nullptr; const uint8_t* end = nullptr;
    uint64_t buf = 0; unsigned n = 0;      // n valid bits in buf (LSB first)
    uint64_t pad = 0;                      // bytes invented past the end of the input (zeros): the stream was truncated if any of them is consumed
    void init(const uint8_t* b, const uint8_t* e, uint64_t bitpos) {
        base = b; end = e; p = b + (bitpos >> 3); buf = 0; n = 0; pad = 0;
        refill();
        const unsigned skip = (unsigned)(bitpos & 7);
        buf >>= skip; n -= skip;
    }
    inline void refill() {                 // afterwards n >= 56
        if (p + 8 <= end) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << n;
            p += (63 - n) >> 3;
            n |= 56;
        } else {
            while (n <= 56) {
                if (p < end) buf |= (uint64_t)*p++ << n; else ++pad;
                n += 8;
            }
        }
    }
    inline uint32_t peek(unsigned k) const { return (uint32_t)(buf & ((1ULL << k) - 1)); }
    inline void drop(unsigned k) { buf >>= k; n -= k; }
    inline uint32_t take(unsigned k) { const uint32_t v = peek(k); drop(k); return v; }
    uint64_t bitpos() const { return (uint64_t)((p - base) + (int64_t)pad) * 8 - n; }
    bool past_end() const { return bitpos() > (uint64_t)(end - base) * 8; }
    void align_byte() { drop(n & 7); }
};

// ------------------------------------------------------------------ Huffman tables --------------------------------------
// entry: val << 16 | op << 8 | bits.  op: 0 literal, 16 | extra: a base with `extra` more bits, 32 end of block, 64 invalid, 128 | sub: link
constexpr unsigned LROOT = 11, DROOT = 8;
constexpr uint32_t OP_BASE = 16, OP_EOB = 32, OP_BAD = 64, OP_LINK = 128;
struct Table { uint32_t e[2048 + 1024]; unsigned root; };
const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t bitrev(uint32_t v, unsigned len) {
    uint32_t r = 0;
    for (unsigned i = 0; i < len; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}
enum Kind { K_CODES, K_LENS, K_DISTS };
// zlib's rules (inftrees.c): over-subscribed -> invalid; incomplete -> invalid unless the code is a single symbol of length 1 (and not
// the code-length code); no symbol at all -> a table whose every entry is invalid (legal for a block that never uses it)
bool build_table(Kind kind, const uint8_t* lens, unsigned n_sym, Table* t) {
    const unsigned root = kind == K_LENS ? LROOT : kind == K_DISTS ? DROOT : 7;
    t->root = root;
    unsigned count[16] = {0};
    for (unsigned s = 0; s < n_sym; ++s) ++count[lens[s]];
    unsigned maxl = 15;
    while (maxl > 0 && !count[maxl]) --maxl;
    const uint32_t bad = OP_BAD << 8 | 1;
    if (maxl == 0) { for (unsigned i = 0; i < (1u << root); ++i) t->e[i] = bad; return kind != K_CODES; }
    int left = 1;
    for (unsigned l = 1; l <= 15; ++l) { left <<= 1; left -= (int)count[l]; if (left < 0) return false; }
    if (left > 0 && (kind == K_CODES || maxl != 1)) return false;
    uint32_t next[16];
    { uint32_t code = 0; unsigned prev = 0; for (unsigned l = 1; l <= 15; ++l) { code = (code + prev) << 1; next[l] = code; prev = count[l]; } }
    for (unsigned i = 0; i < (1u << root); ++i) t->e[i] = bad;
    // sub-tables: the longest code behind every root prefix
    uint8_t sub_bits[1u << LROOT];
    if (maxl > root) memset(sub_bits, 0, (size_t)1 << root);
    uint32_t codes[288];
    for (unsigned s = 0; s < n_sym; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        codes[s] = bitrev(next[l]++, l);
        if (l > root) { uint8_t& b = sub_bits[codes[s] & ((1u << root) - 1)]; b = std::max<uint8_t>(b, (uint8_t)(l - root)); }
    }
    unsigned used = 1u << root;
    if (maxl > root)
        for (unsigned i = 0; i < (1u << root); ++i)
            if (sub_bits[i]) {
                if (used + (1u << sub_bits[i]) > sizeof t->e / sizeof t->e[0]) return false;
                t->e[i] = (uint32_t)used << 16 | (OP_LINK | sub_bits[i]) << 8 | root;
                for (unsigned q = 0; q < (1u << sub_bits[i]); ++q) t->e[used + q] = bad;
                used += 1u << sub_bits[i];
            }
    for (unsigned s = 0; s < n_sym; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        uint32_t val, op;
        if (kind == K_LENS) {
            if (s < 256) { val = s; op = 0; }
            else if (s == 256) { val = 0; op = OP_EOB; }
            else if (s < 286) { val = LEN_BASE[s - 257]; op = OP_BASE | LEN_EXTRA[s - 257]; }
            else { val = 0; op = OP_BAD; }
        } else if (kind == K_DISTS) {
            if (s < 30) { val = DIST_BASE[s]; op = OP_BASE | DIST_EXTRA[s]; } else { val = 0; op = OP_BAD; }
        } else { val = s; op = 0; }
        if (l <= root) {
            const uint32_t ent = val << 16 | op << 8 | l;
            for (uint32_t i = codes[s]; i < (1u << root); i += 1u << l) t->e[i] = ent;
        } else {
            const uint32_t link = t->e[codes[s] & ((1u << root) - 1)];
            const unsigned sb = (link >> 8) & 15, off = link >> 16;
            const uint32_t ent = val << 16 | op << 8 | (l - root);
            for (uint32_t i = codes[s] >> root; i < (1u << sb); i += 1u << (l - root)) t->e[off + i] = ent;
        }
    }
    return true;
}
inline uint32_t lookup(const Table& t, Bits& b) {
    uint32_t e = t.e[b.peek(t.root)];
    if (e & (OP_LINK << 8)) { b.drop(t.root); e = t.e[(e >> 16) + b.peek((e >> 8) & 15)]; }
    b.drop(e & 0xFF);
    return e;
}

struct Codes { Table lit, dist; };
// a dynamic block's header behind its 3 header bits: the two codes.  false: not a valid header (or the input ends in it)
bool read_dynamic(Bits& b, Codes* c) {
    b.refill();
    const unsigned hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; ++i) { if (b.n < 3) b.refill(); cl[CL_ORDER[i]] = (uint8_t)b.take(3); }
    Table clt;
    if (!build_table(K_CODES, cl, 19, &clt)) return false;
    uint8_t lens[286 + 30 + 138];
    unsigned i = 0;
    while (i < hlit + hdist) {
        b.refill();
        const uint32_t e = lookup(clt, b);
        if ((e >> 8) & OP_BAD) return false;
        const unsigned s = e >> 16;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep; uint8_t v = 0;
        if (s == 16) { if (!i) return false; v = lens[i - 1]; rep = 3 + b.take(2); }
        else if (s == 17) rep = 3 + b.take(3);
        else rep = 11 + b.take(7);
        if (i + rep > hlit + hdist) return false;
        memset(lens + i, v, rep);
        i += rep;
    }
    if (b.past_end()) return false;
    if (!lens[256]) return false;                              // no end-of-block code
    return build_table(K_LENS, lens, hlit, &c->lit) && build_table(K_DISTS, lens + hlit, hdist, &c->dist);
}
const Codes& fixed_codes() {
    static const Codes* f = [] {
        Codes* c = new Codes;
        uint8_t l[288];
        for (unsigned s = 0; s < 288; ++s) l[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        build_table(K_LENS, l, 288, &c->lit);
        uint8_t d[32];
        memset(d, 5, 32);
        build_table(K_DISTS, d, 32, &c->dist);
        return c;
    }();
    return *f;
}

// ------------------------------------------------------------------ one chunk's output ----------------------------------
constexpr uint32_t WIN = 32768;
struct MemberEnd { uint64_t at; uint32_t crc, isize; };        // a member ends after `at` symbols of this chunk's output
// a buffer that is not cleared when it grows: anonymous memory the kernel is asked to back with huge pages (first touches of fresh
// memory serialise on the process's address space -- 512 times fewer of them --; the buffers are reused besides)
template <class T> struct Raw {
    T* p = nullptr; size_t cap = 0;
    Raw() = default;
    Raw(Raw&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    Raw& operator=(Raw&& o) noexcept { if (this != &o) { drop(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    Raw(const Raw&) = delete; Raw& operator=(const Raw&) = delete;
    ~Raw() { drop(); }
    void drop() { if (p) munmap((void*)p, bytes_of(cap)); p = nullptr; cap = 0; }
    static size_t bytes_of(size_t n) { return (n * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return cap; }
    size_t capacity() const { return cap; }
    bool reserve(size_t n, size_t keep = ~(size_t)0) {          // keep: how many elements are worth copying
        if (n <= cap) return true;
        const size_t bytes = bytes_of(n);
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return false;
        static const bool huge = penv("KATGPU_PGZ_HUGE", 1) != 0;
        if (huge) madvise(m, bytes, MADV_HUGEPAGE);
        if (p) { memcpy(m, p, std::min(keep, cap) * sizeof(T)); munmap((void*)p, bytes_of(cap)); }
        p = (T*)m; cap = bytes / sizeof(T);
        return true;
    }
    T& operator[](size_t i) { return p[i]; }
};
struct Out {
    Raw<uint16_t> v;               // symbols: < 256 a byte; 256 + i: byte i of the WIN bytes before the chunk's first
    uint64_t n = 0;
    int64_t floor = 0;             // no copy may reach below this index (the member's first byte; -WIN while the history is the unknown window)
    std::vector<MemberEnd> members;
    bool oom = false;
    void room(size_t more) { if (v.size() < n + more && !v.reserve(std::max<size_t>(v.size() + v.size() / 2, n + more + (1 << 20)), n)) oom = true; }
};
enum Stop { S_BLOCK_END, S_ERROR };

// the symbols of one block behind its tables.  TEXT: every literal must be a text byte (the sync search's test).  false: bad data
template <bool TEXT>
bool inflate_block(Bits& b, const Codes& c, Out& o, uint64_t max_symbols) {
    const uint64_t stop_at = o.n + max_symbols;
    for (;;) {
        o.room(600);
        if (o.oom) return false;
        uint16_t* out = o.v.data();
        uint64_t n = o.n;
        const uint64_t lim = std::min<uint64_t>(o.v.size() - 300, stop_at);
        while (n < lim) {
            b.refill();
            uint32_t e = lookup(c.lit, b);
            uint32_t op = (e >> 8) & 0xFF;
            if (op == 0) {
                if (TEXT) { const uint32_t ch = e >> 16; if (!((ch >= 32 && ch < 127) || ch == '\n' || ch == '\r' || ch == '\t')) return false; }
                out[n++] = (uint16_t)(e >> 16);
                e = lookup(c.lit, b);                            // (a second symbol on the same refill: 2 x 15 bits + what a match needs still fit 56)
                op = (e >> 8) & 0xFF;
                if (op == 0) {
                    if (TEXT) { const uint32_t ch = e >> 16; if (!((ch >= 32 && ch < 127) || ch == '\n' || ch == '\r' || ch == '\t')) return false; }
                    out[n++] = (uint16_t)(e >> 16);
                    continue;
                }
                if (b.n < 48) b.refill();
            }
            if (op & OP_BASE) {
                const uint32_t len = (e >> 16) + b.take(op & 15);
                const uint32_t d = lookup(c.dist, b);
                const uint32_t dop = (d >> 8) & 0xFF;
                if (!(dop & OP_BASE)) return false;
                const uint32_t dist = (d >> 16) + b.take(dop & 15);
                const int64_t from = (int64_t)n - (int64_t)dist;
                if (from < o.floor) return false;                // too far back
                if (from >= 0) {
                    const uint16_t* src = out + from;
                    uint16_t* dst = out + n;
                    if (dist >= 8) {                              // eight symbols at a time, past the match's end if need be (the buffer has the slack;
                        uint32_t i = 0;                           //  FASTQ is mostly short matches: a libc memcpy per match costs more than the copy)
                        do { uint64_t a, b2; memcpy(&a, src + i, 8); memcpy(&b2, src + i + 4, 8); memcpy(dst + i, &a, 8); memcpy(dst + i + 4, &b2, 8); i += 8; } while (i < len);
                    } else for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
                    n += len;
                } else {
                    for (uint32_t i = 0; i < len; ++i) { const int64_t f = from + i; out[n + i] = f < 0 ? (uint16_t)(256 + WIN + f) : out[f]; }
                    n += len;
                }
                continue;
            }
            if (op & OP_EOB) { o.n = n; return !b.past_end(); }
            return false;                                      // an invalid code
        }
        o.n = n;
        if (b.past_end()) return false;
        if (n >= stop_at) return false;                        // (the sync search's bound on a block: not a block)
    }
}

// One block at the reader's position.  *final: it was the member's last.  false: bad data.
bool inflate_one(Bits& b, Out& o, bool* final, Codes* scratch) {
    b.refill();
    *final = b.take(1) != 0;
    const uint32_t type = b.take(2);
    if (type == 0) {
        b.align_byte();
        b.refill();
        const uint32_t len = b.take(16), nlen = b.take(16);
        if ((len ^ nlen) != 0xFFFF) return false;
        o.room(len + 16);
        if (o.oom) return false;
        for (uint32_t i = 0; i < len; ++i) { if (b.n < 8) b.refill(); o.v[o.n++] = (uint16_t)b.take(8); }
        return !b.past_end();
    }
    if (type == 1) return inflate_block<false>(b, fixed_codes(), o, ~0ULL >> 1);
    if (type == 2) return read_dynamic(b, scratch) && inflate_block<false>(b, *scratch, o, ~0ULL >> 1);
    return false;
}

// a gzip member header at byte `at`; returns the offset of its deflate data, 0: not a member header, or truncated
size_t gzip_header(const uint8_t* d, size_t size, size_t at) {
    if (at + 18 > size || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8 || (d[at + 3] & 0xE0)) return 0;
    const uint8_t flg = d[at + 3];
    size_t p = at + 10;
    if (flg & 4) { if (p + 2 > size) return 0; p += 2 + (d[p] | d[p + 1] << 8); }
    if (flg & 8) { while (p < size && d[p]) ++p; ++p; }
    if (flg & 16) { while (p < size && d[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    return p < size ? p : 0;
}

// ------------------------------------------------------------------ the sync search -------------------------------------
// Is there a dynamic block (not the member's last) at bit `at`, a whole block of text, with something that looks like a block behind it?
bool plausible_block_start(const uint8_t* d, size_t size, uint64_t at, Out& scratch, Codes* c1, Codes* c2) {
    Bits b;
    b.init(d, d + size, at);
    const uint32_t h = b.peek(17);
    if ((h & 7) != 4) return false;                             // BFINAL 0, BTYPE 2
    if (((h >> 3) & 31) > 29 || ((h >> 8) & 31) > 29) return false;
    {   // the code-length code must be complete: sum over its symbols of 2^(7 - len) == 128
        const unsigned hclen = ((h >> 13) & 15) + 4;
        Bits q = b;
        q.drop(17);
        unsigned sum = 0;
        for (unsigned i = 0; i < hclen; ++i) { if (q.n < 3) q.refill(); const unsigned l = q.take(3); if (l) sum += 128u >> l; }
        if (sum != 128) return false;
    }
    b.drop(3);
    if (!read_dynamic(b, c1)) return false;
    scratch.n = 0; scratch.floor = -(int64_t)WIN; scratch.members.clear();
    if (!inflate_block<true>(b, *c1, scratch, (uint64_t)4 << 20)) return false;
    if (scratch.n < 64) return false;                          // (too little to tell text from chance)
    // what follows: a block header of any kind
    b.refill();
    const uint32_t h2 = b.peek(3);
    const uint32_t type = (h2 >> 1) & 3;
    if (type == 3) return false;
    if (type == 0) {
        Bits q = b; q.drop(3); q.align_byte(); q.refill();
        const uint32_t len = q.take(16), nlen = q.take(16);
        return (len ^ nlen) == 0xFFFF && !q.past_end();
    }
    if (type == 1) return !b.past_end();
    b.drop(3);
    return read_dynamic(b, c2);
}

// ------------------------------------------------------------------ the team ---------------------------------------------
constexpr uint64_t NONE = ~0ULL;
struct Chunk {
    std::mutex sync_mu;
    std::atomic<int> sync_known{0};
    uint64_t sync = NONE;                     // bit position of the block start the chunk is entered at (chunk 0: its member's first block)
    Out out;
    int64_t next_live = -1;                   // the chunk whose sync this one's decoder ended on; -1: it reached the end of the stream
    bool known_start = false;                 // chunk 0: no history (floor 0)
    bool error = false, taken = false;
    std::atomic<int> done{0};
};

struct Inflater {
    const uint8_t* d = nullptr; size_t size = 0;
    size_t CB = 0; size_t n_chunks = 0;
    std::unique_ptr<Chunk[]> chunks;
    std::atomic<size_t> next_chunk{0};
    std::atomic<size_t> horizon{0};           // workers take chunks below this index only (the consumer moves it: bounded memory)
    std::atomic<bool> quit{false};
    std::mutex mu; std::condition_variable cv;
    std::vector<std::thread> workers;
    std::mutex pool_mu; std::vector<Raw<uint16_t>> pool; std::vector<Raw<uint8_t>> bpool;     // symbol / byte buffers, reused

    Raw<uint16_t> get_buf() { std::lock_guard<std::mutex> g(pool_mu); if (pool.empty()) return {}; auto v = std::move(pool.back()); pool.pop_back(); return v; }
    void put_buf(Raw<uint16_t>&& v) { if (v.capacity()) { std::lock_guard<std::mutex> g(pool_mu); if (pool.size() < 512) pool.push_back(std::move(v)); } }
    Raw<uint8_t> get_bytes() { std::lock_guard<std::mutex> g(pool_mu); if (bpool.empty()) return {}; auto v = std::move(bpool.back()); bpool.pop_back(); return v; }
    void put_bytes(Raw<uint8_t>&& v) { if (v.capacity()) { std::lock_guard<std::mutex> g(pool_mu); if (bpool.size() < 512) bpool.push_back(std::move(v)); } }

    uint64_t get_sync(size_t j) {             // chunk j's sync: searched once, by whoever needs it first
        Chunk& c = chunks[j];
        if (c.sync_known.load(std::memory_order_acquire)) return c.sync;
        std::lock_guard<std::mutex> g(c.sync_mu);
        if (c.sync_known.load(std::memory_order_acquire)) return c.sync;
        uint64_t found = NONE;
        if (j > 0) {
            Out scratch; scratch.v = get_buf();
            std::unique_ptr<Codes> c1(new Codes), c2(new Codes);
            const uint64_t lo = (uint64_t)j * CB * 8, hi = std::min<uint64_t>((uint64_t)(j + 1) * CB, size) * 8;
            for (uint64_t at = lo; at < hi; ++at) {
                // (three cheap bits first: most positions end here)
                const uint8_t by = d[at >> 3];
                const unsigned sh = (unsigned)(at & 7);
                const uint32_t three = sh <= 5 ? (by >> sh) & 7 : ((by | (uint32_t)((at >> 3) + 1 < size ? d[(at >> 3) + 1] : 0) << 8) >> sh) & 7;
                if (three != 4) continue;
                if (plausible_block_start(d, size, at, scratch, c1.get(), c2.get())) { found = at; break; }
            }
            put_buf(std::move(scratch.v));
        }
        c.sync = found;
        c.sync_known.store(1, std::memory_order_release);
        return found;
    }

    // chunk j: decode from its sync until a block ends on a later chunk's sync (or the stream ends)
    void decode(size_t j) {
        Chunk& c = chunks[j];
        const uint64_t start = j == 0 ? c.sync : get_sync(j);
        if (start == NONE) { c.error = false; c.next_live = -2; return; }      // no way in: the chunk before decodes through (this one is never live)
        Out& o = c.out;
        o.v = get_buf();
        o.n = 0; o.members.clear();
        o.floor = c.known_start ? 0 : -(int64_t)WIN;
        o.room(CB * 5);
        Bits b;
        b.init(d, d + size, start);
        std::unique_ptr<Codes> scratch(new Codes);
        size_t checked = j;                                     // chunks up to here have been compared with (their syncs lie behind)
        for (;;) {
            bool final = false;
            if (!inflate_one(b, o, &final, scratch.get())) { c.error = true; return; }
            if (final) {                                        // the member's trailer; another member, or the end
                b.align_byte();
                uint64_t at = b.bitpos() >> 3;
                if (at + 8 > size) { c.error = true; return; }
                MemberEnd m;
                m.at = o.n;
                m.crc = d[at] | d[at + 1] << 8 | d[at + 2] << 16 | (uint32_t)d[at + 3] << 24;
                m.isize = d[at + 4] | d[at + 5] << 8 | d[at + 6] << 16 | (uint32_t)d[at + 7] << 24;
                o.members.push_back(m);
                at += 8;
                const size_t data = at < size ? gzip_header(d, size, at) : 0;
                if (!data) { c.next_live = -1; return; }        // the end (what follows, if anything, is not gzip: ignored, as zlib does)
                o.floor = (int64_t)o.n;
                b.init(d, d + size, (uint64_t)data * 8);
            }
            // a block starts here: is it a later chunk's way in?
            const uint64_t pos = b.bitpos();
            const size_t q = (size_t)std::min<uint64_t>(pos / (CB * 8), n_chunks - 1);
            while (checked < q) {
                const uint64_t s = get_sync(checked + 1);
                if (s == pos) { c.next_live = (int64_t)(checked + 1); return; }
                if (s != NONE && s > pos) break;                // still ahead (inside chunk `checked + 1`: q == checked + 1)
                ++checked;                                      // passed over: that chunk is not live
            }
            if (quit.load(std::memory_order_relaxed)) { c.error = true; return; }
        }
    }

    // One team for both kinds of work: a thread takes a chunk's second job (markers -> bytes, CRC, parse) if one waits -- they free
    // buffers and let the walk on -- and the next chunk to decode otherwise: whatever the two cost against each other, every CPU is busy.
    std::deque<std::function<void()>> job_q;
    void work() {
        for (;;) {
            std::function<void()> f;
            size_t j = ~(size_t)0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit.load() || !job_q.empty() || (next_chunk.load() < n_chunks && next_chunk.load() < horizon.load()); });
                if (!job_q.empty()) { f = std::move(job_q.front()); job_q.pop_front(); }
                else if (quit.load()) return;
                else j = next_chunk.fetch_add(1);
            }
            if (f) { f(); continue; }
            decode(j);
            { std::lock_guard<std::mutex> g(mu); chunks[j].done.store(1, std::memory_order_release); }
            cv.notify_all();
        }
    }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> g(mu); job_q.push_back(std::move(f)); } cv.notify_all(); }
    void start(unsigned T) { for (unsigned i = 0; i < T; ++i) workers.emplace_back([this] { work(); }); }
    void stop() {                                               // (every job handed over has been run by then: the pieces' taker waits for them first)
        { std::lock_guard<std::mutex> g(mu); quit.store(true); }
        cv.notify_all();
        for (auto& t : workers) t.join();
        workers.clear();
    }
    void wait_done(size_t j) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return chunks[j].done.load(std::memory_order_acquire) != 0; }); }
    void move_horizon(size_t h) { { std::lock_guard<std::mutex> g(mu); if (h > horizon.load()) horizon.store(h); } cv.notify_all(); }
};

struct Mapped {
    const uint8_t* d = nullptr; size_t size = 0; int fd = -1;
    ~Mapped() { if (d) munmap((void*)d, size); if (fd >= 0) ::close(fd); }
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return false;
        size = (size_t)st.st_size;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        d = (const uint8_t*)m;
        madvise(m, size, MADV_SEQUENTIAL);
        return true;
    }
};

// CRC-32 (the gzip polynomial) by carry-less multiplication where the host has it: four 128-bit lanes folded per 64 bytes, then 128 -> 64 ->
// 32 bits with a Barrett reduction (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Intel 2009; the
// constants are x^(64 k) mod P for the fold distances, P's bit-reflected form).  zlib 1.2.11's table walk does ~1 GB/s per thread --
// as much time as inflating the bytes; this does ~10.  Checked against zlib's at start-up: a host where it disagrees keeps zlib's.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1")))
uint32_t crc32_clmul(const uint8_t* buf, size_t len /* a multiple of 16, >= 64 */, uint32_t crc) {
    static const uint64_t __attribute__((aligned(16))) k1k2[] = {0x0154442bd4ULL, 0x01c6e41596ULL};
    static const uint64_t __attribute__((aligned(16))) k3k4[] = {0x01751997d0ULL, 0x00ccaa009eULL};
    static const uint64_t __attribute__((aligned(16))) k5k0[] = {0x0163cd6124ULL, 0x0000000000ULL};
    static const uint64_t __attribute__((aligned(16))) poly[] = {0x01db710641ULL, 0x01f7011641ULL};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
bool clmul_ok() {
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t t[1024 + 16];
        for (size_t i = 0; i < sizeof t; ++i) t[i] = (uint8_t)(i * 131 + (i >> 3) * 7 + 5);
        for (size_t len : {(size_t)64, (size_t)80, (size_t)512, (size_t)1024}) {
            const uint32_t seed = 0x12345678u ^ (uint32_t)len;
            if (~crc32_clmul(t + 1, len, ~seed) != (uint32_t)crc32(seed, t + 1, (uInt)len)) return false;
        }
        return true;
    }();
    return ok;
}
#else
bool clmul_ok() { return false; }
uint32_t crc32_clmul(const uint8_t*, size_t, uint32_t c) { return c; }
#endif
// crc32(crc, p, n) of zlib, faster
uint32_t fast_crc32(uint32_t crc, const uint8_t* p, size_t n) {
    if (n >= 256 && clmul_ok()) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_clmul(p, body, ~crc);
        p += body; n -= body;
    }
    for (; n; ) { const uInt m = (uInt)std::min<size_t>(n, (size_t)1 << 30); crc = (uint32_t)crc32(crc, p, m); p += m; n -= m; }
    return crc;
}

// markers -> bytes with the WIN bytes before the chunk (win[WIN - have, WIN) are known)
bool translate(const uint16_t* v, uint64_t n, const uint8_t* win, uint32_t have, uint8_t* out) {
    const uint32_t lo = 256 + (WIN - have);
    uint32_t bad = 0;
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {                                 // eight symbols at a time: past a chunk's first tens of KB hardly any is a marker
        uint64_t a, b;
        memcpy(&a, v + i, 8);
        memcpy(&b, v + i + 4, 8);
        if (((a | b) & 0xFF00FF00FF00FF00ULL) == 0) {
            a = (a | (a >> 8)) & 0x0000FFFF0000FFFFULL; a = (a | (a >> 16)) & 0xFFFFFFFFULL;
            b = (b | (b >> 8)) & 0x0000FFFF0000FFFFULL; b = (b | (b >> 16)) & 0xFFFFFFFFULL;
            const uint64_t w = a | (b << 32);
            memcpy(out + i, &w, 8);
            continue;
        }
        for (unsigned q = 0; q < 8; ++q) {
            const uint32_t s = v[i + q];
            if (s < 256) out[i + q] = (uint8_t)s;
            else { bad |= s < lo; out[i + q] = win[s - 256]; }
        }
    }
    for (; i < n; ++i) {
        const uint32_t s = v[i];
        if (s < 256) out[i] = (uint8_t)s;
        else { bad |= s < lo; out[i] = win[s - 256]; }
    }
    return !bad;
}

// What the consumer gets of one live chunk
struct Piece {
    Raw<uint8_t> buf; size_t n_bytes = 0;     // the chunk's inflated bytes
    struct View { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } const uint8_t* begin() const { return p; } const uint8_t* end() const { return p + n; } };
    View bytes{nullptr, 0};
    int64_t s = -1, e = -1;                   // [s, e): parsed here, from a guessed record start to the last one (-1: nothing was)
    std::vector<uint8_t> parsed;
    ParseState end_state;
    bool bad_parse = false, bad_window = false;
    std::vector<uint32_t> crcs;               // CRC-32 of the byte runs between the chunk's member ends (one more run than ends)
    std::vector<MemberEnd> members;
};

}  // namespace

bool pgz_applies(const char* path, uint32_t trim5p) {
    if (trim5p) return false;                                   // (as the plain-file team: is.ignore(trim5p) swallows line starts)
    if (penv("KATGPU_PGZ", 1) == 0) return false;
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if ((uint64_t)st.st_size < penv("KATGPU_PGZ_MIN_BYTES", (uint64_t)8 << 20)) return false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t head[4] = {0, 0, 0, 0};
    const bool got = pread(fd, head, 4, 0) == 4;
    ::close(fd);
    return got && head[0] == 0x1f && head[1] == 0x8b && head[2] == 8 && !bgzf_applies(path);
}

// The inflated file, chunk by chunk in order, to `on_piece(piece, window of the bytes before it)`; parse_type NONE: bytes only.
static int inflate_team(const char* path, bool parse, const std::function<int(Piece&, std::string*)>& on_piece, std::string* err) {
    Mapped f;
    if (!f.open(path)) return -1;
    const size_t data0 = gzip_header(f.d, f.size, 0);
    if (!data0) return -1;
    // the team: every CPU this process may use but two (the walk, and what takes the pieces)
    const unsigned cpus = effective_cpus();
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(penv("KATGPU_PGZ_THREADS", std::min(48u, std::max(2u, cpus > 4 ? cpus - 2 : cpus))), 256));
    Inflater inf;
    inf.d = f.d; inf.size = f.size;
    inf.CB = (size_t)std::max<uint64_t>(1 << 16, penv("KATGPU_PGZ_CHUNK", (uint64_t)8 << 20));
    inf.n_chunks = (f.size + inf.CB - 1) / inf.CB;
    inf.chunks.reset(new Chunk[inf.n_chunks]);
    inf.chunks[0].sync = (uint64_t)data0 * 8; inf.chunks[0].sync_known.store(1); inf.chunks[0].known_start = true;
    // A stream that offers no way in -- stored or fixed-Huffman blocks only, data that is not text -- would have the first chunk's decoder
    // go through the whole file alone, its output growing with it: look for ONE entry point in the first megabytes behind chunk 0 before
    // anything is started; a file without one is left to zlib's single stream (nothing has been handed on yet).
    if (inf.n_chunks > 1 && penv("KATGPU_PGZ_PROBE", 1) != 0) {
        bool any = false;
        const size_t probe_chunks = std::min<size_t>(inf.n_chunks - 1, std::max<size_t>(2, ((size_t)4 << 20) / inf.CB));
        for (size_t q = 1; q <= probe_chunks && !any; ++q) any = inf.get_sync(q) != NONE;
        if (!any) {
            if (getenv("KATGPU_TRACE")) fprintf(stderr, "[katgpu] ingest %s: no deflate block of text starts in the %zu chunk(s) behind the first: one zlib stream\n", path, probe_chunks);
            return -1;
        }
    }
    const size_t ahead = (size_t)T + 4;                          // (chunks in flight: every one owns tens of MB of buffers -- few enough that they are reused early)
    inf.move_horizon(ahead);
    inf.start(T);
    struct Stopper { Inflater& i; ~Stopper() { i.stop(); } } stopper{inf};
    const bool verify = penv("KATGPU_PGZ_VERIFY", 1) != 0;
    const bool trace = getenv("KATGPU_TRACE") != nullptr;

    std::vector<uint8_t> window(WIN, 0);
    uint32_t have = 0;                                          // known bytes at the window's end (a member's first WIN bytes have less history)
    ParseState::Type ptype = ParseState::NONE;
    // The pieces, in order, to whoever takes them -- on a thread of its own: what takes them (the state machine between the chunks' cuts,
    // the sink's copy towards the device, its waits for the counter) must not hold up the walk that lets the decoders on.
    std::deque<std::future<std::unique_ptr<Piece>>> jobs;
    std::mutex jq_mu; std::condition_variable jq_cv;
    bool jq_closed = false;
    std::atomic<int> drain_rc{KATGPU_OK};
    std::string drain_err;
    size_t live = 0, dropped = 0;
    int rc = KATGPU_OK;
    double t_job = 0, t_piece = 0, t_full = 0;
    const size_t max_jobs = 2 * (size_t)T + 4;
    auto clock_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::thread drain([&] {
        for (;;) {
            std::future<std::unique_ptr<Piece>> f;
            {
                std::unique_lock<std::mutex> lk(jq_mu);
                jq_cv.wait(lk, [&] { return jq_closed || !jobs.empty(); });
                if (jobs.empty()) return;
                f = std::move(jobs.front());
                jobs.pop_front();
            }
            jq_cv.notify_all();
            const double t0 = clock_ms();
            std::unique_ptr<Piece> p = f.get();
            const double t1 = clock_ms();
            t_job += t1 - t0;
            if (drain_rc.load() == KATGPU_OK) {                  // (after an error the rest is only waited for)
                if (p->bad_window) { drain_err = std::string("read error on ") + path; drain_rc.store(KATGPU_ERR_IO); }
                else {
                    std::string e;
                    const int r = on_piece(*p, &e);
                    if (r) { drain_err = e; drain_rc.store(r); }
                }
            }
            inf.put_bytes(std::move(p->buf));
            t_piece += clock_ms() - t1;
        }
    });
    auto close_drain = [&] {
        { std::lock_guard<std::mutex> g(jq_mu); jq_closed = true; }
        jq_cv.notify_all();
        if (drain.joinable()) drain.join();
    };
    struct DrainGuard { std::function<void()> f; ~DrainGuard() { f(); } } drain_guard{close_drain};
    int64_t j = 0;
    auto now_ms = clock_ms;
    double ms_wait_decode = 0, ms_launch = 0;
    const double t_start = now_ms();
    while (j >= 0) {
        { const double t0 = now_ms(); inf.wait_done((size_t)j); ms_wait_decode += now_ms() - t0; }
        Chunk& c = inf.chunks[(size_t)j];
        if (c.error || c.next_live == -2) { *err = std::string("read error on ") + path; rc = KATGPU_ERR_IO; break; }
        ++live;
        Out& o = c.out;
        if (ptype == ParseState::NONE && parse && o.n) {         // the format, from the file's first byte (chunk 0 has no markers)
            const uint16_t first = o.v[0];
            ptype = first == '>' ? ParseState::FASTA : first == '@' ? ParseState::FASTQ : ParseState::NONE;
            if (ptype == ParseState::NONE) { *err = "Unsupported format"; rc = KATGPU_ERR_FORMAT; break; }
        }
        // the job: markers -> bytes with a copy of the window as it stands; CRCs; the state machine between the guessed cuts
        const double t_l0 = now_ms();
        auto win = std::make_shared<std::vector<uint8_t>>(window);
        const uint32_t have_now = have;
        std::shared_ptr<Out> op(new Out(std::move(o)));
        // the next chunk's window: this chunk's last WIN bytes, resolved here (WIN look-ups)
        {
            const uint64_t n = op->n;
            // (member ends inside the chunk reset the history: only bytes of the last member count)
            const uint64_t from = op->members.empty() ? 0 : op->members.back().at;
            const uint64_t tail = std::min<uint64_t>(n - from, WIN);
            std::vector<uint8_t> nw(WIN, 0);
            uint32_t nhave;
            bool ok = true;
            if (tail < WIN && op->members.empty()) {             // shorter than the window: the old window's end stays in front
                const uint32_t keep = (uint32_t)std::min<uint64_t>(have, WIN - tail);
                memcpy(nw.data() + WIN - tail - keep, window.data() + WIN - keep, keep);
                nhave = keep + (uint32_t)tail;
            } else nhave = (uint32_t)tail;
            ok = translate(op->v.data() + (n - tail), tail, window.data(), have, nw.data() + WIN - tail);
            if (!ok) { *err = std::string("read error on ") + path; rc = KATGPU_ERR_IO; break; }
            window.swap(nw);
            have = nhave;
        }
        Inflater* infp = &inf;
        static const bool strip_ok = penv("KATGPU_PGZ_STRIP", 1) != 0;
        auto task = std::make_shared<std::packaged_task<std::unique_ptr<Piece>()>>([op, win, have_now, ptype, verify, infp]() {
            std::unique_ptr<Piece> p(new Piece);
            p->buf = infp->get_bytes();
            if (!p->buf.reserve(std::max<size_t>(op->n + 64, infp->CB * 5), 0)) { p->bad_window = true; return p; }      // (with room to spare: the next chunk it serves is a little larger as often as not)
            p->n_bytes = op->n;
            p->bytes = Piece::View{p->buf.data(), p->n_bytes};
            p->bad_window = !translate(op->v.data(), op->n, win->data(), have_now, p->buf.data());
            p->members = op->members;
            infp->put_buf(std::move(op->v));
            if (p->bad_window) return p;
            if (verify) {
                uint64_t a = 0;
                for (size_t m = 0; m <= p->members.size(); ++m) {
                    const uint64_t e = m < p->members.size() ? p->members[m].at : p->bytes.size();
                    p->crcs.push_back(fast_crc32((uint32_t)crc32(0L, Z_NULL, 0), p->bytes.data() + a, (size_t)(e - a)));
                    a = e;
                }
            }
            if (ptype != ParseState::NONE) {
                const int64_t n = (int64_t)p->bytes.size();
                const int64_t s = find_record_start(ptype, p->bytes.data(), 0, n, 1);
                int64_t e = s < 0 ? -1 : find_record_start(ptype, p->bytes.data(), 0, n, std::max<int64_t>(s, n - (256 << 10)));
                if (s >= 0 && e < 0) e = s;
                if (s >= 0 && e > s) {
                    ParseState ps; ps.type = ptype;
                    ps.st = ptype == ParseState::FASTA ? ParseState::LOOP_CHECK : ParseState::QUAL_DONE_SKIPNL;
                    const uint8_t* const b = p->bytes.data();
                    bool stripped = false;
                    if (ptype == ParseState::FASTQ && strip_ok) {
                        // Plain four-line records (nearly every FASTQ file): the reader threads' strip loop (kg_ingest: strip_fastq_records,
                        // four memchr and one memcpy per record) instead of the state machine, twenty times its rate.  It writes
                        // "sequence N" per record where the machine, from a record boundary, writes "N sequence": the same bytes, shifted
                        // by one.  The machine's state behind the piece is the state behind its last record: that one goes through it.
                        p->parsed.resize((size_t)(e - s) / 2 + 2);
                        size_t got = 0;
                        if (strip_fastq_records(b + s, (size_t)(e - s), p->parsed.data() + 1, &got) && got) {
                            p->parsed[0] = 'N';
                            p->parsed.resize(got);                // ('N' + what was written but its last 'N')
                            int64_t last = e - 1;                  // the last record's '@': four line ends back from e
                            for (int nl = 0; nl < 4 && last > s; ++nl) { const void* q = memrchr(b + s, '\n', (size_t)(last - s)); last = q ? (const uint8_t*)q - b : s - 1; }
                            last = last < s ? s : last + 1;
                            std::vector<uint8_t> scratch;
                            bool bad = false;
                            ps.consume(b + last, (size_t)(e - last), scratch, &bad);
                            stripped = !bad && ps.at_record_boundary();
                            if (!stripped) { ps = ParseState(); ps.type = ptype; ps.st = ParseState::QUAL_DONE_SKIPNL; }
                        }
                        if (!stripped) p->parsed.clear();
                    }
                    if (!stripped) {
                        p->parsed.reserve((size_t)(e - s) / 2 + 64);
                        ps.consume(b + s, (size_t)(e - s), p->parsed, &p->bad_parse);
                    }
                    p->end_state = ps;
                    p->s = s; p->e = e;
                }
            }
            return p;
        });
        {
            const double t0 = clock_ms();
            std::unique_lock<std::mutex> lk(jq_mu);
            jq_cv.wait(lk, [&] { return jobs.size() < max_jobs; });
            t_full += clock_ms() - t0;
            jobs.push_back(task->get_future());
        }
        jq_cv.notify_all();
        inf.submit([task] { (*task)(); });
        ms_launch += now_ms() - t_l0;
        if (c.next_live > j + 1) dropped += (size_t)(c.next_live - j - 1);      // (chunks passed over: what their decoders make is never looked at)
        j = c.next_live;
        inf.move_horizon((size_t)(j < 0 ? inf.n_chunks : j) + ahead);
        if (drain_rc.load() != KATGPU_OK) break;
    }
    close_drain();                                              // every piece handed over has been taken (or waited for)
    if (!rc && drain_rc.load() != KATGPU_OK) { rc = drain_rc.load(); *err = drain_err; }
    if (trace) fprintf(stderr, "[katgpu] ingest %s: one gzip stream, %zu chunk(s) of %zu MB by a team of %u (%u CPUs to use), %zu passed over, CRC-32 by %s; %.0f ms: the walk waited %.0f ms for decoders and %.0f ms for room behind it, spent %.0f ms handing chunks to their second jobs; "
                               "behind it %.0f ms were waits for those jobs and %.0f ms went into what takes the pieces\n", path, live, inf.CB >> 20, T, cpus, dropped, !verify ? "nobody" : clmul_ok() ? "carry-less multiplication" : "zlib", now_ms() - t_start,
                       ms_wait_decode, t_full, ms_launch, t_job, t_piece);
    return rc;
}

int parse_gz_parallel(const char* path, uint32_t trim5p, const std::function<int(const uint8_t*, size_t)>& sink, std::string* err) {
    if (!pgz_applies(path, trim5p)) return -1;
    ParseState state;                                           // the machine's TRUE state behind everything handed to the sink so far
    bool begun = false;
    std::vector<uint8_t> out;
    const bool verify = penv("KATGPU_PGZ_VERIFY", 1) != 0;
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    uint64_t member_bytes = 0;
    auto serial = [&](const uint8_t* p, size_t n) -> int {
        if (!n) return KATGPU_OK;
        if (!begun) { if (!state.begin(p[0])) { *err = "Unsupported format"; return KATGPU_ERR_FORMAT; } begun = true; }
        out.clear();
        bool bad = false;
        state.consume(p, n, out, &bad);
        if (bad) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
        if (!out.empty()) { const int rc = sink(out.data(), out.size()); if (rc) { err->clear(); return rc; } }
        return KATGPU_OK;
    };
    auto take = [&](Piece& p) -> int {
        if (verify) {                                           // every member's CRC-32 and length, as zlib checks them
            uint64_t a = 0;
            for (size_t m = 0; m <= p.members.size(); ++m) {
                const uint64_t e = m < p.members.size() ? p.members[m].at : p.bytes.size();
                crc = (uint32_t)crc32_combine(crc, p.crcs[m], (z_off_t)(e - a));
                member_bytes += e - a;
                if (m < p.members.size()) {
                    if (crc != p.members[m].crc || (uint32_t)member_bytes != p.members[m].isize) { *err = std::string("read error on ") + path; return KATGPU_ERR_IO; }
                    crc = (uint32_t)crc32(0L, Z_NULL, 0);
                    member_bytes = 0;
                }
                a = e;
            }
        }
        const size_t n = p.bytes.size();
        if (p.s >= 0 && p.e > p.s) {
            int r = serial(p.bytes.data(), (size_t)p.s);        // up to the guessed record start: through the true machine
            if (r) return r;
            if (begun && state.at_record_boundary() && !p.bad_parse) {      // the guess holds: the team's piece is what the machine would have made of it
                if (!p.parsed.empty()) { r = sink(p.parsed.data(), p.parsed.size()); if (r) { err->clear(); return r; } }
                state = p.end_state;
                return serial(p.bytes.data() + p.e, n - (size_t)p.e);
            }
            return serial(p.bytes.data() + p.s, n - (size_t)p.s);           // it does not (or the piece is bad: the machine words the error)
        }
        return serial(p.bytes.data(), n);
    };
    const int rc = inflate_team(path, true, [&](Piece& p, std::string* e_out) -> int { const int r = take(p); if (r) *e_out = *err; return r; }, err);
    if (rc) return rc;
    if (begun && !state.end_ok()) { *err = "Invalid fastq sequence"; return KATGPU_ERR_FASTQ; }
    return KATGPU_OK;
}

}  // namespace kg

// The inflated bytes of a gzip file through the team (tests, tools): *out is malloc'ed (katgpu_free_host).  KATGPU_ERR_FORMAT: not a
// gzip file the team takes (BGZF, too small, not gzip).
extern "C" int katgpu_inflate_file(const char* path, uint8_t** out, size_t* n, const char** err_msg) {
    static thread_local std::string last;
    if (!path || !n) return KATGPU_ERR_INVALID_ARG;       // (out == null: the bytes are inflated, checked and counted, not kept)
    if (out) *out = nullptr;
    *n = 0;
    last.clear();
    size_t total = 0;
    std::vector<uint8_t> all;
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    uint64_t member_bytes = 0;
    bool crc_bad = false;
    const int rc = kg::inflate_team(path, false, [&](kg::Piece& p, std::string*) -> int {
        uint64_t a = 0;
        for (size_t m = 0; m <= p.members.size() && !p.crcs.empty(); ++m) {
            const uint64_t e = m < p.members.size() ? p.members[m].at : p.bytes.size();
            crc = (uint32_t)crc32_combine(crc, p.crcs[m], (z_off_t)(e - a));
            member_bytes += e - a;
            if (m < p.members.size()) {
                if (crc != p.members[m].crc || (uint32_t)member_bytes != p.members[m].isize) crc_bad = true;
                crc = (uint32_t)crc32(0L, Z_NULL, 0);
                member_bytes = 0;
            }
            a = e;
        }
        if (out) all.insert(all.end(), p.bytes.begin(), p.bytes.end());
        total += p.bytes.size();
        return 0;
    }, &last);
    if (rc < 0) { last = "not a gzip file the team takes"; if (err_msg) *err_msg = last.c_str(); return KATGPU_ERR_FORMAT; }
    if (!rc && crc_bad) { last = std::string("read error on ") + path; if (err_msg) *err_msg = last.c_str(); return KATGPU_ERR_IO; }
    if (rc) { if (err_msg) *err_msg = last.c_str(); return rc; }
    *n = total;
    if (!out) return KATGPU_OK;
    *out = (uint8_t*)malloc(all.size() ? all.size() : 1);
    if (!*out) return KATGPU_ERR_NOMEM;
    memcpy(*out, all.data(), all.size());
    return KATGPU_OK;
}
