/*
 * koracle.c -- CPU ORACLE (test infrastructure, see koracle.h for the pinning status).
 *
 * Plain-C restatement of the reference algorithm for:  count k-mers -> hist / gcp / comp reducers ->
 * byte-exact text writers.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference; JF/ = deps/jellyfish-2.2.0/).
 * Restricted to k <= 32 (one 64-bit word, first base in the most significant bits, A=0 C=1 G=2 T=3:
 * JF/include/jellyfish/mer_dna.hpp:46-63,330-353).
 */
#define _GNU_SOURCE
#include "koracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ k-mer helpers ---------------- */

/* JF/include/jellyfish/mer_dna.hpp:46-63 : codes[] -- only ACGTacgt are >= 0 */
static inline int base_code(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static inline uint64_t kmask(unsigned k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }

int ko_encode(const char* s, unsigned k, uint64_t* out) {
    uint64_t v = 0;
    for (unsigned i = 0; i < k; i++) {
        int c = base_code((uint8_t)s[i]);
        if (c < 0) return -1;
        v = (v << 2) | (uint64_t)c;          /* shift_left: mer_dna.hpp:330-353 */
    }
    *out = v;
    return 0;
}

void ko_decode(uint64_t key, unsigned k, char* out) {      /* to_str: mer_dna.hpp:442-446 */
    static const char rev[4] = {'A', 'C', 'G', 'T'};
    for (unsigned i = 0; i < k; i++) out[i] = rev[(key >> (2 * (k - 1 - i))) & 3];
    out[k] = 0;
}

/* word_reverse_complement: mer_dna.hpp:100-108, then aligned down to 2k bits */
uint64_t ko_revcomp(uint64_t w, unsigned k) {
    w = ((w >> 2) & 0x3333333333333333ULL) | ((w & 0x3333333333333333ULL) << 2);
    w = ((w >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((w & 0x0F0F0F0F0F0F0F0FULL) << 4);
    w = ((w >> 8) & 0x00FF00FF00FF00FFULL) | ((w & 0x00FF00FF00FF00FFULL) << 8);
    w = ((w >> 16) & 0x0000FFFF0000FFFFULL) | ((w & 0x0000FFFF0000FFFFULL) << 16);
    w = (w >> 32) | (w << 32);
    w = ~w;
    return k >= 32 ? w : (w >> (64 - 2 * k));
}

/* get_canonical: mer_dna.hpp:436-439 with operator< (235-258) == unsigned compare for one word */
uint64_t ko_canonical(uint64_t key, unsigned k) {
    uint64_t rc = ko_revcomp(key, k);
    return rc < key ? rc : key;
}

/* ------------------------------------------------------------------ table ------------------------ */
/* Stand-in for large_hash::array (JF/include/jellyfish/large_hash_array.hpp:56-931): only the
 * multiset {(key,count)} matters for hist/gcp/comp, so the layout is a plain
 * open-addressed table with 64-bit exact counts. */

#define KO_EMPTY (~0ULL)

struct ko_table {
    unsigned k;
    int canonical;
    uint64_t cap;        /* power of two */
    uint64_t* keys;
    uint64_t* counts;
    uint64_t distinct;   /* slots in use (excluding the all-ones key) */
    uint64_t ones_count; /* count of the key 0xFFFF...F (== KO_EMPTY; only reachable for k == 32) */
};

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static int table_alloc(ko_table* t, uint64_t cap) {
    t->keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
    t->counts = (uint64_t*)calloc(cap, sizeof(uint64_t));
    if (!t->keys || !t->counts) return -1;
    memset(t->keys, 0xFF, cap * sizeof(uint64_t));
    t->cap = cap;
    return 0;
}

ko_table* ko_table_new(unsigned k, int canonical) {
    if (k < 1 || k > 32) return NULL;
    ko_table* t = (ko_table*)calloc(1, sizeof(ko_table));
    t->k = k;
    t->canonical = canonical;
    if (table_alloc(t, 1u << 16)) { free(t); return NULL; }
    return t;
}

void ko_table_free(ko_table* t) {
    if (!t) return;
    free(t->keys); free(t->counts); free(t);
}

unsigned ko_table_k(const ko_table* t) { return t->k; }
uint64_t ko_table_distinct(const ko_table* t) { return t->distinct + (t->ones_count ? 1 : 0); }

uint64_t ko_table_total(const ko_table* t) {
    uint64_t s = t->ones_count;
    for (uint64_t i = 0; i < t->cap; i++) if (t->keys[i] != KO_EMPTY) s += t->counts[i];
    return s;
}

static void table_grow(ko_table* t, uint64_t newcap) {
    uint64_t* ok = t->keys; uint64_t* oc = t->counts; uint64_t ocap = t->cap;
    if (table_alloc(t, newcap)) { fprintf(stderr, "koracle: out of memory\n"); abort(); }
    uint64_t m = newcap - 1;
    for (uint64_t i = 0; i < ocap; i++) {
        if (ok[i] == KO_EMPTY) continue;
        uint64_t p = mix64(ok[i]) & m;
        while (t->keys[p] != KO_EMPTY) p = (p + 1) & m;
        t->keys[p] = ok[i]; t->counts[p] = oc[i];
    }
    free(ok); free(oc);
}

/* ---- thread team helper for the multi-threaded (CPU baseline) paths ---- */
typedef void (*par_fn)(int tid, int nthreads, void* ctx);
typedef struct { par_fn fn; int tid, n; void* ctx; } par_job;
static void* par_tramp(void* a) { par_job* j = (par_job*)a; j->fn(j->tid, j->n, j->ctx); return NULL; }
static void par_for(int threads, par_fn fn, void* ctx) {
    if (threads <= 1) { fn(0, 1, ctx); return; }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    par_job* jobs = (par_job*)malloc(sizeof(par_job) * threads);
    for (int i = 0; i < threads; i++) { jobs[i] = (par_job){fn, i, threads, ctx}; pthread_create(&th[i], NULL, par_tramp, &jobs[i]); }
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
}

/* parallel regrow (what hash_counter::double_size does with its thread team, hash_counter.hpp:204-244) */
typedef struct { uint64_t *ok, *oc, ocap; ko_table* t; } grow_ctx;
static void grow_init(int tid, int n, void* c) {
    grow_ctx* g = (grow_ctx*)c;
    uint64_t lo = g->t->cap / n * tid, hi = tid == n - 1 ? g->t->cap : g->t->cap / n * (tid + 1);
    memset(g->t->keys + lo, 0xFF, (hi - lo) * 8); memset(g->t->counts + lo, 0, (hi - lo) * 8);
}
static void grow_move(int tid, int n, void* c) {
    grow_ctx* g = (grow_ctx*)c;
    uint64_t lo = g->ocap / n * tid, hi = tid == n - 1 ? g->ocap : g->ocap / n * (tid + 1), m = g->t->cap - 1;
    for (uint64_t i = lo; i < hi; i++) {
        if (g->ok[i] == KO_EMPTY) continue;
        uint64_t p = mix64(g->ok[i]) & m;
        for (;;) {
            uint64_t exp = KO_EMPTY;
            if (__atomic_compare_exchange_n(&g->t->keys[p], &exp, g->ok[i], 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { g->t->counts[p] = g->oc[i]; break; }
            p = (p + 1) & m;
        }
    }
}
static void table_grow_mt(ko_table* t, uint64_t newcap, int threads) {
    grow_ctx g = {t->keys, t->counts, t->cap, t};
    t->keys = (uint64_t*)malloc(newcap * 8); t->counts = (uint64_t*)malloc(newcap * 8);
    if (!t->keys || !t->counts) { fprintf(stderr, "koracle: out of memory\n"); abort(); }
    t->cap = newcap;
    par_for(threads, grow_init, &g);
    par_for(threads, grow_move, &g);
    free(g.ok); free(g.oc);
}

static void table_reserve(ko_table* t, uint64_t extra, int threads) {
    uint64_t need = t->distinct + extra;
    uint64_t cap = t->cap;
    while (need * 10 > cap * 6) cap <<= 1;
    if (cap != t->cap) { if (threads > 1) table_grow_mt(t, cap, threads); else table_grow(t, cap); }
}

void ko_table_add(ko_table* t, uint64_t key, uint64_t amount) {   /* hash_counter::add: hash_counter.hpp:98-130 */
    if (key == KO_EMPTY) { t->ones_count += amount; return; }
    if ((t->distinct + 1) * 10 > t->cap * 6) table_grow(t, t->cap << 1);
    uint64_t m = t->cap - 1, p = mix64(key) & m;
    for (;;) {
        if (t->keys[p] == key) { t->counts[p] += amount; return; }
        if (t->keys[p] == KO_EMPTY) { t->keys[p] = key; t->counts[p] = amount; t->distinct++; return; }
        p = (p + 1) & m;
    }
}

uint64_t ko_table_get(const ko_table* t, uint64_t key) {   /* get_val_for_key: large_hash_array.hpp:358-376 */
    if (key == KO_EMPTY) return t->ones_count;
    uint64_t m = t->cap - 1, p = mix64(key) & m;
    for (;;) {
        if (t->keys[p] == key) return t->counts[p];
        if (t->keys[p] == KO_EMPTY) return 0;
        p = (p + 1) & m;
    }
}

typedef struct { uint64_t k, c; } kc_t;
static int kc_cmp(const void* a, const void* b) {
    uint64_t x = ((const kc_t*)a)->k, y = ((const kc_t*)b)->k;
    return x < y ? -1 : x > y;
}

void ko_table_dump_sorted(const ko_table* t, uint64_t* keys, uint64_t* counts) {
    uint64_t n = ko_table_distinct(t), j = 0;
    kc_t* v = (kc_t*)malloc((n ? n : 1) * sizeof(kc_t));
    for (uint64_t i = 0; i < t->cap; i++)
        if (t->keys[i] != KO_EMPTY) { v[j].k = t->keys[i]; v[j].c = t->counts[i]; j++; }
    if (t->ones_count) { v[j].k = KO_EMPTY; v[j].c = t->ones_count; j++; }
    qsort(v, n, sizeof(kc_t), kc_cmp);
    for (uint64_t i = 0; i < n; i++) { keys[i] = v[i].k; counts[i] = v[i].c; }
    free(v);
}

/* ------------------------------------------------------------------ counting --------------------- */

/* mer_iterator::operator++ (JF/include/jellyfish/mer_iterator.hpp:61-89) + countSlice
 * (lib/src/jellyfish_helper.cc:202-211): rolling forward m_ and reverse-complement rcm_, reset on
 * any code < 0, emit min(m_, rcm_) when canonical. */
void ko_count_bases(ko_table* t, const uint8_t* s, size_t n) {
    const unsigned k = t->k;
    const uint64_t mask = kmask(k);
    const unsigned rshift = 2 * (k - 1);
    uint64_t m = 0, rc = 0;
    unsigned filled = 0;
    for (size_t i = 0; i < n; i++) {
        int code = base_code(s[i]);
        if (code >= 0) {
            m = ((m << 2) | (uint64_t)code) & mask;                    /* shift_left */
            rc = (rc >> 2) | ((uint64_t)(3 - code) << rshift);         /* shift_right(complement) */
            if (filled < k) filled++;
            if (filled >= k) ko_table_add(t, (t->canonical && rc < m) ? rc : m, 1);
        } else {
            filled = 0;
        }
    }
}

/* --- multi-threaded counting (CPU baseline): T threads over slices of the stream, lock-free
 * CAS claim + atomic add on a shared table, like countSeqFile's T x countSlice over one
 * large_hash::array (lib/src/jellyfish_helper.cc:235-243, large_hash_array.hpp:513-601,733-744). */
typedef struct { ko_table* t; const uint8_t* s; size_t lo, hi, n; uint64_t new_distinct, ones; } mt_job;

static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    ko_table* t = j->t;
    const unsigned k = t->k;
    const uint64_t mask = kmask(k), cm = t->cap - 1;
    const unsigned rshift = 2 * (k - 1);
    uint64_t m = 0, rc = 0, nd = 0, ones = 0;
    unsigned filled = 0;
    /* windows STARTING in [lo,hi): consume bytes [lo, min(n, hi+k-1)) and emit when the window start >= lo */
    size_t end = j->hi + k - 1; if (end > j->n) end = j->n;
    for (size_t i = j->lo; i < end; i++) {
        int code = base_code(j->s[i]);
        if (code < 0) { filled = 0; continue; }
        m = ((m << 2) | (uint64_t)code) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - code) << rshift);
        if (filled < k) filled++;
        if (filled < k) continue;
        uint64_t key = (t->canonical && rc < m) ? rc : m;
        if (key == KO_EMPTY) { ones++; continue; }
        uint64_t p = mix64(key) & cm;
        for (;;) {
            uint64_t cur = __atomic_load_n(&t->keys[p], __ATOMIC_RELAXED);
            if (cur == KO_EMPTY) {
                uint64_t exp = KO_EMPTY;
                if (__atomic_compare_exchange_n(&t->keys[p], &exp, key, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { nd++; cur = key; }
                else cur = exp;
            }
            if (cur == key) { __atomic_fetch_add(&t->counts[p], 1, __ATOMIC_RELAXED); break; }
            p = (p + 1) & cm;
        }
    }
    j->new_distinct = nd; j->ones = ones;
    return NULL;
}

void ko_count_bases_mt(ko_table* t, const uint8_t* s, size_t n, int threads) {
    if (threads <= 1) { ko_count_bases(t, s, n); return; }
    if (n < t->k) return;
    const size_t nstart = n - t->k + 1;            /* number of window start positions */
    size_t block = (size_t)16 << 20;              /* window starts per thread-team round */
    if (block < ((size_t)threads << 18)) block = (size_t)threads << 18;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * threads);
    for (size_t b0 = 0; b0 < nstart; b0 += block) {
        size_t b1 = b0 + block < nstart ? b0 + block : nstart;
        table_reserve(t, b1 - b0, threads);        /* no growth while threads run */
        size_t per = (b1 - b0 + threads - 1) / threads;
        for (int i = 0; i < threads; i++) {
            size_t lo = b0 + per * i, hi = lo + per; if (lo > b1) lo = b1; if (hi > b1) hi = b1;
            jobs[i] = (mt_job){t, s, lo, hi, n, 0, 0};
            pthread_create(&th[i], NULL, mt_worker, &jobs[i]);
        }
        for (int i = 0; i < threads; i++) {
            pthread_join(th[i], NULL);
            t->distinct += jobs[i].new_distinct; t->ones_count += jobs[i].ones;
        }
    }
    free(th); free(jobs);
}

/* ------------------------------------------------------------------ file parsing ----------------- */

static int slurp(const char* path, uint8_t** out, size_t* n) {   /* every input goes through zlib: JF/include/jellyfish/stream_manager.hpp:133-145 */
    gzFile f = gzopen(path, "rb");
    if (!f) return KO_ERR_IO;
    size_t cap = 1 << 20, len = 0;
    uint8_t* buf = (uint8_t*)malloc(cap);
    for (;;) {
        if (len == cap) { cap <<= 1; buf = (uint8_t*)realloc(buf, cap); }
        int r = gzread(f, buf + len, (unsigned)((cap - len) > (1u << 30) ? (1u << 30) : (cap - len)));
        if (r < 0) { gzclose(f); free(buf); return KO_ERR_IO; }
        if (r == 0) break;
        len += (size_t)r;
    }
    gzclose(f);
    *out = buf; *n = len;
    return KO_OK;
}

typedef struct { uint8_t* p; size_t n, cap; } obuf;
static void ob_put(obuf* o, const uint8_t* s, size_t n) {
    if (o->n + n > o->cap) { while (o->n + n > o->cap) o->cap = o->cap ? o->cap * 2 : 1 << 16; o->p = (uint8_t*)realloc(o->p, o->cap); }
    memcpy(o->p + o->n, s, n); o->n += n;
}

/* a tiny emulation of the istream calls the reference parser makes, over a memory buffer */
typedef struct { const uint8_t* d; size_t n, pos; } mstream;
static inline int ms_peek(const mstream* s) { return s->pos < s->n ? s->d[s->pos] : -1; }
static inline void ms_ignore_line(mstream* s) {            /* ignore_line: parser.hpp:264-266 */
    const uint8_t* nl = (const uint8_t*)memchr(s->d + s->pos, '\n', s->n - s->pos);
    s->pos = nl ? (size_t)(nl - s->d) + 1 : s->n;
}
static inline void ms_skip_newlines(mstream* s) { while (s->pos < s->n && s->d[s->pos] == '\n') s->pos++; }  /* :268-271 */
/* is.get(buf, big): the rest of the line, '\n' not consumed.  Returns chars read. */
static inline size_t ms_get_line(mstream* s, obuf* o) {
    const uint8_t* nl = (const uint8_t*)memchr(s->d + s->pos, '\n', s->n - s->pos);
    size_t e = nl ? (size_t)(nl - s->d) : s->n, len = e - s->pos;
    ob_put(o, s->d + s->pos, len);
    s->pos = e;
    return len;
}

/* read_sequence (parser.hpp:248-262), un-chunked: the 4096-byte buffering and (k-1)-byte seams of the
 * reference (:189-216) only decide WHERE the stream is cut, not which k-mers exist (checked by the
 * reference's own invariant test, JF/unit_tests/test_mer_overlap_sequence_parser.cc:28-187). */
static size_t read_sequence(mstream* s, obuf* o, int stop, unsigned trim5p) {
    size_t nread = 0;
    if (trim5p > 0) {
        ms_skip_newlines(s);
        s->pos = s->pos + trim5p < s->n ? s->pos + trim5p : s->n;      /* is.ignore(trim5p) */
    }
    while (s->pos < s->n && ms_peek(s) != stop) {
        ms_skip_newlines(s);
        if (s->pos >= s->n) break;          /* get() at EOF fails -> stream no longer good */
        nread += ms_get_line(s, o);
        ms_skip_newlines(s);
    }
    return nread;
}

/* skip_quals (parser.hpp:274-289).  Deviation: a final quality line without '\n' is accepted (the
 * reference throws there, and cooperative_pool2.hpp:260 swallows the exception, dropping its last buffer). */
static int skip_quals(mstream* s, size_t read_len) {
    ms_ignore_line(s);
    size_t quals = 0;
    while (s->pos < s->n && quals < read_len) {
        ms_skip_newlines(s);
        size_t want = read_len - quals + 1, got = 0;
        int saw_nl = 0;
        while (got < want && s->pos < s->n) { uint8_t c = s->d[s->pos++]; got++; if (c == '\n') { saw_nl = 1; break; } }
        quals += got;
        if (saw_nl || got == want) ++read_len;      /* "if(is) ++read_len" compensates the consumed newline */
        if (!saw_nl && s->pos >= s->n) break;       /* EOF without newline: tolerated iff the length is exact (checked below) */
    }
    ms_skip_newlines(s);
    if (quals == read_len && (ms_peek(s) == '@' || ms_peek(s) == -1)) return KO_OK;
    return KO_ERR_FASTQ;
}

int ko_parse_file(const char* path, unsigned trim5p, uint8_t** bases, size_t* n) {
    uint8_t* raw; size_t rn;
    int rc = slurp(path, &raw, &rn);
    if (rc) return rc;
    mstream s = {raw, rn, 0};
    obuf o = {NULL, 0, 0};
    static const uint8_t N = 'N';
    int type = ms_peek(&s);                       /* open_next_file: parser.hpp:158-187 */
    if (type == -1) { free(raw); *bases = NULL; *n = 0; return KO_OK; }
    if (type != '>' && type != '@') { free(raw); return KO_ERR_FORMAT; }
    ms_ignore_line(&s);
    if (type == '>') {                            /* read_fasta: parser.hpp:189-216 */
        int newread = 1;
        while (s.pos < s.n) {
            read_sequence(&s, &o, '>', newread ? trim5p : 0);
            if (ms_peek(&s) == '>') { ob_put(&o, &N, 1); ms_ignore_line(&s); newread = 1; }
            else newread = 0;
        }
    } else {                                      /* read_fastq: parser.hpp:218-246 */
        size_t seq_len = 0;
        while (s.pos < s.n) {
            seq_len += read_sequence(&s, &o, '+', seq_len == 0 ? trim5p : 0);
            if (ms_peek(&s) == '+') {
                rc = skip_quals(&s, seq_len + trim5p);
                if (rc) { free(raw); free(o.p); return rc; }
                if (s.pos < s.n) { ob_put(&o, &N, 1); ms_ignore_line(&s); }
                seq_len = 0;
            }
        }
    }
    free(raw);
    *bases = o.p; *n = o.n;
    return KO_OK;
}

void ko_free(void* p) { free(p); }

int ko_count_files(ko_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p) {
    for (size_t i = 0; i < n_paths; i++) {      /* files of a group accumulate into one table, never joined */
        uint8_t* b; size_t n;
        int rc = ko_parse_file(paths[i], trim5p ? trim5p[i] : 0, &b, &n);
        if (rc) return rc;
        ko_count_bases(t, b, n);
        free(b);
    }
    return KO_OK;
}

/* ------------------------------------------------------------------ .jf reader ------------------- */
/* generic_file_header::read (JF/include/jellyfish/generic_file_header.hpp:127-153): 9 decimal digits
 * = JSON length, JSON, padding up to "offset"; records (binary_dumper.hpp:47-51,114-119):
 * ceil(key_len/8) key bytes (little-endian words of mer_dna) + counter_len count bytes (little-endian). */
static long json_int(const char* js, const char* key) {
    char pat[64]; snprintf(pat, sizeof pat, "\"%s\":", key);
    const char* p = strstr(js, pat);
    return p ? strtol(p + strlen(pat), NULL, 10) : -1;
}

int ko_jf_load(const char* path, ko_table** out, uint64_t* n_records, char* header_json, size_t header_cap) {
    FILE* f = fopen(path, "rb");
    if (!f) return KO_ERR_IO;
    char digits[10] = {0};
    if (fread(digits, 1, 9, f) != 9) { fclose(f); return KO_ERR_FORMAT; }
    long jlen = strtol(digits, NULL, 10);
    char* js = (char*)calloc((size_t)jlen + 1, 1);
    if (fread(js, 1, (size_t)jlen, f) != (size_t)jlen) { fclose(f); free(js); return KO_ERR_FORMAT; }
    if (header_json && header_cap) { strncpy(header_json, js, header_cap - 1); header_json[header_cap - 1] = 0; }
    long key_len = json_int(js, "key_len"), clen = json_int(js, "counter_len"), offset = json_int(js, "offset");
    int canonical = strstr(js, "\"canonical\":true") != NULL;
    if (offset < 0) { long al = json_int(js, "alignment"); if (al <= 0) al = 8; offset = 9 + jlen; offset += (al - offset % al) % al; }
    free(js);
    if (key_len <= 0 || key_len > 64 || clen <= 0 || clen > 8) { fclose(f); return KO_ERR_FORMAT; }
    unsigned k = (unsigned)key_len / 2, kb = (unsigned)(key_len + 7) / 8;
    ko_table* t = ko_table_new(k, canonical);
    fseek(f, offset, SEEK_SET);
    uint8_t rec[16]; uint64_t nrec = 0;
    while (fread(rec, 1, kb + (size_t)clen, f) == kb + (size_t)clen) {
        uint64_t key = 0, cnt = 0;
        for (unsigned i = 0; i < kb; i++) key |= (uint64_t)rec[i] << (8 * i);
        for (long i = 0; i < clen; i++) cnt |= (uint64_t)rec[kb + i] << (8 * i);
        ko_table_add(t, key, cnt); nrec++;
    }
    fclose(f);
    *out = t; if (n_records) *n_records = nrec;
    return KO_OK;
}

/* ------------------------------------------------------------------ reducers --------------------- */

#define FOR_EACH_ENTRY(t, KEY, CNT, ...)                                            \
    do {                                                                            \
        for (uint64_t _i = 0; _i < (t)->cap; _i++) {                                \
            if ((t)->keys[_i] == KO_EMPTY) continue;                                \
            uint64_t KEY = (t)->keys[_i], CNT = (t)->counts[_i]; __VA_ARGS__             \
        }                                                                           \
        if ((t)->ones_count) { uint64_t KEY = KO_EMPTY, CNT = (t)->ones_count; __VA_ARGS__ } \
    } while (0)

/* Histogram::binSlice, src/histogram.cc:183-199 */
void ko_hist(const ko_table* t, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t* out, size_t nb) {
    memset(out, 0, nb * sizeof(uint64_t));
    FOR_EACH_ENTRY(t, key, val, {
        (void)key;
        if (val < base) ++out[0];
        else if (val > ceil_) ++out[nb - 1];
        else ++out[(val - base) / inc];
    });
}

/* gcCount on the packed key (lib/include/kat/str_utils.hpp:151-161): C=01, G=10 */
static inline unsigned gc_count(uint64_t key, unsigned k) {
    uint64_t x = (key ^ (key >> 1)) & 0x5555555555555555ULL & kmask(k);
    return (unsigned)__builtin_popcountll(x);
}

/* Gcp::analyseSlice, src/gcp.cc:179-197; matrix has k rows (src/gcp.cc:93) so GC == k is never printed */
void ko_gcp(const ko_table* t, double cvg_scale, uint32_t cvg_bins, uint64_t* out) {
    const unsigned k = t->k;
    const size_t cols = (size_t)cvg_bins + 1;
    memset(out, 0, (size_t)k * cols * sizeof(uint64_t));
    FOR_EACH_ENTRY(t, key, cnt, {
        unsigned g = gc_count(key, k);
        uint64_t pos = cnt == 0 ? 0 : (uint64_t)ceil((double)cnt * cvg_scale);
        if (pos > cvg_bins) pos = cvg_bins;
        if (g < k) out[(size_t)g * cols + pos]++;
    });
}

/* Comp::scaleCounter, src/comp.hpp:303-306 */
static inline uint64_t scale_counter(uint64_t c, double s) { return c == 0 ? 0 : (uint64_t)ceil((double)c * s); }
/* CompCounters::updateSpectrum, lib/src/comp_counters.cc:130-140 */
static inline void update_spectrum(uint64_t* sp, size_t size, uint64_t c) {
    if (c == 0) ++sp[0]; else if (c >= size) ++sp[size - 1]; else ++sp[c];
}

enum { H1_TOTAL, H2_TOTAL, H3_TOTAL, H1_DISTINCT, H2_DISTINCT, H3_DISTINCT, H1_ONLY_TOTAL, H2_ONLY_TOTAL,
       H1_ONLY_DISTINCT, H2_ONLY_DISTINCT, SH_H1_TOTAL, SH_H2_TOTAL, SH_DISTINCT };

/* Comp::compareSlice, src/comp.cc:387-484 (two-input form) */
void ko_comp(const ko_table* t1, const ko_table* t2, int canon1, int canon2,
             double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
             uint64_t* mx, uint64_t cc[13], uint64_t* spectra) {
    (void)canon1;
    const unsigned k = t1->k;
    const size_t ss = d1_bins < d2_bins ? d1_bins : d2_bins;
    uint64_t *sp1 = spectra, *sp2 = spectra + ss, *shs1 = spectra + 2 * ss, *shs2 = spectra + 3 * ss;
    memset(mx, 0, (size_t)d1_bins * d2_bins * sizeof(uint64_t));
    memset(cc, 0, 13 * sizeof(uint64_t));
    memset(spectra, 0, 4 * ss * sizeof(uint64_t));
    /* pass 1 (:392-433): hash-1 key looked up in hash 2, canonicalised iff input 2 is canonical (:401) */
    FOR_EACH_ENTRY(t1, key, c1, {
        uint64_t c2 = ko_table_get(t2, canon2 ? ko_canonical(key, k) : key);
        cc[H1_TOTAL] += c1; cc[H1_DISTINCT]++; update_spectrum(sp1, ss, c1);            /* updateHash1Counters :91-100 */
        if (!c2) { cc[H1_ONLY_TOTAL] += c1; cc[H1_ONLY_DISTINCT]++; }
        if (c1 && c2) {                                                                  /* updateSharedCounters :119-128 */
            cc[SH_H1_TOTAL] += c1; cc[SH_H2_TOTAL] += c2; cc[SH_DISTINCT]++;
            update_spectrum(shs1, ss, c1); update_spectrum(shs2, ss, c2);
        }
        uint64_t s1 = scale_counter(c1, d1_scale), s2 = scale_counter(c2, d2_scale);
        if (s1 >= d1_bins) s1 = d1_bins - 1;
        if (s2 >= d2_bins) s2 = d2_bins - 1;
        mx[s1 * d2_bins + s2]++;
    });
    /* pass 2 (:439-463): hash-2 key looked up in hash 1 ALWAYS canonicalised (:447 passes a pointer as the bool) */
    FOR_EACH_ENTRY(t2, key, c2, {
        uint64_t c1 = ko_table_get(t1, ko_canonical(key, k));
        cc[H2_TOTAL] += c2; cc[H2_DISTINCT]++; update_spectrum(sp2, ss, c2);            /* updateHash2Counters :102-111 */
        if (!c1) {
            cc[H2_ONLY_TOTAL] += c2; cc[H2_ONLY_DISTINCT]++;
            uint64_t s2 = scale_counter(c2, d2_scale);
            if (s2 >= d2_bins) s2 = d2_bins - 1;
            mx[s2]++;                                                                    /* main_matrix[0][s2] */
        }
    });
}

/* Multi-threaded ko_comp for the CPU baseline: T x compareSlice over table slices with private accumulators, merged under
 * a lock -- the structure of Comp::compare / compareSlice / merge (src/comp.cc:366-385,387-484,248-265). */
typedef struct {
    const ko_table *t1, *t2; int canon2; double d1_scale, d2_scale; uint32_t d1_bins, d2_bins;
    uint64_t *mx, *cc, *spectra; pthread_mutex_t mu;
} comp_ctx;

static void comp_slice(int tid, int n, void* c) {
    comp_ctx* x = (comp_ctx*)c;
    const unsigned k = x->t1->k;
    const uint32_t d1 = x->d1_bins, d2 = x->d2_bins;
    const size_t ss = d1 < d2 ? d1 : d2, cells = (size_t)d1 * d2;
    uint64_t* mx = (uint64_t*)calloc(cells, 8);
    uint64_t* sp = (uint64_t*)calloc(4 * ss, 8);
    uint64_t cc[13] = {0};
    uint64_t lo = x->t1->cap / n * tid, hi = tid == n - 1 ? x->t1->cap : x->t1->cap / n * (tid + 1);
    for (uint64_t i = lo; i <= hi; i++) {                       /* i == hi only for the last thread: the all-ones key */
        uint64_t key, c1;
        if (i < hi) { key = x->t1->keys[i]; if (key == KO_EMPTY) continue; c1 = x->t1->counts[i]; }
        else { if (tid != n - 1 || !x->t1->ones_count) break; key = KO_EMPTY; c1 = x->t1->ones_count; }
        uint64_t c2 = ko_table_get(x->t2, x->canon2 ? ko_canonical(key, k) : key);
        cc[H1_TOTAL] += c1; cc[H1_DISTINCT]++; update_spectrum(sp, ss, c1);
        if (!c2) { cc[H1_ONLY_TOTAL] += c1; cc[H1_ONLY_DISTINCT]++; }
        if (c1 && c2) { cc[SH_H1_TOTAL] += c1; cc[SH_H2_TOTAL] += c2; cc[SH_DISTINCT]++; update_spectrum(sp + 2 * ss, ss, c1); update_spectrum(sp + 3 * ss, ss, c2); }
        uint64_t s1 = scale_counter(c1, x->d1_scale), s2 = scale_counter(c2, x->d2_scale);
        if (s1 >= d1) s1 = d1 - 1;
        if (s2 >= d2) s2 = d2 - 1;
        mx[s1 * d2 + s2]++;
    }
    lo = x->t2->cap / n * tid; hi = tid == n - 1 ? x->t2->cap : x->t2->cap / n * (tid + 1);
    for (uint64_t i = lo; i <= hi; i++) {
        uint64_t key, c2;
        if (i < hi) { key = x->t2->keys[i]; if (key == KO_EMPTY) continue; c2 = x->t2->counts[i]; }
        else { if (tid != n - 1 || !x->t2->ones_count) break; key = KO_EMPTY; c2 = x->t2->ones_count; }
        uint64_t c1 = ko_table_get(x->t1, ko_canonical(key, k));
        cc[H2_TOTAL] += c2; cc[H2_DISTINCT]++; update_spectrum(sp + ss, ss, c2);
        if (!c1) {
            cc[H2_ONLY_TOTAL] += c2; cc[H2_ONLY_DISTINCT]++;
            uint64_t s2 = scale_counter(c2, x->d2_scale);
            if (s2 >= d2) s2 = d2 - 1;
            mx[s2]++;
        }
    }
    pthread_mutex_lock(&x->mu);
    for (size_t i = 0; i < cells; i++) x->mx[i] += mx[i];
    for (size_t i = 0; i < 4 * ss; i++) x->spectra[i] += sp[i];
    for (int i = 0; i < 13; i++) x->cc[i] += cc[i];
    pthread_mutex_unlock(&x->mu);
    free(mx); free(sp);
}

void ko_comp_mt(const ko_table* t1, const ko_table* t2, int canon1, int canon2, double d1_scale, double d2_scale,
                uint32_t d1_bins, uint32_t d2_bins, uint64_t* mx, uint64_t cc[13], uint64_t* spectra, int threads) {
    (void)canon1;
    const size_t ss = d1_bins < d2_bins ? d1_bins : d2_bins;
    memset(mx, 0, (size_t)d1_bins * d2_bins * 8); memset(cc, 0, 13 * 8); memset(spectra, 0, 4 * ss * 8);
    comp_ctx x = {t1, t2, canon2, d1_scale, d2_scale, d1_bins, d2_bins, mx, cc, spectra, PTHREAD_MUTEX_INITIALIZER};
    par_for(threads < 1 ? 1 : threads, comp_slice, &x);
}

/* Comp::compareSlice with a third hash (src/comp.cc:403-433,466-479) */
void ko_comp3(const ko_table* t1, const ko_table* t2, const ko_table* t3, int canon1, int canon2, int canon3,
              double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
              uint64_t* mx, uint64_t* ends, uint64_t* middle, uint64_t* mixed, uint64_t cc[13], uint64_t* spectra) {
    const unsigned k = t1->k;
    const size_t cells = (size_t)d1_bins * d2_bins;
    ko_comp(t1, t2, canon1, canon2, d1_scale, d2_scale, d1_bins, d2_bins, mx, cc, spectra);
    memset(ends, 0, cells * sizeof(uint64_t)); memset(middle, 0, cells * sizeof(uint64_t)); memset(mixed, 0, cells * sizeof(uint64_t));
    FOR_EACH_ENTRY(t1, key, c1, {
        uint64_t c2 = ko_table_get(t2, canon2 ? ko_canonical(key, k) : key);
        uint64_t c3 = ko_table_get(t3, canon3 ? ko_canonical(key, k) : key);
        uint64_t s1 = scale_counter(c1, d1_scale), s2 = scale_counter(c2, d2_scale), s3 = scale_counter(c3, d2_scale);
        if (s1 >= d1_bins) s1 = d1_bins - 1;
        if (s2 >= d2_bins) s2 = d2_bins - 1;
        if (s3 >= d2_bins) s3 = d2_bins - 1;
        if (s2 == s3) ends[s1 * d2_bins + s3]++;                 /* :426-427 */
        else if (s3 > 0) mixed[s1 * d2_bins + s3]++;             /* :428-429 */
        else middle[s1 * d2_bins + s3]++;                        /* :430-431 */
    });
    FOR_EACH_ENTRY(t3, key, c3, { (void)key; cc[H3_TOTAL] += c3; cc[H3_DISTINCT]++; });   /* updateHash3Counters :113-117 */
}

/* ------------------------------------------------------------------ distance metrics ------------- */
/* lib/include/kat/distance_metrics.hpp:39-127.  Integer accumulation where the reference uses
 * uint64_t (Minkowski sum, :52-57; the Cosine products s1[i]*s2[i] are uint64 products added to a double, :85). */
double ko_distance(int which, const uint64_t* s1, const uint64_t* s2, size_t n) {
    switch (which) {
    case 0: case 1: {
        int p = which == 0 ? 1 : 2;
        uint64_t sum = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t diff = s1[i] < s2[i] ? s2[i] - s1[i] : s1[i] - s2[i];
            sum = (uint64_t)((double)sum + pow((double)diff, p));   /* "sum += std::pow(diff, p)": uint64 += double (:55) */
        }
        return p == 1 ? (double)sum : pow((double)sum, 1.0 / (double)p);
    }
    case 2: {
        double dot = 0.0, da = 0.0, db = 0.0;
        for (size_t i = 0; i < n; i++) {
            dot += (double)(s1[i] * s2[i]);
            da += pow((double)s1[i], 2); db += pow((double)s2[i], 2);
        }
        return 1.0 - (dot / (sqrt(da) * sqrt(db)));
    }
    case 3: {
        double sum = 0.0;
        for (size_t i = 0; i < n; i++) {
            double diff = (double)s1[i] - (double)s2[i];
            double si = (double)(s1[i] + s2[i]);
            if (si > 0) sum += fabs(diff) / si;
        }
        return sum;
    }
    default: {
        double a = 0.0, b = 0.0;
        for (size_t i = 0; i < n; i++) a += (double)(s1[i] < s2[i] ? s1[i] : s2[i]);
        for (size_t i = 0; i < n; i++) b += (double)(s1[i] > s2[i] ? s1[i] : s2[i]);
        return 1.0 - (a / b);
    }
    }
}

/* ------------------------------------------------------------------ writers ---------------------- */

static int is_pipe(const char* p) { return strncmp(p, "/proc", 5) == 0 || strncmp(p, "/dev", 4) == 0; }   /* jellyfish_helper.cc:258-260 */

/* InputHandler::pathString / fileName, lib/src/input_handler.cc:160-178 */
static void path_string(FILE* f, const char* const* paths, size_t n) {
    for (size_t i = 0; i < n; i++) fprintf(f, "%s%s", i ? " " : "", is_pipe(paths[i]) ? "<pipe>" : paths[i]);
}
static void file_name(FILE* f, const char* const* paths, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const char* b = strrchr(paths[i], '/');
        fprintf(f, "%s%s", i ? " " : "", b ? b + 1 : paths[i]);
    }
}
/* boost::filesystem::path operator<< : quoted, '&' escapes '"' and '&' */
static void quoted_path(FILE* f, const char* p) {
    fputc('"', f);
    for (; *p; p++) { if (*p == '"' || *p == '&') fputc('&', f); fputc(*p, f); }
    fputc('"', f);
}

static uint64_t mx_max(const uint64_t* mx, size_t n) { uint64_t m = 0; for (size_t i = 0; i < n; i++) if (mx[i] > m) m = mx[i]; return m; }
/* SparseMatrix::printMatrix (non-transposed), lib/include/kat/sparse_matrix.hpp:269-277 */
static void print_matrix(FILE* f, const uint64_t* mx, size_t rows, size_t cols) {
    for (size_t i = 0; i < rows; i++) {
        fprintf(f, "%llu", (unsigned long long)mx[i * cols]);
        for (size_t j = 1; j < cols; j++) fprintf(f, " %llu", (unsigned long long)mx[i * cols + j]);
        fputc('\n', f);
    }
}

/* Histogram::print, src/histogram.cc:131-144 */
int ko_write_hist(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                  uint64_t base, uint64_t inc, const uint64_t* data, size_t nb) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    fprintf(f, "# Title:%u-mer spectra for: ", k); file_name(f, paths, n_paths); fputc('\n', f);
    fprintf(f, "# XLabel:%u-mer frequency\n", k);
    fprintf(f, "# YLabel:# distinct %u-mers\n", k);
    fprintf(f, "# Kmer value:%u\n", k);
    fprintf(f, "# Input 1:"); path_string(f, paths, n_paths); fputc('\n', f);
    fprintf(f, "###\n");
    uint64_t col = base;
    for (size_t i = 0; i < nb; i++, col += inc) fprintf(f, "%llu %llu\n", (unsigned long long)col, (unsigned long long)data[i]);
    fclose(f);
    return KO_OK;
}

/* Gcp::printMainMatrix, src/gcp.cc:140-156 */
int ko_write_gcp(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                 uint32_t cvg_bins, const uint64_t* mx) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    size_t cols = (size_t)cvg_bins + 1;
    fprintf(f, "# Title:K-mer coverage vs GC count plot for: "); file_name(f, paths, n_paths); fputc('\n', f);
    fprintf(f, "# XLabel:%u-mer frequency\n", k);
    fprintf(f, "# YLabel:GC count\n");
    fprintf(f, "# ZLabel:# distinct %u-mers\n", k);
    fprintf(f, "# Columns:%zu\n", cols);
    fprintf(f, "# Rows:%u\n", k);
    fprintf(f, "# MaxVal:%llu\n", (unsigned long long)mx_max(mx, (size_t)k * cols));
    fprintf(f, "# Transpose:0\n");
    fprintf(f, "# Kmer value:%u\n", k);
    fprintf(f, "# Input 1:"); path_string(f, paths, n_paths); fputc('\n', f);
    fprintf(f, "###\n");
    print_matrix(f, mx, k, cols);
    fclose(f);
    return KO_OK;
}

/* Comp::printMainMatrix, src/comp.cc:308-326 */
int ko_write_comp_main(const char* out_path, unsigned k,
                       const char* const* paths1, size_t n1, const char* const* paths2, size_t n2,
                       uint32_t d1_bins, uint32_t d2_bins, const uint64_t* mx) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    fprintf(f, "# Title:K-mer comparison plot\n");
    fprintf(f, "# XLabel:%u-mer frequency for: ", k); file_name(f, paths1, n1); fputc('\n', f);
    fprintf(f, "# YLabel:%u-mer frequency for: ", k); file_name(f, paths2, n2); fputc('\n', f);
    fprintf(f, "# ZLabel:# distinct %u-mers\n", k);
    fprintf(f, "# Columns:%u\n", d2_bins);
    fprintf(f, "# Rows:%u\n", d1_bins);
    fprintf(f, "# MaxVal:%llu\n", (unsigned long long)mx_max(mx, (size_t)d1_bins * d2_bins));
    fprintf(f, "# Transpose:1\n");
    fprintf(f, "# Kmer value:%u\n", k);
    fprintf(f, "# Input 1:"); path_string(f, paths1, n1); fputc('\n', f);
    fprintf(f, "# Input 2:"); path_string(f, paths2, n2); fputc('\n', f);
    fprintf(f, "###\n");
    print_matrix(f, mx, d1_bins, d2_bins);
    fclose(f);
    return KO_OK;
}

/* CompCounters::printCounts, lib/src/comp_counters.cc:144-206 (two-input form: hash3_total == 0) */
int ko_write_comp_stats(const char* out_path, const char* hash1_path, const char* hash2_path,
                        const uint64_t c[13], const uint64_t* spectra, uint32_t ss) {
    return ko_write_comp_stats3(out_path, hash1_path, hash2_path, "", c, spectra, ss);
}

int ko_write_comp_stats3(const char* out_path, const char* hash1_path, const char* hash2_path, const char* hash3_path,
                         const uint64_t c[13], const uint64_t* spectra, uint32_t ss) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    static const char* names[5] = {"Manhattan", "Euclidean", "Cosine", "Canberra", "Jaccard"};
    const int h3 = c[H3_TOTAL] > 0;
    fprintf(f, "K-mer statistics for: \n");
    fprintf(f, " - Hash 1: "); quoted_path(f, hash1_path); fputc('\n', f);
    fprintf(f, " - Hash 2: "); quoted_path(f, hash2_path); fputc('\n', f);
    if (h3) { fprintf(f, " - Hash 3: "); quoted_path(f, hash3_path); fputc('\n', f); }
    fprintf(f, "\nTotal K-mers in: \n - Hash 1: %llu\n - Hash 2: %llu\n", (unsigned long long)c[H1_TOTAL], (unsigned long long)c[H2_TOTAL]);
    if (h3) fprintf(f, " - Hash 3: %llu\n", (unsigned long long)c[H3_TOTAL]);
    fprintf(f, "\nDistinct K-mers in:\n - Hash 1: %llu\n - Hash 2: %llu\n", (unsigned long long)c[H1_DISTINCT], (unsigned long long)c[H2_DISTINCT]);
    if (h3) fprintf(f, " - Hash 3: %llu\n", (unsigned long long)c[H3_DISTINCT]);
    fprintf(f, "\nTotal K-mers only found in:\n - Hash 1: %llu\n - Hash 2: %llu\n", (unsigned long long)c[H1_ONLY_TOTAL], (unsigned long long)c[H2_ONLY_TOTAL]);
    fprintf(f, "\nDistinct K-mers only found in:\n - Hash 1: %llu\n - Hash 2: %llu\n\n", (unsigned long long)c[H1_ONLY_DISTINCT], (unsigned long long)c[H2_ONLY_DISTINCT]);
    fprintf(f, "Shared K-mers:\n - Total shared found in hash 1: %llu\n - Total shared found in hash 2: %llu\n - Distinct shared K-mers: %llu\n\n",
            (unsigned long long)c[SH_H1_TOTAL], (unsigned long long)c[SH_H2_TOTAL], (unsigned long long)c[SH_DISTINCT]);
    fprintf(f, "Distance between spectra 1 and 2 (all k-mers):\n");
    for (int i = 0; i < 5; i++) fprintf(f, " - %s distance: %g\n", names[i], ko_distance(i, spectra, spectra + ss, ss));
    fprintf(f, "\nDistance between spectra 1 and 2 (shared k-mers):\n");
    for (int i = 0; i < 5; i++) fprintf(f, " - %s distance: %g\n", names[i], ko_distance(i, spectra + 2 * (size_t)ss, spectra + 3 * (size_t)ss, ss));
    fputc('\n', f);
    fclose(f);
    return KO_OK;
}

/* Comp::printEndsMatrix / printMiddleMatrix / printMixedMatrix, src/comp.cc:330-358 */
int ko_write_comp_extra(const char* out_path, int which, const char* p1, const char* p2, const char* p3,
                        uint32_t d1_bins, uint32_t d2_bins, const uint64_t* mx) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    if (which == 0) {
        fprintf(f, "# Each row represents K-mer frequency for: %s\n", p1);
        fprintf(f, "# Each column represents K-mer frequency for sequence ends: %s\n", p3);
    } else if (which == 1) {
        fprintf(f, "# Each row represents K-mer frequency for: %s\n", p1);
        fprintf(f, "# Each column represents K-mer frequency for sequence middles: %s\n", p2);
    } else {
        fprintf(f, "# Each row represents K-mer frequency for hash file 1: %s\n", p1);
        fprintf(f, "# Each column represents K-mer frequency for mixed: %s and %s\n", p2, p3);
    }
    print_matrix(f, mx, d1_bins, d2_bins);
    fclose(f);
    return KO_OK;
}

/* Comp::printHist, src/comp.cc:235-246 (note: pathString(), not fileName()) */
int ko_write_comp_hist(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                       const uint64_t* sp, uint32_t ss) {
    FILE* f = fopen(out_path, "w");
    if (!f) return KO_ERR_IO;
    fprintf(f, "# Title:%u-mer spectra for: ", k); path_string(f, paths, n_paths); fputc('\n', f);
    fprintf(f, "# XLabel:%u-mer frequency\n", k);
    fprintf(f, "# YLabel:# distinct %u-mers\n", k);
    fprintf(f, "###\n");
    for (uint32_t i = 0; i < ss; i++) fprintf(f, "%u %llu\n", i, (unsigned long long)sp[i]);
    fclose(f);
    return KO_OK;
}
