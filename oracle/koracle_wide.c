/* koracle_wide.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path) for k > 32.
 *
 * The same count / hist / gcp / comp semantics as koracle.c, restated for k-mers of up to 64 bases held in one
 * unsigned __int128 (the reference's mer_dna keeps 2k bits in uint64_t[ceil(k/32)], first base most significant:
 * deps/jellyfish-2.2.0/include/jellyfish/mer_dna.hpp:46-63,235-258,330-378).  It accepts every k in 1..64, so for k <= 32 it
 * is checked against koracle.c entry by entry, and for k > 32 against
 *   - the reference's own parser + mer_iterator + mer_dna (oracle/_ref/jf_ref kmers, any k) and
 *   - the naive Python statement of Appendix D (tests/naive.py, strings)
 * in tests/test_oracle_wide.py.  The reducers follow the same reference lines as koracle.c's:
 *   Histogram::binSlice src/histogram.cc:183-199, Gcp::analyseSlice src/gcp.cc:179-197 (+ gcCount str_utils.hpp:151-161),
 *   Comp::compareSlice src/comp.cc:387-484, CompCounters lib/src/comp_counters.cc:91-140.
 * The table is a plain open-addressed array with an occupancy flag per slot (no sentinel key: at k = 64 every 128-bit
 * value is a k-mer).
 */
#define _GNU_SOURCE
#include "koracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

struct ko_wtable {
    unsigned k;
    int canonical;
    uint64_t cap, distinct;      /* cap: power of two */
    u128* keys;
    uint64_t* counts;
    uint8_t* used;
};

static inline u128 wmask(unsigned k) { return k >= 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
static inline u128 mk128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

static inline int wbase_code(uint8_t c) {                      /* mer_dna.hpp:46-63: ACGTacgt -> 0..3, everything else breaks the k-mer */
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static inline uint64_t wmix(u128 x) {
    uint64_t h = (uint64_t)x ^ ((uint64_t)(x >> 64) * 0x9E3779B97F4A7C15ULL);
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
    return h;
}

/* reverse complement, base by base (mer_dna.hpp:330-378 does it word-wise; this is the definition) */
static u128 wrevcomp(u128 x, unsigned k) {
    u128 r = 0;
    for (unsigned i = 0; i < k; i++) { r = (r << 2) | (3 - (unsigned)(x & 3)); x >>= 2; }
    return r;
}
static inline u128 wcanonical(u128 x, unsigned k) { u128 r = wrevcomp(x, k); return r < x ? r : x; }   /* mer_dna.hpp: get_canonical */

static int walloc(ko_wtable* t, uint64_t cap) {
    t->keys = (u128*)malloc(cap * sizeof(u128));
    t->counts = (uint64_t*)calloc(cap, sizeof(uint64_t));
    t->used = (uint8_t*)calloc(cap, 1);
    t->cap = cap;
    return t->keys && t->counts && t->used;
}

ko_wtable* ko_wtable_new(unsigned k, int canonical) {
    if (k < 1 || k > 64) return NULL;
    ko_wtable* t = (ko_wtable*)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->k = k; t->canonical = canonical ? 1 : 0;
    if (!walloc(t, 1024)) { ko_wtable_free(t); return NULL; }
    return t;
}

void ko_wtable_free(ko_wtable* t) {
    if (!t) return;
    free(t->keys); free(t->counts); free(t->used); free(t);
}

unsigned ko_wtable_k(const ko_wtable* t) { return t->k; }
uint64_t ko_wtable_distinct(const ko_wtable* t) { return t->distinct; }
uint64_t ko_wtable_total(const ko_wtable* t) {
    uint64_t s = 0;
    for (uint64_t i = 0; i < t->cap; i++) if (t->used[i]) s += t->counts[i];
    return s;
}

static void wadd(ko_wtable* t, u128 key, uint64_t amount);

static void wgrow(ko_wtable* t) {
    ko_wtable old = *t;
    if (!walloc(t, old.cap * 2)) abort();
    t->distinct = 0;
    for (uint64_t i = 0; i < old.cap; i++) if (old.used[i]) wadd(t, old.keys[i], old.counts[i]);
    free(old.keys); free(old.counts); free(old.used);
}

static void wadd(ko_wtable* t, u128 key, uint64_t amount) {       /* hash_counter::add, hash_counter.hpp:98-130 */
    if ((t->distinct + 1) * 10 > t->cap * 7) wgrow(t);
    uint64_t p = wmix(key) & (t->cap - 1);
    while (t->used[p] && t->keys[p] != key) p = (p + 1) & (t->cap - 1);
    if (!t->used[p]) { t->used[p] = 1; t->keys[p] = key; t->distinct++; }
    t->counts[p] += amount;
}

static uint64_t wget(const ko_wtable* t, u128 key) {              /* get_val_for_key, large_hash_array.hpp:358-376 */
    uint64_t p = wmix(key) & (t->cap - 1);
    while (t->used[p]) {
        if (t->keys[p] == key) return t->counts[p];
        p = (p + 1) & (t->cap - 1);
    }
    return 0;
}

void ko_wtable_add(ko_wtable* t, uint64_t hi, uint64_t lo, uint64_t amount) { wadd(t, mk128(hi, lo) & wmask(t->k), amount); }
uint64_t ko_wtable_get(const ko_wtable* t, uint64_t hi, uint64_t lo) { return wget(t, mk128(hi, lo)); }

/* JellyfishHelper::getCount for a k-mer given as text (lib/src/jellyfish_helper.cc:189-194): mer_dna(string), canonicalised on demand */
uint64_t ko_wtable_get_mer(const ko_wtable* t, const char* mer, int canonical) {
    u128 x = 0;
    for (unsigned i = 0; i < t->k; i++) x = (x << 2) | (u128)(unsigned)wbase_code((uint8_t)mer[i]);
    return wget(t, canonical ? wcanonical(x, t->k) : x);
}

typedef struct { u128 k; uint64_t c; } wkc_t;
static int wkc_cmp(const void* a, const void* b) {
    const wkc_t *x = (const wkc_t*)a, *y = (const wkc_t*)b;
    return x->k < y->k ? -1 : (x->k > y->k ? 1 : 0);
}

void ko_wtable_dump_sorted(const ko_wtable* t, uint64_t* hi, uint64_t* lo, uint64_t* counts) {
    wkc_t* v = (wkc_t*)malloc((t->distinct ? t->distinct : 1) * sizeof(wkc_t));
    uint64_t n = 0;
    for (uint64_t i = 0; i < t->cap; i++) if (t->used[i]) { v[n].k = t->keys[i]; v[n].c = t->counts[i]; n++; }
    qsort(v, n, sizeof(wkc_t), wkc_cmp);
    for (uint64_t i = 0; i < n; i++) { hi[i] = (uint64_t)(v[i].k >> 64); lo[i] = (uint64_t)v[i].k; counts[i] = v[i].c; }
    free(v);
}

/* mer_iterator (mer_iterator.hpp:59-89): roll the forward word and the reverse complement; anything outside ACGTacgt
 * resets the window; emit once k bases are in; canonical = the smaller of the two. */
void ko_wcount_bases(ko_wtable* t, const uint8_t* s, size_t n) {
    const unsigned k = t->k;
    const u128 mask = wmask(k);
    const unsigned rshift = 2 * (k - 1);
    u128 m = 0, rc = 0;
    unsigned filled = 0;
    for (size_t i = 0; i < n; i++) {
        int code = wbase_code(s[i]);
        if (code >= 0) {
            m = ((m << 2) | (u128)(unsigned)code) & mask;                 /* shift_left */
            rc = (rc >> 2) | ((u128)(unsigned)(3 - code) << rshift);      /* shift_right of the complement */
            if (filled < k) filled++;
            if (filled >= k) wadd(t, (t->canonical && rc < m) ? rc : m, 1);
        } else {
            filled = 0;
        }
    }
}

int ko_wcount_files(ko_wtable* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p) {
    for (size_t i = 0; i < n_paths; i++) {                       /* files of a group never join (parser.hpp:151-155) */
        uint8_t* b = NULL; size_t n = 0;
        int rc = ko_parse_file(paths[i], trim5p ? trim5p[i] : 0, &b, &n);
        if (rc) return rc;
        ko_wcount_bases(t, b, n);
        ko_free(b);
    }
    return KO_OK;
}

#define WFOR_EACH(t, KEY, CNT, ...)                                                        \
    do {                                                                                   \
        for (uint64_t _i = 0; _i < (t)->cap; _i++) {                                       \
            if (!(t)->used[_i]) continue;                                                  \
            u128 KEY = (t)->keys[_i]; uint64_t CNT = (t)->counts[_i]; __VA_ARGS__          \
        }                                                                                  \
    } while (0)

void ko_whist(const ko_wtable* t, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t* out, size_t nb) {
    memset(out, 0, nb * sizeof(uint64_t));
    WFOR_EACH(t, key, val, {
        (void)key;
        if (val < base) ++out[0];
        else if (val > ceil_) ++out[nb - 1];
        else ++out[(val - base) / inc];
    });
}

static unsigned wgc(u128 key, unsigned k) {                      /* gcCount: #G + #C (str_utils.hpp:151-161) */
    unsigned g = 0;
    for (unsigned i = 0; i < k; i++) { unsigned c = (unsigned)(key & 3); g += (c == 1 || c == 2); key >>= 2; }
    return g;
}

void ko_wgcp(const ko_wtable* t, double cvg_scale, uint32_t cvg_bins, uint64_t* out) {
    const unsigned k = t->k;
    const size_t cols = (size_t)cvg_bins + 1;
    memset(out, 0, (size_t)k * cols * sizeof(uint64_t));
    WFOR_EACH(t, key, cnt, {
        unsigned g = wgc(key, k);
        uint64_t pos = cnt == 0 ? 0 : (uint64_t)ceil((double)cnt * cvg_scale);
        if (pos > cvg_bins) pos = cvg_bins;
        if (g < k) out[(size_t)g * cols + pos]++;                /* k rows only (src/gcp.cc:93): GC == k is dropped */
    });
}

static inline uint64_t wscale(uint64_t c, double s) { return c == 0 ? 0 : (uint64_t)ceil((double)c * s); }
static inline void wspec(uint64_t* sp, size_t size, uint64_t c) { if (c == 0) ++sp[0]; else if (c >= size) ++sp[size - 1]; else ++sp[c]; }

enum { H1_TOTAL, H2_TOTAL, H3_TOTAL, H1_DISTINCT, H2_DISTINCT, H3_DISTINCT, H1_ONLY_TOTAL, H2_ONLY_TOTAL,
       H1_ONLY_DISTINCT, H2_ONLY_DISTINCT, SH_H1_TOTAL, SH_H2_TOTAL, SH_DISTINCT };

void ko_wcomp(const ko_wtable* t1, const ko_wtable* t2, int canon1, int canon2,
              double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
              uint64_t* mx, uint64_t cc[13], uint64_t* spectra) {
    (void)canon1;
    const unsigned k = t1->k;
    const size_t ss = d1_bins < d2_bins ? d1_bins : d2_bins;
    uint64_t *sp1 = spectra, *sp2 = spectra + ss, *shs1 = spectra + 2 * ss, *shs2 = spectra + 3 * ss;
    memset(mx, 0, (size_t)d1_bins * d2_bins * sizeof(uint64_t));
    memset(cc, 0, 13 * sizeof(uint64_t));
    memset(spectra, 0, 4 * ss * sizeof(uint64_t));
    WFOR_EACH(t1, key, c1, {                                     /* pass 1, src/comp.cc:392-433 */
        uint64_t c2 = wget(t2, canon2 ? wcanonical(key, k) : key);
        cc[H1_TOTAL] += c1; cc[H1_DISTINCT]++; wspec(sp1, ss, c1);
        if (!c2) { cc[H1_ONLY_TOTAL] += c1; cc[H1_ONLY_DISTINCT]++; }
        if (c1 && c2) { cc[SH_H1_TOTAL] += c1; cc[SH_H2_TOTAL] += c2; cc[SH_DISTINCT]++; wspec(shs1, ss, c1); wspec(shs2, ss, c2); }
        uint64_t s1 = wscale(c1, d1_scale), s2 = wscale(c2, d2_scale);
        if (s1 >= d1_bins) s1 = d1_bins - 1;
        if (s2 >= d2_bins) s2 = d2_bins - 1;
        mx[s1 * d2_bins + s2]++;
    });
    WFOR_EACH(t2, key, c2, {                                     /* pass 2, :439-463: the lookup is ALWAYS canonicalised (:447) */
        uint64_t c1 = wget(t1, wcanonical(key, k));
        cc[H2_TOTAL] += c2; cc[H2_DISTINCT]++; wspec(sp2, ss, c2);
        if (!c1) {
            cc[H2_ONLY_TOTAL] += c2; cc[H2_ONLY_DISTINCT]++;
            uint64_t s2 = wscale(c2, d2_scale);
            if (s2 >= d2_bins) s2 = d2_bins - 1;
            mx[s2]++;
        }
    });
}

void ko_wcomp3(const ko_wtable* t1, const ko_wtable* t2, const ko_wtable* t3, int canon1, int canon2, int canon3,
               double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
               uint64_t* mx, uint64_t* ends, uint64_t* middle, uint64_t* mixed, uint64_t cc[13], uint64_t* spectra) {
    const unsigned k = t1->k;
    const size_t cells = (size_t)d1_bins * d2_bins;
    ko_wcomp(t1, t2, canon1, canon2, d1_scale, d2_scale, d1_bins, d2_bins, mx, cc, spectra);
    memset(ends, 0, cells * 8); memset(middle, 0, cells * 8); memset(mixed, 0, cells * 8);
    WFOR_EACH(t1, key, c1, {                                     /* src/comp.cc:403-433 */
        uint64_t c2 = wget(t2, canon2 ? wcanonical(key, k) : key);
        uint64_t c3 = wget(t3, canon3 ? wcanonical(key, k) : key);
        uint64_t s1 = wscale(c1, d1_scale), s2 = wscale(c2, d2_scale), s3 = wscale(c3, d2_scale);
        if (s1 >= d1_bins) s1 = d1_bins - 1;
        if (s2 >= d2_bins) s2 = d2_bins - 1;
        if (s3 >= d2_bins) s3 = d2_bins - 1;
        if (s2 == s3) ends[s1 * d2_bins + s3]++;
        else if (s3 > 0) mixed[s1 * d2_bins + s3]++;
        else middle[s1 * d2_bins + s3]++;
    });
    WFOR_EACH(t3, key, c3, { (void)key; cc[H3_TOTAL] += c3; cc[H3_DISTINCT]++; });
}
