/*
 * koracle_sect.c -- CPU ORACLE for `kat sect`.  TEST INFRASTRUCTURE ONLY, like koracle.c.
 *
 * Restates src/sect.cc of TGAC/KAT 2.4.2: for each record of a sequence file, look every k-window up in a k-mer hash
 * (JellyfishHelper::getCount), derive the per-record statistics, and write the -counts.cvg / -counts.gc / -stats.tsv /
 * -contamination.mx / -(non_)repetitive.fa files byte for byte.  The record reader restates the vendored SeqAn 2.0.0
 * FASTA/FASTQ reader (deps/seqan-library-2.0.0/include/seqan/seq_io/fasta_fastq.h:306-380) for a CharString target.
 *
 * PINNING: the reference holds no golden sect outputs and no sect unit test (tests/ has only the two input files
 * tests/data/sect_test.fa and sect_length_test.fa; tests/test_sect.sh checks exit codes).  This restatement is
 * cross-checked against an independent pure-Python restatement (tests/naive.py) -- end-to-end parity of the text
 * outputs with a reference binary is UNPINNED.
 */
#include "koracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { char* p; size_t n, cap; } buf_t;

static void buf_push(buf_t* b, char c) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 256; b->p = (char*)realloc(b->p, b->cap); }
    b->p[b->n++] = c;
}

static int slurp_gz(const char* path, char** out, size_t* n) {      /* SeqAn opens .gz through zlib; plain files pass through */
    gzFile f = gzopen(path, "rb");
    if (!f) return KO_ERR_IO;
    size_t cap = 1 << 20, len = 0;
    char* buf = (char*)malloc(cap);
    for (;;) {
        if (len == cap) { cap *= 2; buf = (char*)realloc(buf, cap); }
        int r = gzread(f, buf + len, (unsigned)((cap - len) > (1u << 30) ? (1u << 30) : (cap - len)));
        if (r < 0) { gzclose(f); free(buf); return KO_ERR_IO; }
        if (r == 0) break;
        len += (size_t)r;
    }
    gzclose(f);
    *out = buf; *n = len;
    return KO_OK;
}

static int is_newline(char c) { return c == '\n' || c == '\r'; }    /* seqan/stream/tokenization.h:148 */

/* readLine / skipLine: up to the newline, then consume "\r\n", "\r" or "\n" (tokenization.h:408-455) */
static size_t read_line(const char* d, size_t n, size_t i, buf_t* into) {
    while (i < n && !is_newline(d[i])) { if (into) buf_push(into, d[i]); i++; }
    if (i < n && d[i] == '\r') i++;
    if (i < n && d[i] == '\n') i++;
    return i;
}

static int ends_with_ci(const char* s, const char* suffix) {
    size_t ls = strlen(s), lx = strlen(suffix);
    if (lx > ls) return 0;
    for (size_t i = 0; i < lx; i++) {
        char a = s[ls - lx + i], b = suffix[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (a != b) return 0;
    }
    return 1;
}

/* 0 = FASTA, 1 = FASTQ, 2 = Raw (.txt: every line a nameless record), -1 = unknown.  SeqAn decides on the file name alone
 * (fasta_fastq.h:104-143), case-insensitively, after stripping a compression extension; any other name is
 * seqan::UnknownExtensionError (checked against the real library: oracle/_ref/seqan_ref). */
static int guess_format(const char* path, const char* d, size_t n) {
    char base[4096];
    (void)d; (void)n;
    snprintf(base, sizeof base, "%s", path);
    if (ends_with_ci(base, ".gz")) base[strlen(base) - 3] = 0;
    if (ends_with_ci(base, ".fa") || ends_with_ci(base, ".fasta")) return 0;
    if (ends_with_ci(base, ".fq") || ends_with_ci(base, ".fastq")) return 1;
    if (ends_with_ci(base, ".txt")) return 2;
    return -1;
}

typedef struct { buf_t name, seq; } record_t;

/* One record (fasta_fastq.h:310-324 and :342-380, char alphabet: only newlines are dropped from the sequence, and the
 * sequence ends at the next '>' / '+' wherever it stands).  Returns the new position. */
static size_t read_record(const char* d, size_t n, size_t i, int fastq, record_t* r) {
    r->name.n = 0; r->seq.n = 0;
    if (fastq == 2) return read_line(d, n, i, &r->seq);      /* Raw: readUntil(seq, IsNewline); skipLine (fasta_fastq.h:283-292) */
    const char begin = fastq ? '@' : '>';
    while (i < n && d[i] != begin) i++;          /* skipUntil(begin) */
    if (i >= n) return (size_t)-1;               /* skipOne at the end: seqan::UnexpectedEnd */
    i++;
    i = read_line(d, n, i, &r->name);
    const char stop = fastq ? '+' : '>';
    while (i < n && d[i] != stop) { if (!is_newline(d[i])) buf_push(&r->seq, d[i]); i++; }
    if (fastq) {
        if (i >= n) return (size_t)-1;           /* skipOne('+') at the end */
        i++;
        i = read_line(d, n, i, NULL);            /* skipLine: optional second id */
        size_t left = r->seq.n;                  /* CountDownFunctor over the non-newline characters */
        while (i < n && left) { if (!is_newline(d[i])) left--; i++; }
        while (i < n && d[i] != '@') i++;        /* forward to the next '@' */
    }
    return i;
}

static int valid_base(char c) {                  /* lib/include/kat/str_utils.hpp:183-201 */
    switch (c) { case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1; default: return 0; }
}

/* The hash behind a profile: the one-word table (koracle.c) or, for k > 32, the wide one (koracle_wide.c). */
typedef struct { unsigned k; const ko_table* n; const ko_wtable* w; } lookup_t;

/* Per-position coverage of one sequence: src/sect.cc:516-535.  counts / gcs hold n-k+1 entries. */
static void profile_l(const lookup_t* L, int canonical, const char* seq, size_t n, uint64_t* counts, int16_t* gcs) {
    const unsigned k = L->k;
    if (n < k) return;
    char mer[72];
    for (size_t i = 0; i + k <= n; i++) {
        int ok = 1, gc = 0;
        for (unsigned j = 0; j < k; j++) {
            char c = seq[i + j];
            if (!valid_base(c)) { ok = 0; break; }
            if (c == 'G' || c == 'g' || c == 'C' || c == 'c') gc++;
            mer[j] = c;
        }
        if (!ok) { counts[i] = 0; if (gcs) gcs[i] = -1; continue; }
        mer[k] = 0;
        if (L->w) counts[i] = ko_wtable_get_mer(L->w, mer, canonical);
        else {
            uint64_t key;
            ko_encode(mer, k, &key);
            counts[i] = ko_table_get(L->n, canonical ? ko_canonical(key, k) : key);    /* lib/src/jellyfish_helper.cc getCount */
        }
        if (gcs) gcs[i] = (int16_t)gc;
    }
}

void ko_profile(const ko_table* t, int canonical, const char* seq, size_t n, uint64_t* counts, int16_t* gcs) {
    const lookup_t L = {ko_table_k(t), t, NULL};
    profile_l(&L, canonical, seq, n, counts, gcs);
}
void ko_wprofile(const ko_wtable* t, int canonical, const char* seq, size_t n, uint64_t* counts, int16_t* gcs) {
    const lookup_t L = {ko_wtable_k(t), NULL, t};
    profile_l(&L, canonical, seq, n, counts, gcs);
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}

static void put_name(FILE* f, const buf_t* name) { fwrite(name->p, 1, name->n, f); }

/* Sect::printRegions, src/sect.cc:373-424 */
static void print_regions(FILE* out, const record_t* r, const uint64_t* counts, size_t nb, unsigned k, uint32_t min_count, uint32_t max_count) {
    char maxs[32];
    if (max_count > 0) snprintf(maxs, sizeof maxs, "-%u", max_count); else snprintf(maxs, sizeof maxs, "+");
    if (!nb) return;
    uint32_t index = 1, start = 0;
    int in_region = 0;
    buf_t ss = {0, 0, 0};
    for (size_t j = 0; j < nb; j++) {
        uint64_t c = counts[j];
        if (c >= min_count && (c <= max_count || max_count == 0)) {
            if (!in_region) { start = (uint32_t)j; in_region = 1; }
            buf_push(&ss, r->seq.p[j]);
        } else if (in_region) {
            uint32_t end = (uint32_t)(j + k - 1);
            fputc('>', out); put_name(out, &r->name);
            fprintf(out, "___region:%u_length:%u_pos:%u:%u_cov:%u%s\n", index++, end - start - 1, start + 1, end, min_count, maxs);
            fwrite(ss.p, 1, ss.n, out);
            for (size_t q = j + 1; q < end; q++) fputc(r->seq.p[q], out);
            fputc('\n', out);
            in_region = 0;
            ss.n = 0;
        }
    }
    if (in_region) {
        uint32_t end = (uint32_t)(nb + k - 1);
        fputc('>', out); put_name(out, &r->name);
        fprintf(out, "___region:%u_length:%u_pos:%u:%u_cov:%u%s\n", index++, end - start - 1, start + 1, end, min_count, maxs);
        fwrite(ss.p, 1, ss.n, out);
        for (size_t q = nb; q < end; q++) fputc(r->seq.p[q], out);
        fputc('\n', out);
    }
    free(ss.p);
}

/* `kat cold` (Cold::processSeqFile + processSeq + printStatTable, src/cold.cc:126-408, 254-271): every record of the
 * assembly file profiled against the reads hash and the assembly's own hash; one -stats.tsv row per record. */
static int cold_l(const lookup_t* reads, int canon_reads, const lookup_t* assembly, int canon_asm, const char* asm_path, const char* prefix) {
    const unsigned k = reads->k;
    char* data; size_t n;
    int rc = slurp_gz(asm_path, &data, &n);
    if (rc) return rc;
    int fmt = guess_format(asm_path, data, n);
    if (fmt < 0) { free(data); return KO_ERR_FORMAT; }
    char path[4096];
    snprintf(path, sizeof path, "%s-stats.tsv", prefix);
    FILE* f = fopen(path, "w");
    if (!f) { free(data); return KO_ERR_IO; }
    fprintf(f, "seq_name\tread_median_cvg\tread_mean_cvg\tasm_cn\tgc%%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%%_invalid\tnon_zero_kmers\t%%_non_zero\t%%_non_zero_corrected\n");
    record_t r; memset(&r, 0, sizeof r);
    uint64_t *rc_counts = NULL, *as_counts = NULL; int16_t* gcs = NULL; size_t cap = 0;
    size_t pos = 0;
    while (pos < n) {
        pos = read_record(data, n, pos, fmt, &r);
        if (pos == (size_t)-1) { rc = KO_ERR_FORMAT; break; }
        const uint64_t seq_len = r.seq.n;
        const int64_t nb_counts = (int64_t)seq_len - (int64_t)k + 1;
        const size_t nb = nb_counts > 0 ? (size_t)nb_counts : 0;
        uint64_t nb_nonzero = 0, nb_invalid = 0;
        uint32_t median = 0, asm_cn = 0; double mean = 0.0;
        if (nb) {
            if (nb > cap) { cap = nb; rc_counts = realloc(rc_counts, cap * 8); as_counts = realloc(as_counts, cap * 8); gcs = realloc(gcs, cap * 2); }
            profile_l(reads, canon_reads, r.seq.p, r.seq.n, rc_counts, gcs);
            profile_l(assembly, canon_asm, r.seq.p, r.seq.n, as_counts, NULL);
            uint64_t sum = 0;
            for (size_t i = 0; i < nb; i++) { if (gcs[i] < 0) nb_invalid++; else { sum += rc_counts[i]; if (rc_counts[i]) nb_nonzero++; } }
            qsort(rc_counts, nb, 8, cmp_u64);
            qsort(as_counts, nb, 8, cmp_u64);
            median = (uint32_t)(double)rc_counts[nb / 2];
            asm_cn = (uint32_t)(double)as_counts[nb / 2];
            mean = (double)sum / (double)nb_counts;
        }
        const double pct_nonzero = nb_nonzero == 0 || nb_counts <= 0 ? 0.0 : ((double)nb_nonzero / (double)nb_counts) * 100.0;
        const double pct_invalid = nb_invalid == 0 || nb_counts <= 0 ? 0.0 : ((double)nb_invalid / (double)nb_counts) * 100.0;
        const uint64_t not_invalid = (uint64_t)nb_counts - nb_invalid;
        const double pct_nz_corr = nb_nonzero == 0 || not_invalid == 0 ? 0.0 : ((double)nb_nonzero / (double)not_invalid) * 100.0;
        uint64_t gs = 0, cs = 0, ns = 0;
        for (uint64_t i = 0; i < seq_len; i++) {
            char c = r.seq.p[i];
            if (c == 'G' || c == 'g') gs++; else if (c == 'C' || c == 'c') cs++; else if (c == 'N' || c == 'n') ns++;
        }
        volatile double num = (double)(gs + cs), den = (double)(seq_len - ns);
        const double gc_perc = num / den;
        put_name(f, &r.name);
        fprintf(f, "\t%u\t%.5f\t%u\t%.5f\t%u\t%u\t%u\t%.5f\t%u\t%.5f\t%.5f\n", median, mean, asm_cn, gc_perc, (uint32_t)seq_len,
                (uint32_t)((uint32_t)seq_len - k + 1), (uint32_t)nb_invalid, pct_invalid, (uint32_t)nb_nonzero, pct_nonzero, pct_nz_corr);
    }
    fclose(f);
    free(rc_counts); free(as_counts); free(gcs); free(r.name.p); free(r.seq.p); free(data);
    return rc;
}

int ko_cold(const ko_table* reads, int canon_reads, const ko_table* assembly, int canon_asm, const char* asm_path, const char* prefix) {
    const lookup_t a = {ko_table_k(reads), reads, NULL}, b = {ko_table_k(assembly), assembly, NULL};
    return cold_l(&a, canon_reads, &b, canon_asm, asm_path, prefix);
}
int ko_wcold(const ko_wtable* reads, int canon_reads, const ko_wtable* assembly, int canon_asm, const char* asm_path, const char* prefix) {
    const lookup_t a = {ko_wtable_k(reads), NULL, reads}, b = {ko_wtable_k(assembly), NULL, assembly};
    return cold_l(&a, canon_reads, &b, canon_asm, asm_path, prefix);
}

/* `kat sect` end to end (Sect::execute + save, src/sect.cc:86-143).  flags: bit0 no_count_stats, bit1 output_gc_stats,
 * bit2 extract_nr, bit3 extract_r, bit4 cvg_logscale, bit5 also Sect::save() (the contamination matrix). */
static int sect_l(const lookup_t* t, int canonical, const char* seq_path, const char* prefix, uint32_t gc_bins, uint32_t cvg_bins,
                  unsigned flags, uint32_t min_repeat, uint32_t max_repeat) {
    const unsigned k = t->k;
    char* data; size_t n;
    int rc = slurp_gz(seq_path, &data, &n);
    if (rc) return rc;
    int fmt = guess_format(seq_path, data, n);
    if (fmt < 0) { free(data); return KO_ERR_FORMAT; }

    char path[4096];
    FILE *f_cvg = NULL, *f_gc = NULL, *f_nr = NULL, *f_r = NULL, *f_stats = NULL;
#define OPEN(fp, suffix) do { snprintf(path, sizeof path, "%s%s", prefix, suffix); fp = fopen(path, "w"); if (!fp) { free(data); return KO_ERR_IO; } } while (0)
    if (!(flags & 1)) OPEN(f_cvg, "-counts.cvg");
    if (flags & 2) OPEN(f_gc, "-counts.gc");
    if (flags & 4) OPEN(f_nr, "-non_repetitive.fa");
    if (flags & 8) OPEN(f_r, "-repetitive.fa");
    OPEN(f_stats, "-stats.tsv");
    fprintf(f_stats, "seq_name\tmedian\tmean\tgc%%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%%_invalid\tnon_zero_kmers\t%%_non_zero\t%%_non_zero_corrected\n");

    uint64_t* mx = (uint64_t*)calloc((size_t)gc_bins * cvg_bins, 8);     /* ThreadedSparseMatrix(gcBins, cvgBins) */
    record_t r; memset(&r, 0, sizeof r);
    uint64_t *counts = NULL, *sorted = NULL; int16_t* gcs = NULL; size_t cap = 0;

    size_t pos = 0;
    while (pos < n) {                                                    /* while (!atEnd(reader)) readRecords(...) */
        pos = read_record(data, n, pos, fmt, &r);
        if (pos == (size_t)-1) { rc = KO_ERR_FORMAT; break; }
        const uint64_t seq_len = r.seq.n;
        const int64_t nb_counts = (int64_t)seq_len - (int64_t)k + 1;
        uint64_t nb_nonzero = 0, nb_invalid = 0;
        uint32_t median = 0; double mean = 0.0;
        const size_t nb = nb_counts > 0 ? (size_t)nb_counts : 0;
        if (nb) {                                                        /* Sect::processSeq, src/sect.cc:486-546 */
            if (nb > cap) { cap = nb; counts = realloc(counts, cap * 8); sorted = realloc(sorted, cap * 8); gcs = realloc(gcs, cap * 2); }
            profile_l(t, canonical, r.seq.p, r.seq.n, counts, gcs);
            uint64_t sum = 0;
            for (size_t i = 0; i < nb; i++) { if (gcs[i] < 0) nb_invalid++; else { sum += counts[i]; if (counts[i]) nb_nonzero++; } }
            memcpy(sorted, counts, nb * 8);
            qsort(sorted, nb, 8, cmp_u64);
            median = (uint32_t)(double)sorted[nb / 2];                   /* vector<uint32_t> medians, :542 */
            mean = (double)sum / (double)nb_counts;
        }
        const double pct_nonzero = nb_nonzero == 0 || nb_counts <= 0 ? 0.0 : ((double)nb_nonzero / (double)nb_counts) * 100.0;
        const double pct_invalid = nb_invalid == 0 || nb_counts <= 0 ? 0.0 : ((double)nb_invalid / (double)nb_counts) * 100.0;
        const uint64_t not_invalid = (uint64_t)nb_counts - nb_invalid;   /* unsigned: "<= 0" is "== 0" (:558-561) */
        const double pct_nz_corr = nb_nonzero == 0 || not_invalid == 0 ? 0.0 : ((double)nb_nonzero / (double)not_invalid) * 100.0;

        uint64_t gs = 0, cs = 0, ns = 0;                                 /* :565-579 */
        for (uint64_t i = 0; i < seq_len; i++) {
            char c = r.seq.p[i];
            if (c == 'G' || c == 'g') gs++; else if (c == 'C' || c == 'c') cs++; else if (c == 'N' || c == 'n') ns++;
        }
        volatile double num = (double)(gs + cs), den = (double)(seq_len - ns);
        const double gc_perc = num / den;                                /* 0/0 -> the x86 default NaN, printed "-nan" */

        /* :581-592.  average_cvg is never assigned in the reference, so the coverage bin is always 0: without
         * --cvg_logscale compressed_cvg = 0.0 * 0.1; with it log10(0) = -inf and the uint16_t conversion of -inf is the
         * x86 "integer indefinite" 0x80000000 whose low 16 bits are 0.  The same conversion sends a NaN gc_perc to x = 0. */
        const double xd = gc_perc * gc_bins;
        const uint16_t x = xd != xd ? 0 : (uint16_t)xd;
        const uint16_t y = 0;
        if (x < gc_bins && y < cvg_bins) mx[(size_t)x * cvg_bins + y] += seq_len;     /* entries outside are never merged, sparse_matrix.hpp:325-331 */

        if (f_cvg) {                                                     /* Sect::printCounts, :328-346 */
            fputc('>', f_cvg); put_name(f_cvg, &r.name); fputc('\n', f_cvg);
            if (nb) {
                fprintf(f_cvg, "%llu", (unsigned long long)counts[0]);
                for (size_t j = 1; j < nb; j++) fprintf(f_cvg, " %llu", (unsigned long long)counts[j]);
                fputc('\n', f_cvg);
            } else fputs("0\n", f_cvg);
        }
        if (f_gc) {                                                      /* Sect::printGCCounts, :352-371 */
            fputc('>', f_gc); put_name(f_gc, &r.name); fputc('\n', f_gc);
            if (nb) {
                for (size_t j = 0; j < nb; j++)
                    fprintf(f_gc, "%s%.1f", j ? " " : "", gcs[j] == -1 ? -0.1 : ((double)gcs[j] / (double)k) * 100.0);
                fputc('\n', f_gc);
            } else fputs("0.0\n", f_gc);
        }
        if (f_nr) print_regions(f_nr, &r, counts, nb, k, 1, min_repeat);
        if (f_r) print_regions(f_r, &r, counts, nb, k, min_repeat, max_repeat);

        put_name(f_stats, &r.name);                                      /* Sect::printStatTable, :427-445 */
        fprintf(f_stats, "\t%u\t%.5f\t%.5f\t%u\t%u\t%u\t%.5f\t%u\t%.5f\t%.5f\n", median, mean, gc_perc, (uint32_t)seq_len,
                (uint32_t)((uint32_t)seq_len - k + 1), (uint32_t)nb_invalid, pct_invalid, (uint32_t)nb_nonzero, pct_nonzero, pct_nz_corr);
    }

    if (f_cvg) fclose(f_cvg);
    if (f_gc) fclose(f_gc);
    if (f_nr) fclose(f_nr);
    if (f_r) fclose(f_r);
    fclose(f_stats);

    if (rc || !(flags & 32)) goto done;      /* Sect::main never calls Sect::save(): the CLI writes no contamination matrix */
    FILE* f; OPEN(f, "-contamination.mx");                               /* Sect::printContaminationMatrix, :449-463 */
    uint64_t maxv = 0;
    for (size_t i = 0; i < (size_t)gc_bins * cvg_bins; i++) if (mx[i] > maxv) maxv = mx[i];
    fprintf(f, "# Title:Contamination Plot for %s and \"\"\n", seq_path);  /* hashFile is an unset bfs::path: prints "" */
    fprintf(f, "# XLabel:GC%%\n# YLabel:Average K-mer Coverage\n# ZLabel:Base Count per bin\n");
    fprintf(f, "# Columns:%u\n# Rows:%u\n# MaxVal:%llu\n# Transpose:0\n###\n", gc_bins, cvg_bins, (unsigned long long)maxv);
    for (uint32_t i = 0; i < gc_bins; i++) {
        for (uint32_t j = 0; j < cvg_bins; j++) fprintf(f, j ? " %llu" : "%llu", (unsigned long long)mx[(size_t)i * cvg_bins + j]);
        fputc('\n', f);
    }
    fclose(f);
done:
#undef OPEN
    free(mx); free(counts); free(sorted); free(gcs); free(r.name.p); free(r.seq.p); free(data);
    return rc;
}

int ko_sect(const ko_table* t, int canonical, const char* seq_path, const char* prefix, uint32_t gc_bins, uint32_t cvg_bins,
            unsigned flags, uint32_t min_repeat, uint32_t max_repeat) {
    const lookup_t L = {ko_table_k(t), t, NULL};
    return sect_l(&L, canonical, seq_path, prefix, gc_bins, cvg_bins, flags, min_repeat, max_repeat);
}
int ko_wsect(const ko_wtable* t, int canonical, const char* seq_path, const char* prefix, uint32_t gc_bins, uint32_t cvg_bins,
             unsigned flags, uint32_t min_repeat, uint32_t max_repeat) {
    const lookup_t L = {ko_wtable_k(t), NULL, t};
    return sect_l(&L, canonical, seq_path, prefix, gc_bins, cvg_bins, flags, min_repeat, max_repeat);
}
