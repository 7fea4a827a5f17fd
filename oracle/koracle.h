/*
 * koracle.h -- CPU ORACLE for the kat hist / gcp / comp hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference's algorithm (TGAC/KAT 2.4.2 + bundled
 * Jellyfish 2.2.0).  It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg can check (and time) the HIP path against an independent CPU implementation.  Nothing in the
 * product (kat_amd/, include/) may include, link, import or execute anything in oracle/.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - The reference's TOOL DRIVERS are unbuildable in this image: the .cc files under src/, lib/src/input_handler.cc and jellyfish_helper.cc
 *     include the autoconf-generated <config.h> unconditionally (e.g. lib/src/input_handler.cc:19), as do Jellyfish's
 *     lib/allocators_mmap.cc, misc.cc and storage.cc (so its hash_counter / large_hash_array cannot be driven either), and
 *     autotools are absent.  No `kat` binary exists here.
 *   - The HEART of the semantics does build, from the sources where they lie and with nothing stubbed (oracle/Makefile `ref`,
 *     drivers in oracle/ref/): Jellyfish 2.2.0's parser + mer_iterator + mer_dna + file_header / binary_reader
 *     (oracle/_ref/jf_ref) and KAT's CompCounters, distance metrics, SparseMatrix, str_utils (oracle/_ref/kat_ref_parts).
 *     tests/test_oracle_vs_reference.py checks this oracle against that real code: the k-mer multiset of every input (the
 *     reference's test data, generated messy FASTA/FASTQ/gzip, edge files), k-mer arithmetic for k = 1..32, the .jf fixture
 *     and .jf files written by the product as the reference's reader sees them, CompCounters::printCounts with its five
 *     distance metrics byte for byte, the counter arithmetic of Comp::compareSlice, SparseMatrix's bounds behaviour, and
 *     validKmer / gcCount.  tests/golden/reference_vectors.json keeps the digests for the fixed inputs.
 *     koracle_wide.c (k up to 64) is checked the same way against jf_ref's multi-word mer_dna for k = 33..64, and against
 *     koracle.c for k <= 32 (tests/test_oracle_wide.py).
 *   - Also pinned against the known answers the reference's own tests hold for this path:
 *       tests/check_jellyfish.cc:38-116  (.jf header fields, 1889 records, k-mer lookups 3/1/1/1 and
 *                                         canonical lookups 3/1/0/0 on tests/data/ecoli.header.jf27)
 *       tests/check_compcounters.cc:30-62 (CompCounters arithmetic: distinct 4, total 60)
 *       tests/data/kat.hist, scripts/test/resources/{hist1.hist,gcp1.mx,spectracn1.mx} (file formats)
 *   - What remains a restatement only (the reference holds NO golden hist/.mx/.stats outputs; its CLI tests check exit codes):
 *     the bodies of Histogram::bin, Gcp::analyse, Comp::compare (binning, scaling, which matrix a k-mer lands in), the text
 *     headers of the .hist / .mx files, and `kat sect` / `kat cold` end to end.  End-to-end numbers quoted in SURVEY.md 8(c)
 *     (recorded by the survey stage from a hand-built reference binary) serve as additional known answers in
 *     tests/test_oracle_known_answers.py, with that provenance stated there.
 */
#ifndef KORACLE_H
#define KORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ko_table ko_table;

enum { KO_OK = 0, KO_ERR_IO = 1, KO_ERR_FORMAT = 2, KO_ERR_FASTQ = 3, KO_ERR_K = 4, KO_ERR_MEM = 5 };

/* --- k-mer helpers (mer_dna.hpp:46-63,100-108,235-258,436-446) --- */
int      ko_encode(const char* s, unsigned k, uint64_t* out);   /* 0 ok, -1 if a non-ACGT char */
void     ko_decode(uint64_t key, unsigned k, char* out /* k+1 */);
uint64_t ko_revcomp(uint64_t key, unsigned k);
uint64_t ko_canonical(uint64_t key, unsigned k);

/* --- table --- */
ko_table* ko_table_new(unsigned k, int canonical);
void      ko_table_free(ko_table*);
unsigned  ko_table_k(const ko_table*);
uint64_t  ko_table_distinct(const ko_table*);
uint64_t  ko_table_total(const ko_table*);
uint64_t  ko_table_get(const ko_table*, uint64_t key);                 /* JellyfishHelper::getCount, canonical=false */
void      ko_table_add(ko_table*, uint64_t key, uint64_t amount);
/* sorted (ascending key) dump; arrays must hold ko_table_distinct() entries */
void      ko_table_dump_sorted(const ko_table*, uint64_t* keys, uint64_t* counts);

/* count every k-window of every maximal ACGTacgt run of a byte stream (mer_iterator.hpp:61-89) */
void      ko_count_bases(ko_table*, const uint8_t* bases, size_t n);
/* multi-threaded variant used for the CPU baseline timing (same result) */
void      ko_count_bases_mt(ko_table*, const uint8_t* bases, size_t n, int threads);
/* parse one FASTA/FASTQ(.gz) file into the 'N'-joined base stream (mer_overlap_sequence_parser.hpp:132-289) */
int       ko_parse_file(const char* path, unsigned trim5p, uint8_t** bases, size_t* n);
void      ko_free(void*);
/* InputHandler::count over a group of files (lib/src/input_handler.cc:180-202) */
int       ko_count_files(ko_table*, const char* const* paths, size_t n_paths, const uint16_t* trim5p);
/* .jf (binary/sorted) reader: binary_dumper.hpp:94-119, generic_file_header.hpp:96-153 */
int       ko_jf_load(const char* path, ko_table** out, uint64_t* n_records, char* header_json, size_t header_cap);

/* --- reducers --- */
/* Histogram::binSlice (src/histogram.cc:183-199) */
void ko_hist(const ko_table*, uint64_t base, uint64_t ceil, uint64_t inc, uint64_t* out, size_t nb);
/* Gcp::analyseSlice (src/gcp.cc:179-197); out is k rows x (cvg_bins+1), GC==k dropped (gcp.cc:93) */
void ko_gcp(const ko_table*, double cvg_scale, uint32_t cvg_bins, uint64_t* out);
/* Comp::compareSlice (src/comp.cc:387-484) + CompCounters (lib/src/comp_counters.cc:91-140).
 * counters[13] order: hash1_total, hash2_total, hash3_total, hash1_distinct, hash2_distinct, hash3_distinct,
 * hash1_only_total, hash2_only_total, hash1_only_distinct, hash2_only_distinct,
 * shared_hash1_total, shared_hash2_total, shared_distinct.
 * spectra: 4 x min(d1_bins,d2_bins): spectrum1, spectrum2, shared_spectrum1, shared_spectrum2. */
void ko_comp(const ko_table* t1, const ko_table* t2, int canon1, int canon2,
             double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
             uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra);

/* multi-threaded form for the CPU baseline (T x compareSlice + merge, src/comp.cc:366-385,248-265); same result */
void ko_comp_mt(const ko_table* t1, const ko_table* t2, int canon1, int canon2,
                double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
                uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra, int threads);

/* three-input form (src/comp.cc:123-127,403-433,466-479): adds ends / middle / mixed matrices (each d1_bins x d2_bins)
 * and counters[2] = hash3_total, counters[5] = hash3_distinct */
void ko_comp3(const ko_table* t1, const ko_table* t2, const ko_table* t3, int canon1, int canon2, int canon3,
              double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
              uint64_t* main_mx, uint64_t* ends_mx, uint64_t* middle_mx, uint64_t* mixed_mx,
              uint64_t counters[13], uint64_t* spectra);

/* --- k > 32 (koracle_wide.c): the same semantics for k-mers of 1..64 bases, one k-mer = (hi, lo) = the 2k-bit word's upper and
 * lower 64 bits.  Checked against koracle.c for k <= 32 and against oracle/_ref/jf_ref (the reference's parser + mer_dna)
 * and tests/naive.py for k > 32 (tests/test_oracle_wide.py). --- */
typedef struct ko_wtable ko_wtable;
ko_wtable* ko_wtable_new(unsigned k, int canonical);
void      ko_wtable_free(ko_wtable*);
unsigned  ko_wtable_k(const ko_wtable*);
uint64_t  ko_wtable_distinct(const ko_wtable*);
uint64_t  ko_wtable_total(const ko_wtable*);
void      ko_wtable_add(ko_wtable*, uint64_t hi, uint64_t lo, uint64_t amount);
uint64_t  ko_wtable_get(const ko_wtable*, uint64_t hi, uint64_t lo);
uint64_t  ko_wtable_get_mer(const ko_wtable*, const char* mer /* k valid bases */, int canonical);
void      ko_wtable_dump_sorted(const ko_wtable*, uint64_t* hi, uint64_t* lo, uint64_t* counts);   /* ko_wtable_distinct entries, by key */
void      ko_wcount_bases(ko_wtable*, const uint8_t* bases, size_t n);
int       ko_wcount_files(ko_wtable*, const char* const* paths, size_t n_paths, const uint16_t* trim5p);
void      ko_whist(const ko_wtable*, uint64_t base, uint64_t ceil_, uint64_t inc, uint64_t* out, size_t nb);
void      ko_wgcp(const ko_wtable*, double cvg_scale, uint32_t cvg_bins, uint64_t* out);
void      ko_wcomp(const ko_wtable* t1, const ko_wtable* t2, int canon1, int canon2, double d1_scale, double d2_scale,
                   uint32_t d1_bins, uint32_t d2_bins, uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra);
void      ko_wcomp3(const ko_wtable* t1, const ko_wtable* t2, const ko_wtable* t3, int canon1, int canon2, int canon3,
                    double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
                    uint64_t* main_mx, uint64_t* ends_mx, uint64_t* middle_mx, uint64_t* mixed_mx, uint64_t counters[13], uint64_t* spectra);

/* --- writers (byte-exact text) --- */
int ko_write_hist(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                  uint64_t base, uint64_t inc, const uint64_t* data, size_t nb);
int ko_write_gcp(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                 uint32_t cvg_bins, const uint64_t* mx);
int ko_write_comp_main(const char* out_path, unsigned k,
                       const char* const* paths1, size_t n1, const char* const* paths2, size_t n2,
                       uint32_t d1_bins, uint32_t d2_bins, const uint64_t* mx);
int ko_write_comp_stats(const char* out_path, const char* hash1_path, const char* hash2_path,
                        const uint64_t counters[13], const uint64_t* spectra, uint32_t spec_size);
/* same with the " - Hash 3:" lines (printed iff hash3_total > 0, lib/src/comp_counters.cc:150-151,160-161,169-170) */
int ko_write_comp_stats3(const char* out_path, const char* hash1_path, const char* hash2_path, const char* hash3_path,
                         const uint64_t counters[13], const uint64_t* spectra, uint32_t spec_size);
/* Comp::printEndsMatrix / printMiddleMatrix / printMixedMatrix (src/comp.cc:330-358); which: 0 ends, 1 middle, 2 mixed */
int ko_write_comp_extra(const char* out_path, int which, const char* path1, const char* path2, const char* path3,
                        uint32_t d1_bins, uint32_t d2_bins, const uint64_t* mx);
int ko_write_comp_hist(const char* out_path, unsigned k, const char* const* paths, size_t n_paths,
                       const uint64_t* spectrum, uint32_t spec_size);
/* distance metrics (lib/include/kat/distance_metrics.hpp:39-127): 0 Manhattan 1 Euclidean 2 Cosine 3 Canberra 4 Jaccard */
double ko_distance(int which, const uint64_t* s1, const uint64_t* s2, size_t n);

/* ---- `kat sect` (koracle_sect.c; src/sect.cc) ---- */
/* per-position coverage of one sequence: counts[i] (and gcs[i], may be NULL; -1 = invalid window) for i in [0, n-k] */
void ko_profile(const ko_table* t, int canonical, const char* seq, size_t n, uint64_t* counts, int16_t* gcs);
/* the whole tool; flags: 1 no_count_stats, 2 output_gc_stats, 4 extract_nr, 8 extract_r, 16 cvg_logscale, 32 save() */
/* `kat cold` (src/cold.cc): <prefix>-stats.tsv for every record of the assembly file */
int ko_cold(const ko_table* reads, int canon_reads, const ko_table* assembly, int canon_asm, const char* asm_path, const char* prefix);
int ko_sect(const ko_table* t, int canonical, const char* seq_path, const char* prefix, uint32_t gc_bins, uint32_t cvg_bins,
            unsigned flags, uint32_t min_repeat, uint32_t max_repeat);
/* the same three on a wide table (k > 32, koracle_wide.c) */
void ko_wprofile(const ko_wtable* t, int canonical, const char* seq, size_t n, uint64_t* counts, int16_t* gcs);
int ko_wcold(const ko_wtable* reads, int canon_reads, const ko_wtable* assembly, int canon_asm, const char* asm_path, const char* prefix);
int ko_wsect(const ko_wtable* t, int canonical, const char* seq_path, const char* prefix, uint32_t gc_bins, uint32_t cvg_bins,
             unsigned flags, uint32_t min_repeat, uint32_t max_repeat);

#ifdef __cplusplus
}
#endif
#endif
