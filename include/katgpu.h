/*
 * katgpu.h -- C ABI of libkatgpu.so: the MI355X (gfx950) k-mer counting / spectra-reduction engine that
 * drops in behind KAT's `hist` / `gcp` / `comp` hot path.
 *
 * KAT (TGAC/KAT 2.4.2) has no plugin/FFI layer: the seam is a handful of C++ call sites.  Each entry point
 * below names the reference routine it replaces (paths relative to the KAT source tree; JF/ =
 * deps/jellyfish-2.2.0/).  INTEGRATION.md shows the patch a KAT maintainer would apply.
 *
 * Conventions
 *   - every function returns a katgpu_status (0 = ok); katgpu_last_error() gives the message the reference
 *     would have thrown (same wording where the reference has one);
 *   - handles are opaque; result buffers are caller-allocated HOST memory unless a parameter is named dev_*;
 *   - one caller thread per ctx; a table is immutable once counting has finished, so reducers may be
 *     called in any order, any number of times;
 *   - k-mers are 2-bit packed, first base in the most significant bits, A=0 C=1 G=2 T=3
 *     (JF/include/jellyfish/mer_dna.hpp:46-63,330-353).  1 <= k <= KATGPU_MAX_K = 63: one 64-bit word up to k = 32,
 *     two 63-bit words beyond ("wide" tables); k >= 64 gives KATGPU_ERR_K.
 *   - there is NO CPU fallback: without a gfx950 device katgpu_init fails with KATGPU_ERR_DEVICE.
 */
#ifndef KATGPU_H
#define KATGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct katgpu_ctx katgpu_ctx;
typedef struct katgpu_table katgpu_table;

/* k-mer lengths.  1..32: one 64-bit word per k-mer (all entry points).  33..KATGPU_MAX_K: "wide" tables, the k-mer's 2k bits
 * in two words -- counted and reduced (count*, stats, hist, gcp, comp, comp3) exactly like the narrow ones; records move
 * through the *_wide entry points as (hi, lo) = the upper and lower 64 bits of the 2k-bit word (first base most significant,
 * A=0 C=1 G=2 T=3, as mer_dna: JF/include/jellyfish/mer_dna.hpp:235-258); .jf files and the sect/cold profile work for both.
 * Entry points that take 64-bit keys and the region-ordered extraction calls return KATGPU_ERR_K for a wide table (its exchange --
 * katgpu_exchange_merge does it -- is katgpu_table_partition_sizes + katgpu_table_partition_wide + katgpu_table_merge_device_wide). */
#define KATGPU_MAX_K 63

typedef enum katgpu_status {
    KATGPU_OK = 0,
    KATGPU_ERR_INVALID_ARG = 1,
    KATGPU_ERR_IO = 2,          /* "Could not find input file at: ..."   lib/src/input_handler.cc:119-122 */
    KATGPU_ERR_FORMAT = 3,      /* "Unsupported format"                  JF/include/jellyfish/mer_overlap_sequence_parser.hpp:184 */
    KATGPU_ERR_FASTQ = 4,       /* "Invalid fastq sequence"              mer_overlap_sequence_parser.hpp:288 */
    KATGPU_ERR_NOMEM = 5,       /* device allocation failed */
    KATGPU_ERR_K = 6,           /* k outside 1..KATGPU_MAX_K (reference: any k, JF/include/jellyfish/mer_dna.hpp:725), or an
                                 * entry point that handles one-word k-mers only called on a k > 32 table */
    KATGPU_ERR_TABLE_FULL = 7,  /* "Hash full"                           JF/include/jellyfish/hash_counter.hpp:198-199 */
    KATGPU_ERR_DEVICE = 8,      /* HIP runtime error / no gfx950 device */
    KATGPU_ERR_MISMATCH = 9     /* tables with different k   lib/src/input_handler.cc:145-158 (validateMerLen) */
} katgpu_status;

/* ---- context ---------------------------------------------------------------------------------------- */
/* device: HIP ordinal, or -1 for the current device. */
int         katgpu_init(int device, katgpu_ctx** ctx);
/* HIP devices this process sees (the runtime's count: *_VISIBLE_DEVICES honoured); 0 when there is none or no runtime. */
int         katgpu_device_count(void);
void        katgpu_shutdown(katgpu_ctx* ctx);
const char* katgpu_last_error(const katgpu_ctx* ctx);
const char* katgpu_version(void);
/* wait for everything the ctx has queued on its HIP streams */
int         katgpu_sync(katgpu_ctx* ctx);
/* return cached device memory (parked table arrays, the partitioned counter's arena) to the driver; it is re-acquired on demand */
int         katgpu_release_scratch(katgpu_ctx* ctx);
/* borrow that arena as raw device scratch of at least `bytes` (valid until the next katgpu_count_* call on the context) */
int         katgpu_scratch_acquire(katgpu_ctx* ctx, size_t bytes, void** dev_ptr, size_t* got_bytes);

/* ---- counting: replaces InputHandler::count (lib/src/input_handler.cc:180-202) and everything below it:
 *      JellyfishHelper::countSeqFile/countSlice (lib/src/jellyfish_helper.cc:219-246,202-211),
 *      mer_overlap_sequence_parser (JF/.../mer_overlap_sequence_parser.hpp:132-289), mer_iterator
 *      (JF/.../mer_iterator.hpp:61-89), hash_counter::add (JF/.../hash_counter.hpp:98-130),
 *      large_hash::array::add/claim_key/add_val (JF/.../large_hash_array.hpp:298-302,513-601,733-744). ---- */

/* One input group -> one table.  paths: FASTA/FASTQ, plain or gzip, told apart by their first byte.
 * trim5p: per-file count of leading bases to ignore in every record, or NULL.
 * size_hint: initial number of table slots (KAT's -H); 0 = size from the input.
 * disable_grow: KAT's -g; when set a full table is KATGPU_ERR_TABLE_FULL instead of a regrow
 * (hash_counter::double_size, JF/.../hash_counter.hpp:204-244). */
int katgpu_count(katgpu_ctx* ctx, const char* const* paths, size_t n_paths, uint32_t k, int canonical,
                 const uint16_t* trim5p, uint64_t size_hint, int disable_grow, katgpu_table** out);

/* The same in steps (used by the batch/bench drivers and by multi-GPU sharding). */
int katgpu_table_create(katgpu_ctx* ctx, uint32_t k, int canonical, uint64_t size_hint, int disable_grow,
                        katgpu_table** out);
/* As katgpu_table_create, but with the region grid of `like` (the table this one will be compared with): katgpu_comp then
 * joins the two region against region in LDS instead of probing HBM.  Falls back to an own grid if the sizes are too far apart,
 * and for k > 32 (wide tables are compared by probes). */
int katgpu_table_create_like(katgpu_ctx* ctx, const katgpu_table* like, uint32_t k, int canonical, uint64_t size_hint,
                             int disable_grow, katgpu_table** out);
/* Files and host buffers take the same counter as device-resident input: the parsed stream goes through pinned staging into
 * two device rings (KATGPU_RING_MB, default 1024 each); a full ring is counted (partition rounds for anything of size) on a
 * worker thread while the parser fills the other. */
int katgpu_count_files(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p);
/* The share of rank `rank` of `world` processes (one per GPU) in counting the group: every rank calls it with the same file list into
 * a table of its own, then katgpu_exchange_merge makes the tables one (below).  Plain FASTQ files of size are cut between the ranks
 * batch by batch, at record starts each rank finds for itself; every other file goes whole to one rank. */
int katgpu_count_files_sharded(katgpu_table* t, const char* const* paths, size_t n_paths, const uint16_t* trim5p, int rank, int world);
/* A base stream is what the reference's parser hands to mer_iterator: sequence bytes, records separated by any
 * byte outside ACGTacgt (the reference inserts 'N', mer_overlap_sequence_parser.hpp:202,234).  Every k-window
 * of every maximal ACGTacgt run is counted once. */
int katgpu_count_bases_host(katgpu_table* t, const uint8_t* bases, size_t n);
int katgpu_count_bases_device(katgpu_table* t, const uint8_t* dev_bases, size_t n);
void katgpu_table_free(katgpu_table* t);

/* Host-only: run the ingest parser alone and return the base stream of one file (malloc'd; release with
 * katgpu_free_host).  Needs no device; *err_msg (optional, static storage, valid until the next call on this thread)
 * receives the message on failure.  This is what katgpu_count_files streams to the GPU. */
int  katgpu_parse_file(const char* path, uint32_t trim5p, uint8_t** bases, size_t* n, const char** err_msg);
/* The same for one input group (InputHandler::count's file list, lib/src/input_handler.cc:180-202): the stream
 * katgpu_count_files feeds to the counter for these paths.  Files that stream (gzip, 5' trim, small) are read
 * concurrently, so the stream interleaves their blocks -- with 'N' + the file's previous k-1 bytes at every switch
 * of source, which keeps the k-mer multiset equal to that of the files read one after the other.  The byte ORDER
 * may differ from call to call; KATGPU_INGEST_FILES=1 reads the files in sequence. */
int  katgpu_parse_files(const char* const* paths, size_t n_paths, const uint16_t* trim5p, uint32_t k,
                        uint8_t** bases, size_t* n, const char** err_msg);
void katgpu_free_host(void* p);
/* Host-only: what the reader threads of the large-FASTQ ingest do to a record-aligned piece of a plain four-line FASTQ file before it
 * crosses PCIe (kg_ingest.hpp: strip_fastq_records) -- each record's sequence line followed by 'N'.  out holds n / 2 + 1 bytes.
 * Returns KATGPU_OK, or KATGPU_ERR_FASTQ when the bytes are not whole plain four-line records (such pieces go through the host state
 * machine, katgpu_parse_file's, instead: it alone knows what multi-line records and odd quality lengths mean). */
int  katgpu_strip_fastq(const uint8_t* fastq, size_t n, uint8_t* out, size_t* out_n);
/* The inflated bytes of ONE ordinary gzip stream through the thread team that katgpu_count_files / katgpu_parse_file use for .gz
 * files of size (kg_pgzip.cpp: the file cut into chunks, each entered at a deflate block start found by search and decoded without
 * its 32 KiB of history, stitched and resolved in order; every member's CRC-32 and ISIZE checked) -- for tests and tools: the
 * reference reads every input through one zlib stream (deps/jellyfish-2.2.0/include/jellyfish/stream_manager.hpp:41-51,133-145).
 * *out is malloc'ed (katgpu_free_host).  KATGPU_PGZ_THREADS / KATGPU_PGZ_CHUNK (bytes of compressed input per chunk) size the team.
 * KATGPU_ERR_FORMAT: not a gzip file; KATGPU_ERR_IO ("read error on <path>"): corrupt or truncated. */
int  katgpu_inflate_file(const char* path, uint8_t** out, size_t* n, const char** err_msg);

/* The placement hash of one-word tables (kg_device.hpp "placement"; the counterpart of the invertible hash + remainder storage of
 * JF/include/jellyfish/large_hash_array.hpp:169-171), on the host, for a table of p1 x 2^l2 regions: per key the two region digits,
 * the remainder a partition item carries, the key the inverse gives back and (offset != NULL) the home slot inside a region of
 * region_slots slots.  *rem_bits = bits of a remainder.  No GPU needed; the parity tests use it to check that the hash is one to
 * one, that its inverse is its inverse and that it spreads k-mers evenly. */
int katgpu_place_keys(uint32_t k, uint32_t p1, uint32_t l2, const uint64_t* keys, size_t n, uint32_t* d1, uint32_t* d2, uint64_t* rem,
                      uint64_t* back, uint32_t* rem_bits, uint32_t region_slots, uint32_t* offset);

/* A hint: a table of about size_hint slots for k-mers of length k will be asked for soon (katgpu_count / katgpu_table_create*).  Its
 * memory is allocated now, on a thread of the library's, beside whatever the caller does next -- `kat comp` calls it for its second
 * input before it counts the first (InputHandler::count of input 2 follows input 1's, src/comp.cc:139-143).  Returns at once. */
int katgpu_reserve(katgpu_ctx* ctx, uint32_t k, uint64_t size_hint);

/* distinct k-mers, sum of counts, slots allocated */
int katgpu_table_stats(katgpu_table* t, uint64_t* distinct, uint64_t* total, uint64_t* capacity);
uint32_t katgpu_table_k(const katgpu_table* t);
int      katgpu_table_canonical(const katgpu_table* t);
/* How many times the table has grown so far (hash_counter::double_size, JF/include/jellyfish/hash_counter.hpp:204-244, which
 * prints "Warning: Specified hash size insufficent - attempting to double hash size... success!" each time): the size hint
 * (KAT's -H) was too small.  A growth step here may more than double. */
uint32_t katgpu_table_regrows(const katgpu_table* t);
/* HBM bytes per slot: 8 = packed (one word: the placement hash's remainder | count -- the quotienting of
 * JF/include/jellyfish/large_hash_array.hpp:169-171; every k <= 32 table of size), 12 = k-mer + 32-bit count (small tables), 20 = k > 32 */
uint32_t katgpu_table_slot_bytes(const katgpu_table* t);

/* JellyfishHelper::getCount (lib/src/jellyfish_helper.cc:189-194) for a batch of packed k-mers. */
int katgpu_table_get(katgpu_table* t, const uint64_t* keys, size_t n, int canonicalise, uint64_t* counts);
/* Per-position coverage of a sequence: the loop of Sect::processSeq (src/sect.cc:516-535 -- validKmer + mer_dna +
 * JellyfishHelper::getCount for every k-window) and of Cold::processSeq (src/cold.cc).  counts[i], i in [0, n-k], is the
 * count of the window starting at bases[i], or 0 when that window holds any byte other than ACGTacgt; nothing is
 * written when n < k.  Several records may be profiled in one call by joining them with any non-base byte.
 * `canonicalise` is InputHandler::canonical.  The _device form takes device pointers and is asynchronous on the
 * context's stream. */
int katgpu_table_profile_host(katgpu_table* t, const char* bases, size_t n, int canonicalise, uint64_t* counts);
int katgpu_table_profile_device(katgpu_table* t, const uint8_t* dev_bases, size_t n, int canonicalise, uint64_t* dev_counts);
/* All (key,count) pairs in unspecified order (the eager_iterator walk, JF/.../large_hash_iterator.hpp:28-65).
 * Pass cap = 0 to query *n_out only. */
int katgpu_table_export(katgpu_table* t, uint64_t* keys, uint64_t* counts, size_t cap, size_t* n_out);
/* The same three for wide tables (33 <= k <= KATGPU_MAX_K). */
int katgpu_table_get_wide(katgpu_table* t, const uint64_t* keys_hi, const uint64_t* keys_lo, size_t n, int canonicalise, uint64_t* counts);
int katgpu_table_export_wide(katgpu_table* t, uint64_t* keys_hi, uint64_t* keys_lo, uint64_t* counts, size_t cap, size_t* n_out);
/* hash_counter::add for a batch of (k-mer, amount) records (what katgpu_table_merge_host is for narrow tables) */
int katgpu_table_merge_host_wide(katgpu_table* t, const uint64_t* keys_hi, const uint64_t* keys_lo, const uint64_t* counts, size_t n);
/* Owner-partitioned export / record merge of a wide table for the multi-GPU exchange (katgpu_table_partition_sizes serves both
 * table kinds; these two are katgpu_table_partition / katgpu_table_merge_device with (hi, lo) keys; device pointers). */
int katgpu_table_partition_wide(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets, uint64_t* dev_keys_hi, uint64_t* dev_keys_lo,
                                uint64_t* dev_counts);
int katgpu_table_merge_device_wide(katgpu_table* t, const uint64_t* dev_keys_hi, const uint64_t* dev_keys_lo, const uint64_t* dev_counts, size_t n);

/* ---- Jellyfish hash files (.jf, "binary/sorted"): replaces HashLoader::loadHash / JellyfishHelper::dumpHash
 *      (lib/src/jellyfish_helper.cc:97-187,248-256) and InputHandler::dump (lib/src/input_handler.cc:221-243).
 *      Records are written sorted by (matrix x k-mer) & (size-1), then k-mer, with the matrix in the JSON header, counts
 *      saturated to 4 bytes (lib/src/input_handler.cc:196) -- the layout jellyfish / KAT read.  Errors of the two
 *      host-only calls (and of katgpu_jf_load before it touches the device) are reported by katgpu_jf_last_error(). ---- */
int katgpu_jf_load(katgpu_ctx* ctx, const char* path, katgpu_table** out);     /* k and canonical come from the header */
int katgpu_jf_dump(katgpu_table* t, const char* path);
/* host only, no device needed */
int katgpu_jf_write_records(const char* path, uint32_t k, int canonical, const uint64_t* keys, const uint64_t* counts, size_t n);
int katgpu_jf_read_records(const char* path, uint32_t* k, int* canonical, uint64_t** keys, uint64_t** counts, size_t* n);  /* free with katgpu_free_host */
/* the two host-only calls for any k <= KATGPU_MAX_K: a k-mer is (hi, lo), hi = 0 for k <= 32 */
int katgpu_jf_write_records_wide(const char* path, uint32_t k, int canonical, const uint64_t* keys_hi, const uint64_t* keys_lo,
                                 const uint64_t* counts, size_t n);
int katgpu_jf_read_records_wide(const char* path, uint32_t* k, int* canonical, uint64_t** keys_hi, uint64_t** keys_lo, uint64_t** counts, size_t* n);
const char* katgpu_jf_last_error(void);

/* ---- reducers ---------------------------------------------------------------------------------------- */

/* Histogram::bin + merge (src/histogram.cc:162-199,146-160).  base/ceil from calcBase/calcCeil
 * (src/histogram.hpp:172-178); nb = ceil + 1 - base. */
int katgpu_hist(katgpu_table* t, uint64_t base, uint64_t ceil, uint64_t inc, uint64_t* out, size_t nb);

/* Gcp::analyse + merge (src/gcp.cc:158-197,128-138).  out: k rows (GC count 0..k-1; GC == k is dropped exactly as
 * the reference's k-row matrix drops it, src/gcp.cc:93) x (cvg_bins+1) columns, row-major. */
int katgpu_gcp(katgpu_table* t, double cvg_scale, uint32_t cvg_bins, uint64_t* out);

/* Comp::compare + merge (src/comp.cc:366-484,248-265) with CompCounters (lib/src/comp_counters.cc:91-140),
 * two-input form.  canon1/canon2 are the `canonical` flags of the two InputHandlers (quirks kept: pass 1
 * canonicalises the probe iff canon2, src/comp.cc:401; pass 2 always canonicalises, src/comp.cc:447).
 * main_mx: d1_bins x d2_bins row-major [scaled count in 1][scaled count in 2].
 * counters[13]: hash1_total, hash2_total, hash3_total, hash1_distinct, hash2_distinct, hash3_distinct,
 *   hash1_only_total, hash2_only_total, hash1_only_distinct, hash2_only_distinct,
 *   shared_hash1_total, shared_hash2_total, shared_distinct  (lib/include/kat/comp_counters.hpp).
 * spectra: 4 x min(d1_bins,d2_bins): spectrum1, spectrum2, shared_spectrum1, shared_spectrum2. */
int katgpu_comp(katgpu_table* t1, katgpu_table* t2, int canon1, int canon2,
                double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
                uint64_t* main_mx, uint64_t counters[13], uint64_t* spectra);

/* Three-input form (src/comp.cc:123-127,403-433,466-479): as katgpu_comp, plus the ends / middle / mixed matrices
 * (each d1_bins x d2_bins, row = scaled count in input 1, column = scaled count in input 3) and hash3_total /
 * hash3_distinct in counters[2] / counters[5]. */
int katgpu_comp3(katgpu_table* t1, katgpu_table* t2, katgpu_table* t3, int canon1, int canon2, int canon3,
                 double d1_scale, double d2_scale, uint32_t d1_bins, uint32_t d2_bins,
                 uint64_t* main_mx, uint64_t* ends_mx, uint64_t* middle_mx, uint64_t* mixed_mx,
                 uint64_t counters[13], uint64_t* spectra);

/* ---- multi-GPU: owner-partitioned merge of per-GPU partial tables (no reference analogue: KAT is one
 *      address space; this is the exchange step of BASELINE.json's north_star).  owner(kmer) depends only on the
 *      canonical form of the k-mer, so both comp inputs and both strands land on the same rank. ---- */
/* records per destination part */
int katgpu_table_partition_sizes(katgpu_table* t, uint32_t n_parts, uint64_t* sizes);
/* write (key,count) records grouped by part into caller-provided DEVICE buffers of sum(sizes) entries;
 * offsets[p] = exclusive prefix sum of sizes */
int katgpu_table_partition(katgpu_table* t, uint32_t n_parts, const uint64_t* offsets,
                           uint64_t* dev_keys, uint64_t* dev_counts);
/* add n (key,count) records held in DEVICE memory into t (exact 64-bit sums) */
int katgpu_table_merge_device(katgpu_table* t, const uint64_t* dev_keys, const uint64_t* dev_counts, size_t n);
int katgpu_table_merge_host(katgpu_table* t, const uint64_t* keys, const uint64_t* counts, size_t n);

/* ---- region-ordered exchange (multi-GPU; no counterpart in KAT, which merges per-thread results in one address space:
 *      ThreadedSparseMatrix::mergeThreadedMatricies lib/include/kat/sparse_matrix.hpp:325-335, Histogram::merge
 *      src/histogram.cc:146-160) ----
 * Ranks that count into tables of the same region grid (p1, p2) hold a k-mer in the same region index.  The sender extracts
 * its records grouped by owner and, within an owner, ordered by region; the owner applies the runs of each region in LDS. */
typedef struct { uint32_t k, canonical, n_regions, region_slots, p1, p2; uint64_t capacity; } katgpu_geometry;
int katgpu_table_geometry(const katgpu_table* t, katgpu_geometry* g);
/* pass 1: dev_region_counts[p * n_regions + g] (device, u32) = records of region g owned by part p (owner = hash of the canonical
 * k-mer, the rule of katgpu_table_partition); part_sizes[p] (host) = records owned by part p.  n_parts <= 256. */
int katgpu_table_extract_sizes(katgpu_table* t, uint32_t n_parts, uint32_t* dev_region_counts, uint64_t* part_sizes);
/* pass 2: the records.  Part p starts at sum(part_sizes[0..p)); inside a part records are ordered by region.  Counts are 32 bit:
 * a k-mer whose count does not fit (and the all-ones k-mer, which has no slot) is returned in the host arrays big_keys /
 * big_counts (*n_big entries, at most big_cap; 4200 always suffices) and its record, if any, carries count 0. */
int katgpu_table_extract(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint64_t* dev_keys, uint32_t* dev_counts,
                         uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big);
/* empty the table, keeping its storage and its region grid (the extracted table becomes the owner table) */
int katgpu_table_clear(katgpu_table* t);
/* add records with 32-bit counts (count 0 = skip) through the direct path */
int katgpu_table_merge_device32(katgpu_table* t, const uint64_t* dev_keys, const uint32_t* dev_counts, size_t n);
/* Owner side.  Source i holds n_records records of the sender's regions [g_lo, g_hi) in region order (dev_region_counts: u32 per
 * region, NULL if unknown); p1 / p2 name the grid the sender ordered them by.  Sources of this table's grid are applied region
 * by region in LDS, the others through the direct path.  Exact either way. */
typedef struct { const uint64_t* dev_keys; const uint32_t* dev_counts; const uint32_t* dev_region_counts; uint64_t n_records; uint32_t p1, p2; } katgpu_merge_source;
int katgpu_table_merge_regions(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const katgpu_merge_source* src);
/* The same records in 9 bytes instead of 12, for owners that share the sender's region grid: a record is what a packed slot holds of
 * the k-mer -- the remainder below the two placement digits, rb <= 44 bits -- below its count, 72 bits cut into dev_rem_lo (u32),
 * dev_rem_hi (u8) and dev_counts (u32: for rb > 40 its low rb - 40 bits are the remainder's top ones and the count has the >= 28 bits above
 * them; a count that does not fit travels in the big list, its record carrying 0); the region it lies in
 * says the rest (the merged hashes of KAT's workers travel as whole k-mers, lib/include/kat/sparse_matrix.hpp:324-335 and
 * lib/src/comp_counters.cc:230-254 sum in one address space: this is the wire format of their replacement).  katgpu_table_packed_records:
 * 1 when the table can give such records (a packed table: k <= 32 and >= 2^(2k - 44) regions, every table of size).  A merge
 * source of this form MUST carry its region counts and the sender's grid; if the receiving table has another grid by then (it grew), the
 * k-mers are rebuilt from the sender's grid and go through the direct path.  Exact either way. */
int katgpu_table_packed_records(const katgpu_table* t);
int katgpu_table_extract_packed(katgpu_table* t, uint32_t n_parts, const uint32_t* dev_region_counts, uint32_t* dev_rem_lo, uint8_t* dev_rem_hi,
                                uint32_t* dev_counts, uint64_t* big_keys, uint64_t* big_counts, uint32_t big_cap, uint32_t* n_big);
typedef struct { const uint32_t* dev_rem_lo; const uint8_t* dev_rem_hi; const uint32_t* dev_counts; const uint32_t* dev_region_counts; uint64_t n_records; uint32_t p1, p2; } katgpu_merge_source_packed;
int katgpu_table_merge_regions_packed(katgpu_table* t, uint32_t g_lo, uint32_t g_hi, uint32_t n_src, const katgpu_merge_source_packed* src);

/* ---- the exchange itself, over RCCL: one process per GPU (kat_amd/csrc/kg_comm.hip) ----
 * Replaces, across GPUs, what the reference does across threads of one process at the end of a run:
 * ThreadedSparseMatrix::mergeThreadedMatricies (lib/include/kat/sparse_matrix.hpp:324-335), ThreadedCompCounters::merge
 * (lib/src/comp_counters.cc:230-254), Histogram::merge (src/histogram.cc:146-160) -- and, before them, makes every k-mer's count
 * whole on one rank, which one address space gives the reference for free.
 *   rank 0:      katgpu_comm_unique_id(id); hand the KATGPU_COMM_ID_BYTES bytes to the other ranks (a file, a pipe, MPI, ...)
 *   every rank:  katgpu_init(own device); katgpu_comm_init(ctx, rank, world, id, &comm);
 *                count its share of the input into tables of the same size hint (-> the same region grid);
 *                katgpu_exchange_merge(comm, table) for every table; reduce (katgpu_hist / _gcp / _comp);
 *                katgpu_allreduce_u64(comm, result, n): every rank now holds the whole run's result.
 * Transport: RCCL (librccl is dlopen'ed when the first id is made; grouped ncclSend / ncclRecv on a stream of its own, chunk c on
 * the wire while chunk c-1 is merged) or, when ranks SHARE a device (RCCL refuses them) or KATGPU_COMM_TRANSPORT=shm asks for it,
 * files in /dev/shm (one node).  Ranks on distinct devices that cannot have RCCL do NOT fall back on /dev/shm: katgpu_comm_init
 * fails and says why (KATGPU_COMM_ALLOW_SHM=1 allows the fall-back).  katgpu_comm_transport() says which transport runs; _note()
 * why it is not RCCL.  No wait inside a collective is bounded by a wall clock (ranks may arrive minutes apart); a peer that died
 * is told from its heartbeat (KATGPU_COMM_TIMEOUT_S seconds of silence, default 60), a peer that failed from the abort flag it
 * raised.  world == 1 is legal and runs the whole protocol on the rank's own records. */
#define KATGPU_COMM_ID_BYTES 256
typedef struct katgpu_comm katgpu_comm;
int  katgpu_comm_unique_id(void* id_out /* KATGPU_COMM_ID_BYTES */);
int  katgpu_comm_init(katgpu_ctx* ctx, int rank, int world, const void* id, katgpu_comm** out);
void katgpu_comm_free(katgpu_comm* comm);
int  katgpu_comm_rank(const katgpu_comm* comm);
int  katgpu_comm_world(const katgpu_comm* comm);
const char* katgpu_comm_transport(const katgpu_comm* comm);        /* "rccl" | "shm" */
const char* katgpu_comm_transport_note(const katgpu_comm* comm);   /* "" or why RCCL is not in use */
int  katgpu_comm_distinct_devices(const katgpu_comm* comm);        /* how many different devices the ranks run on (1: they share one) */
int  katgpu_comm_barrier(katgpu_comm* comm);
/* Route every record of `t` to its owner rank, in place: afterwards the table holds exactly the k-mers this rank owns, counts summed
 * over all ranks; it keeps its storage and its region grid.  Collective: every rank calls it, with tables of one k / strand mode.
 * Wide tables (k > 32) take the simple route: records grouped by owner, all to all, the table emptied and refilled (it may grow). */
int  katgpu_exchange_merge(katgpu_comm* comm, katgpu_table* t);
/* The same exchange in two calls, so that the next input is counted while this table's records are on the wire (kat comp: the second
 * hash is counted while the first one's merge travels; the reference has nothing to overlap -- its merges are memory operations of one
 * process, lib/include/kat/sparse_matrix.hpp:324-335): begin extracts the records, empties the table and posts every chunk, from and
 * into a buffer of the exchange's own (send list + what arrives: ~2 x 9..12 bytes per record; the context's scratch arena stays the
 * counter's); finish waits chunk by chunk and applies.  Between the two calls `t` must not be touched; counting into OTHER tables of the
 * context is what the gap is for.  Two exchanges may be under way at once -- begin(t1), count input 2, begin(t2), finish(t1), finish(t2):
 * table 2's records travel while table 1's are applied -- and are finished in the order they were begun; no other collective of `comm`
 * (katgpu_exchange_merge, katgpu_allreduce_u64) may run until all are finished.  When a rank has no room for the buffer -- all ranks agree
 * on that -- and for wide tables, begin runs the whole exchange and finish returns at once.  Same result as katgpu_exchange_merge. */
int  katgpu_exchange_begin(katgpu_comm* comm, katgpu_table* t);
int  katgpu_exchange_finish(katgpu_comm* comm, katgpu_table* t);
/* buf[i] = sum over ranks of buf[i], on every rank (host memory; collective) */
int  katgpu_allreduce_u64(katgpu_comm* comm, uint64_t* buf, size_t n);
/* wall time spent so far in extraction / on the wire (posting + waiting) / merging / all-reducing (ms), bytes sent, merge calls */
int  katgpu_comm_stats(katgpu_comm* comm, double* ms_extract, double* ms_exchange, double* ms_merge, double* ms_allreduce,
                       uint64_t* bytes_sent, uint64_t* merge_launches);
/* the records katgpu_exchange_merge has sent so far, their bytes, and the form of the last exchange's (1: remainder + count, 9 bytes --
 * every rank's table had one grid; 0: key + count, 12 bytes) */
int  katgpu_comm_wire(katgpu_comm* comm, uint64_t* records_sent, uint64_t* record_bytes_sent, int* packed);

/* ---- measurement ------------------------------------------------------------------------------------- */
/* HIP-event timing of the kernels this ctx launched, per kernel class, accumulated since the last reset. */
typedef enum katgpu_kernel {
    KATGPU_K_COUNT = 0,      /* direct counter: extract + canonicalise + insert (one atomic per k-mer) */
    KATGPU_K_REGROW = 1,
    KATGPU_K_HIST = 2,
    KATGPU_K_GCP = 3,
    KATGPU_K_COMP_PASS1 = 4,
    KATGPU_K_COMP_PASS2 = 5,
    KATGPU_K_PARTITION = 6,
    KATGPU_K_MERGE = 7,
    KATGPU_K_PART_L1 = 8,    /* partitioned counter: extract + level-1 bucket histogram (k_p1_count + k_p1_scan) */
    KATGPU_K_PART_L2 = 9,    /* level-2 partition: one run per table region (k_p2) */
    KATGPU_K_PART_APPLY = 10,/* regions updated in LDS (k_p3_apply_pk, k_p3_apply2) */
    KATGPU_K_PART_L1S = 11,  /* extract + level-1 scatter (k_p1_scatter) */
    KATGPU_K_PROFILE = 12,   /* per-position lookups (k_profile) */
    KATGPU_K_SCAN = 13,      /* device-side record scan of raw FASTQ / FASTA bytes (kg_scan.hpp), units = raw bytes */
    KATGPU_K_NCLASSES = 14
} katgpu_kernel;
int katgpu_profile_reset(katgpu_ctx* ctx);
int katgpu_profile_get(katgpu_ctx* ctx, int kernel_class, uint64_t* launches, double* total_ms, uint64_t* units);

/* ---- device buffers + synthetic workload (bench / test support; not part of KAT's surface) ------------- */
int katgpu_dev_alloc(katgpu_ctx* ctx, size_t bytes, void** dev_ptr);
int katgpu_dev_free(katgpu_ctx* ctx, void* dev_ptr);
int katgpu_dev_upload(katgpu_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int katgpu_dev_download(katgpu_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int katgpu_dev_mem_info(katgpu_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
/* Counter-based generator (SplitMix64), identical bit-for-bit to kat_amd/synth.py:
 * genome: n output bytes of uniform ACGT bases -- with contig_len > 0 an 'N' follows every contig_len bases (the
 * base stream of an assembly FASTA cut into contigs), with contig_len == 0 there are no separators; reads: n_reads records of read_len bases + 'N' separator each
 * (stride read_len+1), sampled as PE fragments of `frag_len` from the genome with substitution-error
 * rate err_ppm / 1e6.  first_read = global index of this shard's first read. */
int katgpu_synth_genome_device(katgpu_ctx* ctx, uint8_t* dev_out, uint64_t n, uint64_t seed, uint64_t contig_len);
int katgpu_synth_reads_device(katgpu_ctx* ctx, const uint8_t* dev_genome, uint64_t genome_len,
                              uint8_t* dev_out, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                              uint32_t frag_len, uint32_t err_ppm, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* KATGPU_H */
