// kg_device.hpp -- device-side building blocks for gfx950: packed k-mer arithmetic and the HBM-resident
// open-addressed count table.  CDNA4 only (wave64); no portability layer.
//
// Two slot layouts for one-word k-mers (k <= 32):
//  "P8" (packed, DevTable::cbits != 0): ONE 64-bit word per slot = (remainder << cbits) | count, 0 = empty.  The placement hash
//       (below) is one to one, so the slot position (its region) already says the hash's digits and the slot keeps only the
//       remainder -- Jellyfish's quotienting (JF/include/jellyfish/large_hash_array.hpp:169-171) -- and what is left of the word
//       counts.  8 bytes per slot to sweep, scan and join instead of 12; a hit is one 64-bit add.  Used whenever the remainder
//       leaves at least PACK_MIN_CBITS count bits (tables of >= 2^(2k-44) regions: every table of size).
//  "KV12": keys[cap] (u64 k-mer, 0xFFFF..F = empty) and counts[cap] (u32) as two separate arrays (two coalesced streams for
//       the slot scans).  Small tables, k = 32, and the two-word tables of k > 32 (a third array).
// Counts are exact to 64 bits: whatever does not fit the 32-bit slot counter is chained into a small side table
// (key -> extra amount), the same idea as Jellyfish's "large" entries
// (deps/jellyfish-2.2.0/include/jellyfish/large_hash_array.hpp:668-700): count = counts[slot] + extra[key].  and
// the one key that collides with the empty marker (k = 32, all T, non-canonical) lives in a scalar.
// The layout is free to differ from Jellyfish's bit-packed array because hist/gcp/comp are
// order-independent sums over the multiset {(k-mer, count)}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kg {

constexpr uint64_t EMPTY = ~0ULL;
constexpr uint32_t OVF_CAP = 1u << 16;   // side table for amounts beyond the slot counter (32 bits, or cbits of a packed slot)
constexpr uint32_t PACK_MIN_CBITS = 20;  // a packed slot counts to at least 2^20 - 1 in place (what is beyond goes to the side table)
constexpr int MAX_PARTS = 1024;          // region digits per level of the partitioned counter (one lane per bucket in its scans): p1, p2 <= MAX_PARTS
constexpr uint32_t REGION_SLOTS = 8192;  // default slots per region: 96 KB of LDS (8 B key + 4 B count) in the apply kernel
constexpr uint32_t REGION_SLOTS_WIDE = 6144;   // ... of a wide table (k > 32: 20 B per slot, 120 KB)

// ctrs[] layout (u64 each)
constexpr int CTR_DISTINCT0 = 0;   // 64 stripes, summed on the host
constexpr int CTR_NSTRIPES = 64;
constexpr int CTR_ONES = 64;       // count of the all-ones key
constexpr int CTR_OVF_USED = 65;   // entries in the carry table
constexpr int CTR_FULL = 66;       // != 0: an insert ran out of probes / carry table full
constexpr int CTR_SCRATCH = 67;    // cursors / scratch for export & partition (8 words)
constexpr int CTR_WORDS = 80;

struct DevTable {
    uint64_t* keys;        // KV12: the k-mer (EMPTY = free).  P8: (remainder << cbits) | count (0 = free)
    uint64_t* keys_b;      // wide tables only (k > 32, see "wide keys" below): the second key word per slot; nullptr otherwise
    uint32_t* counts;      // nullptr for packed tables
    uint64_t cap;          // == n_regions * region_slots
    uint32_t n_regions;    // a k-mer hashes to one region and probes (linearly, wrapping) only inside it, so a region
    uint32_t region_slots; // is a self-contained little table that the partitioned counter can hold in LDS
    uint32_t p1, p2;       // n_regions == p1 * p2: region = b1 * p2 + b2, the two radix digits of the partitioned counter
    uint64_t* ovf_keys;   // OVF_CAP
    uint64_t* ovf_hi;     // OVF_CAP, extra amount (added to the slot's 32-bit counter)
    uint64_t* ctrs;       // CTR_WORDS
    uint32_t k;
    uint32_t canonical;
    uint32_t n1, l2;      // one-word tables ("placement" below): bits of the level-1 remainder (from k and p1); p2 == 1 << l2
    uint32_t cbits;       // packed tables: bits of the in-slot counter (64 - remainder bits); 0: KV12
    double inv_slots;     // packed tables: 1.0 / region_slots (slot index -> region without an integer division)
};

// ---- packed k-mer arithmetic (first base in the MSBs, A=0 C=1 G=2 T=3) ----

__device__ __forceinline__ uint64_t kmer_mask(uint32_t k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }

// reverse complement of a 2k-bit packed k-mer: bit-reverse the word (v_bfrev), swap the two bits of every
// base back, complement, align down.  Same function as word_reverse_complement (mer_dna.hpp:100-108).
__device__ __forceinline__ uint64_t kmer_revcomp(uint64_t x, uint32_t k) {
    uint64_t r = __brevll(x);
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
    r = ~r;
    return r >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t kmer_canonical(uint64_t x, uint32_t k) {
    uint64_t rc = kmer_revcomp(x, k);
    return rc < x ? rc : x;
}

// #G + #C (str_utils.hpp:151-161 on the packed form): C = 01, G = 10 -> the two bits differ
__device__ __forceinline__ uint32_t kmer_gc(uint64_t x, uint32_t k) {
    return __popcll((x ^ (x >> 1)) & 0x5555555555555555ULL & kmer_mask(k));
}

// ---- hashing ----
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t x) {   // murmur3 finaliser: bijective, avalanching
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// home slot from three disjoint bit fields of the 64-bit hash, each reduced with one 32-bit high multiply (no division):
// level-1 digit from bits 63..32, level-2 digit from bits 43..12, offset inside the region from bits 31..0
__device__ __forceinline__ uint32_t digit1_of_hash(uint64_t h, uint32_t p1) { return __umulhi((uint32_t)(h >> 32), p1); }
__device__ __forceinline__ uint32_t digit2_of_hash(uint64_t h, uint32_t p2) { return __umulhi((uint32_t)(h >> 12), p2); }
__device__ __forceinline__ uint32_t region_of_hash(uint64_t h, uint32_t p1, uint32_t p2) { return digit1_of_hash(h, p1) * p2 + digit2_of_hash(h, p2); }
__device__ __forceinline__ uint32_t offset_of_hash(uint64_t h, uint32_t region_slots) { return __umulhi((uint32_t)h, region_slots); }
// ---- placement of one-word k-mers (k <= 32): a one-to-one map of the 2k-bit k-mer onto (digit 1, digit 2, remainder) ----
// Jellyfish stores only what the slot position does not already say about a key (its hash is an invertible matrix product and the
// array keeps the remainder, JF/include/jellyfish/large_hash_array.hpp:169-171).  Same idea for the partitioned counter, built so
// that every level PEELS bits off the k-mer instead of re-hashing it (two Feistel-style steps: a field is displaced by a hash of
// the bits below it, which stay as they are):
//     key = H : L      (L = the low n1 bits, H < 2^hb1 <= p1 the bits above; hb1 = floor(log2 p1), n1 = 2k - hb1)
//     d1  = (H + g1(L)) mod p1                                   level-1 digit; r1 = L is what a level-1 item carries (n1 bits)
//     L   = H2 : L2    (L2 = the low rb = n1 - l2 bits)
//     d2  = H2 ^ g2(L2)                                          level-2 digit (p2 = 2^l2); rem = L2 is what a level-2 item and a packed slot carry
//     home offset = g3(rem) scaled to the region's slots
// region = d1 * p2 + d2.  g1, g2, g3 are three multiplicative hashes (fold to 32 bits, xor-shift, one 32-bit multiply by an odd
// constant; the TOP bits are used, which depend on every input bit) -- they need not be invertible: given (d1, d2, rem), H2 = d2 ^
// g2(rem), L = H2 : rem, H = (d1 - g1(L)) mod p1.  (key) -> (d1, d2, rem) is one to one, digits are uniform whenever the hashed low
// bits are, and NOTHING is multiplied in 64 bits: level 1 spends one 32-bit multiply per k-mer, level 2 one, the apply one, and the
// items shrink as they go (8 -> 6 -> 5 bytes at k = 27 with 2^19 regions).  The first edition (two multiply / xor-shift stages on
// 2k and n1 bits) cost level 1 and level 2 five 32-bit multiplies and a dozen 64-bit shifts per k-mer each.
// p1 is any number <= 1024 (so that a table of any size has full-size regions); p2 a power of two.
constexpr uint32_t PLACE_G1 = 0x9E3779B1u, PLACE_G2 = 0x85EBCA6Bu, PLACE_G3 = 0xC2B2AE35u;   // odd

struct Place {             // the bit budget of one table: wave-uniform, a handful of SGPRs
    uint32_t n, p1;        // 2k; level-1 digits
    uint32_t n1, rb;       // bits of r1 (the k-mer's low bits a level-1 item carries); bits of the remainder (what a level-2 item and a packed slot carry)
    uint32_t l2e;          // bits of the level-2 digit's field: min(l2, n1)
    uint64_t m1, mr;       // masks of n1 and rb bits
};
__device__ __host__ __forceinline__ uint64_t low_mask(uint32_t bits) { return bits >= 64 ? ~0ULL : (1ULL << bits) - 1; }
// floor(h * n / 2^32): a 32-bit hash scaled to [0, n) with one v_mul_hi_u32 (full rate on gfx950: tools/ubench_valu.hip)
__device__ __host__ __forceinline__ uint32_t place_scale(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * n) >> 32); }
// 32 well-mixed bits (the top ones) of up to 64 input bits
__device__ __host__ __forceinline__ uint32_t place_mix(uint32_t lo, uint32_t hi, uint32_t c) {
    uint32_t x = lo ^ ((hi << 19) | (hi >> 13));
    x ^= x >> 15;
    return x * c;
}
__device__ __host__ __forceinline__ uint32_t place_mix(uint64_t v, uint32_t c) { return place_mix((uint32_t)v, (uint32_t)(v >> 32), c); }
// bits of r1 (host: once per table)
inline uint32_t place_n1(uint32_t k, uint32_t p1) {
    uint32_t hb1 = 0;
    while ((2u << hb1) <= p1) ++hb1;                               // floor(log2 p1)
    const uint32_t n = 2 * k;
    return n > hb1 ? n - hb1 : 0;
}
__device__ __host__ __forceinline__ Place place_make(uint32_t k, uint32_t p1, uint32_t n1, uint32_t l2) {
    Place p;
    p.n = 2 * k; p.p1 = p1; p.n1 = n1;
    p.l2e = l2 < n1 ? l2 : n1;                                     // (a key space smaller than the grid: the digit just stays small)
    p.rb = n1 - p.l2e;
    p.m1 = low_mask(n1); p.mr = low_mask(p.rb);
    return p;
}
struct Placed { uint32_t d1, d2; uint64_t rem; };
// g1 scaled to [0, p1)
__device__ __host__ __forceinline__ uint32_t place_g1(uint32_t l_lo, uint32_t l_hi, const Place& p) { return place_scale(place_mix(l_lo, l_hi, PLACE_G1), p.p1); }
__device__ __host__ __forceinline__ uint32_t place_g2(uint32_t r_lo, uint32_t r_hi, const Place& p) { return p.l2e ? place_mix(r_lo, r_hi, PLACE_G2) >> (32 - p.l2e) : 0u; }
// level-1 digit of a k-mer
__device__ __host__ __forceinline__ uint32_t place_digit1_of(uint64_t key, const Place& p) {
    const uint64_t L = key & p.m1;
    const uint32_t H = p.n1 < 64 ? (uint32_t)(key >> p.n1) : 0u;
    uint32_t d = H + place_g1((uint32_t)L, (uint32_t)(L >> 32), p);
    return d >= p.p1 ? d - p.p1 : d;
}
// level-2 digit and remainder of r1 (= the k-mer's low n1 bits)
__device__ __host__ __forceinline__ uint32_t place_digit2_of(uint64_t r1, const Place& p) {
    const uint64_t L2 = r1 & p.mr;
    const uint32_t H2 = p.rb < 64 ? (uint32_t)(r1 >> p.rb) : 0u;
    return H2 ^ place_g2((uint32_t)L2, (uint32_t)(L2 >> 32), p);
}
// y2 = d2 : rem, the form the level-2 digit and the remainder travel in together (n1 <= 64 bits); "base1" is what the inverse needs
// to know of the level-1 digit: the digit (a name from the first edition, where it was the smallest hash value of the digit)
__device__ __host__ __forceinline__ uint64_t place_base1(uint32_t d, const Place&) { return d; }
__device__ __host__ __forceinline__ uint64_t place_stage2(uint64_t r1, const Place& p) { return (p.rb < 64 ? (uint64_t)place_digit2_of(r1, p) << p.rb : 0ULL) | (r1 & p.mr); }   // y2 = d2 : rem
__device__ __host__ __forceinline__ uint32_t place_digit2(uint64_t y2, const Place& p) { return p.rb < 64 ? (uint32_t)(y2 >> p.rb) : 0u; }
__device__ __host__ __forceinline__ uint64_t place_rem(uint64_t y2, const Place& p) { return y2 & p.mr; }
__device__ __host__ __forceinline__ Placed place_hash(uint64_t key, const Place& p) {
    Placed r;
    r.d1 = place_digit1_of(key, p);
    r.d2 = place_digit2_of(key & p.m1, p);
    r.rem = key & p.mr;
    return r;
}
// the k-mer of a level-1 item: its low n1 bits r1 and its level-1 digit
__device__ __host__ __forceinline__ uint64_t place_key_r1(uint32_t d1, uint64_t r1, const Place& p) {
    uint32_t H = d1 + p.p1 - place_g1((uint32_t)r1, (uint32_t)(r1 >> 32), p);
    H = H >= p.p1 ? H - p.p1 : H;
    return (p.n1 < 64 ? (uint64_t)H << p.n1 : 0ULL) | r1;
}
// the inverse: the k-mer of remainder `rem` in region (d1, d2) (the digits are the caller's: uniform over a region)
__device__ __host__ __forceinline__ uint64_t place_key_d(uint32_t d1, uint32_t d2, uint64_t rem, const Place& p) {
    const uint32_t H2 = d2 ^ place_g2((uint32_t)rem, (uint32_t)(rem >> 32), p);
    const uint64_t L = (p.rb < 64 ? (uint64_t)H2 << p.rb : 0ULL) | rem;
    uint32_t H = d1 + p.p1 - place_g1((uint32_t)L, (uint32_t)(L >> 32), p);
    H = H >= p.p1 ? H - p.p1 : H;
    return (p.n1 < 64 ? (uint64_t)H << p.n1 : 0ULL) | L;
}
// ... from base1 = place_base1(d1) and y2 = d2 : rem
__device__ __host__ __forceinline__ uint64_t place_key(uint64_t base1, uint64_t y2, const Place& p) { return place_key_d((uint32_t)base1, place_digit2(y2, p), y2 & p.mr, p); }
// home offset inside a region of S slots from the remainder
__device__ __host__ __forceinline__ uint32_t place_offset(uint64_t rem, const Place& p, uint32_t S) {
    return place_scale(place_mix((uint32_t)rem, (uint32_t)(rem >> 32), PLACE_G3), S);
}

struct Probe {
    uint64_t base;   // first slot of the region
    uint32_t s;      // current offset inside the region
    uint32_t S;
    uint64_t rem;    // the placement hash's remainder: what a packed slot holds of the k-mer
    __device__ __forceinline__ uint64_t pos() const { return base + s; }
    __device__ __forceinline__ void next() { s = s + 1 == S ? 0 : s + 1; }
};
// home offset of a k-mer inside its region (kernels that are handed the region -- the applies, the merges -- need only this)
__device__ __forceinline__ uint32_t home_offset(const uint64_t key, const DevTable& t) {
    const Place pl = place_make(t.k, t.p1, t.n1, t.l2);
    return place_offset(place_hash(key, pl).rem, pl, t.region_slots);
}
// the same for kernels that go through one region's k-mers: the level-1 digit is the region's, so its base is computed once
struct RegionPlace { Place pl; uint32_t d1, d2; uint32_t S; };
__device__ __forceinline__ RegionPlace region_place(const DevTable& t, uint32_t region) {
    RegionPlace rp;
    rp.pl = place_make(t.k, t.p1, t.n1, t.l2);
    rp.d1 = region >> t.l2;
    rp.d2 = region & (t.p2 - 1);
    rp.S = t.region_slots;
    return rp;
}
__device__ __forceinline__ uint64_t rem_in(const uint64_t key, const RegionPlace& rp) { return key & rp.pl.mr; }      // (of a k-mer that lies in the region)
__device__ __forceinline__ uint32_t home_offset_in(const uint64_t key, const RegionPlace& rp) { return place_offset(rem_in(key, rp), rp.pl, rp.S); }
__device__ __forceinline__ uint64_t key_in(const uint64_t rem, const RegionPlace& rp) { return place_key_d(rp.d1, rp.d2, rem, rp.pl); }
__device__ __forceinline__ Probe probe_start(const uint64_t key, const DevTable& t) {
    Probe p;
    const uint32_t region_slots = t.region_slots;
    p.S = region_slots;
    const Place pl = place_make(t.k, t.p1, t.n1, t.l2);
    const Placed h = place_hash(key, pl);
    p.base = (uint64_t)((h.d1 << t.l2) | h.d2) * region_slots;
    p.s = place_offset(h.rem, pl, region_slots);
    p.rem = h.rem;
    return p;
}
// owner part of a k-mer for the multi-GPU merge: a second, independent mix of the CANONICAL form
__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t k, uint32_t n_parts) {
    uint64_t c = kmer_canonical(key, k);
    return (uint32_t)__umul64hi(mix64(c ^ 0x9E3779B97F4A7C15ULL), (uint64_t)n_parts);
}

// ---- carry side table ----
// keyed by the k-mer (KV12 one-word tables) or by the SLOT (packed and wide tables: slots never move within a table's life, and a
// regrow re-adds full counts)
__device__ inline void ovf_add(const DevTable& t, uint64_t key, uint64_t hi /* extra amount */) {
    uint32_t p = (uint32_t)(mix64(key) >> 40) & (OVF_CAP - 1);
    for (uint32_t i = 0; i < OVF_CAP; ++i) {
        uint64_t cur = atomicCAS((unsigned long long*)&t.ovf_keys[p], (unsigned long long)EMPTY, (unsigned long long)key);
        if (cur == EMPTY) atomicAdd((unsigned long long*)&t.ctrs[CTR_OVF_USED], 1ULL);
        if (cur == EMPTY || cur == key) { atomicAdd((unsigned long long*)&t.ovf_hi[p], (unsigned long long)hi); return; }
        p = (p + 1) & (OVF_CAP - 1);
    }
    atomicOr((unsigned long long*)&t.ctrs[CTR_FULL], 2ULL);
}
__device__ inline uint64_t ovf_get(const DevTable& t, uint64_t key) {
    uint32_t p = (uint32_t)(mix64(key) >> 40) & (OVF_CAP - 1);
    for (uint32_t i = 0; i < OVF_CAP; ++i) {
        uint64_t cur = t.ovf_keys[p];
        if (cur == key) return t.ovf_hi[p];
        if (cur == EMPTY) return 0;
        p = (p + 1) & (OVF_CAP - 1);
    }
    return 0;
}
__device__ __forceinline__ bool ovf_by_slot(const DevTable& t) { return t.keys_b != nullptr || t.cbits != 0; }

// ---- packed slots ----
__device__ __host__ __forceinline__ uint64_t pk_cmask(uint32_t cbits) { return (1ULL << cbits) - 1; }
__device__ __host__ __forceinline__ uint64_t pk_half(uint32_t cbits) { return 1ULL << (cbits - 1); }
__device__ __forceinline__ uint64_t pk_count(uint64_t w, uint32_t cbits) { return w & pk_cmask(cbits); }
__device__ __forceinline__ uint64_t pk_rem(uint64_t w, uint32_t cbits) { return w >> cbits; }
__device__ __forceinline__ bool pk_holds(uint64_t w, uint64_t rem, uint32_t cbits) { return w != 0 && (w >> cbits) == rem; }
// region of a slot index: pos / region_slots through the reciprocal (pos < 2^40, region_slots < 2^14: one correction step suffices)
__device__ __forceinline__ uint32_t region_of_pos(const DevTable& t, uint64_t pos) {
    uint32_t r = (uint32_t)((double)pos * t.inv_slots);
    if ((uint64_t)r * t.region_slots > pos) --r;
    else if ((uint64_t)(r + 1) * t.region_slots <= pos) ++r;
    return r;
}
// the k-mer a packed slot holds (slow form: any slot; the region kernels use key_in)
__device__ __forceinline__ uint64_t pk_key(const DevTable& t, uint64_t pos, uint64_t w) {
    const Place pl = place_make(t.k, t.p1, t.n1, t.l2);
    const uint32_t region = region_of_pos(t, pos);
    return place_key_d(region >> t.l2, region & (t.p2 - 1), w >> t.cbits, pl);
}
// amount = q * half + r (r < half): the slot takes r (and, should that carry it past its field, gives `half` back), the side table q * half
__device__ __forceinline__ void pk_split(uint64_t amount, uint32_t cbits, uint64_t& q, uint64_t& r) { q = amount >> (cbits - 1); r = amount & (pk_half(cbits) - 1); }

// One slot of a one-word table, whatever its layout: occupied?, the k-mer, the in-slot count (the side table's part is slot_count's)
struct SlotView { bool occ; uint64_t key; uint64_t cnt; };
__device__ __forceinline__ SlotView slot_view(const DevTable& t, uint64_t pos) {
    SlotView v;
    const uint64_t w = t.keys[pos];
    if (t.cbits) { v.occ = w != 0; v.cnt = pk_count(w, t.cbits); v.key = v.occ ? pk_key(t, pos, w) : EMPTY; }
    else { v.occ = w != EMPTY; v.key = w; v.cnt = v.occ ? t.counts[pos] : 0; }
    return v;
}
// ... of a region the caller walks (rp = region_place of that region)
__device__ __forceinline__ SlotView slot_view_in(const DevTable& t, const RegionPlace& rp, uint64_t pos) {
    SlotView v;
    const uint64_t w = t.keys[pos];
    if (t.cbits) { v.occ = w != 0; v.cnt = pk_count(w, t.cbits); v.key = v.occ ? key_in(pk_rem(w, t.cbits), rp) : EMPTY; }
    else { v.occ = w != EMPTY; v.key = w; v.cnt = v.occ ? t.counts[pos] : 0; }
    return v;
}

// full 64-bit count of an occupied slot: in-slot count + the side table's amount
__device__ __forceinline__ uint64_t slot_total(const DevTable& t, uint64_t pos, uint64_t key, uint64_t in_slot, uint32_t n_ovf) {
    uint64_t c = in_slot;
    if (n_ovf) c += ovf_get(t, ovf_by_slot(t) ? pos : key);
    return c;
}
__device__ __forceinline__ uint64_t slot_count(const DevTable& t, uint64_t pos, uint64_t key, uint32_t n_ovf) {      // KV12 and wide tables
    return slot_total(t, pos, key, t.counts[pos], n_ovf);
}

// An insert that walked its whole region without finding room: the host keeps every table under its fill limit and the hash
// spreads k-mers evenly, so this is "Hash full" (reported through CTR_FULL).
__device__ inline bool table_full(const DevTable& t) {
    atomicOr((unsigned long long*)&t.ctrs[CTR_FULL], 1ULL);
    return false;
}

// insert-or-add into a packed table.  A claim is ONE compare-and-swap on the slot word (0 -> remainder | count): claim and first
// count together.  INVARIANT of every kernel that writes packed slots: at kernel boundaries an occupied slot counts 1 .. half
// (half = 2^(cbits-1)); what is beyond lives in the side table, keyed by the slot.  It is what lets the apply kernels add with
// no-return LDS atomics (a walk adds less than half) and the +1 path below add with a plain atomic.
//  UNIT (amount == 1; the direct counter, spill lists): atomicAdd on the word; the one add that lifts the count to half + 1 hands
//       `half` to the side table.  No retry loop: a heavy hitter -- every lane of the chip on one slot -- costs what its atomics
//       cost (a compare-and-swap loop there is quadratic: each success fails everyone else's expected value).  The count
//       overshoots half + 1 only by the adds that land before the hand-over does (thousands at most; the field holds half - 1 more).
//  general amounts (regrow, merges): a compare-and-swap loop that leaves the count in 1 .. half.  Its callers bring every k-mer
//       once per source, so there is no contention to speak of.  One kernel uses one of the two forms, never both on a slot.
template <bool UNIT>
__device__ inline bool table_add_pk(const DevTable& t, uint64_t key, uint64_t amount, uint32_t& new_distinct) {
    const uint32_t cb = t.cbits;
    const uint64_t cmask = pk_cmask(cb), half = pk_half(cb);
    Probe pr = probe_start(key, t);
    uint64_t q, r;
    pk_split(amount, cb, q, r);                                    // amount = q * half + r, r < half
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        unsigned long long* slot = (unsigned long long*)&t.keys[pr.pos()];
        unsigned long long w = *slot;
        if (w == 0) {                                              // claim with the in-slot part of the amount (never 0: an occupied slot counts >= 1)
            const uint64_t in = r ? r : half, qq = r ? q : q - 1;
            w = atomicCAS(slot, 0ULL, (unsigned long long)((pr.rem << cb) | in));
            if (w == 0) { ++new_distinct; if (qq) ovf_add(t, pr.pos(), qq * half); return true; }
        }
        if ((w >> cb) != pr.rem) continue;
        if (UNIT) {
            const unsigned long long old = atomicAdd(slot, 1ULL);
            if ((old & cmask) == half) {                           // this add made it half + 1
                atomicAdd(slot, (unsigned long long)(0ULL - half));     // (no borrow: the count is at least half + 1 until this lands)
                ovf_add(t, pr.pos(), half);
            }
            return true;
        }
        if (r) {
            for (;;) {
                uint64_t c = (w & cmask) + r, qq = q;
                if (c > half) { c -= half; ++qq; }                 // (old count <= half, r < half: 1 <= c <= half after this)
                const unsigned long long got = atomicCAS(slot, w, (unsigned long long)((w & ~cmask) | c));
                if (got == w) { q = qq; break; }
                w = got;
            }
        }
        if (q) ovf_add(t, pr.pos(), q * half);
        return true;
    }
    return table_full(t);
}

// ---- insert-or-add: the replacement for array_base::add (large_hash_array.hpp:298-302) ----
// KV12: claim = CAS on the key word; add = returning 32-bit atomic add whose carry is chained.  The first look at a
// slot is a plain load: keys are write-once, so a stale "empty" only costs the CAS we would have issued anyway.
// new_distinct is accumulated per lane and flushed once per wave (one striped atomic instead of one per claim).
__device__ __forceinline__ bool table_add(const DevTable& t, uint64_t key, uint64_t amount, uint32_t& new_distinct) {
    if (amount == 0) return true;                                  // (a stored k-mer always has a count: comp's pass forms rely on it)
    if (t.cbits) return table_add_pk<false>(t, key, amount, new_distinct);
    if (key == EMPTY) { atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], (unsigned long long)amount); return true; }
    Probe pr = probe_start(key, t);
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        const uint64_t pos = pr.pos();
        uint64_t cur = t.keys[pos];
        if (cur == EMPTY) {
            cur = atomicCAS((unsigned long long*)&t.keys[pos], (unsigned long long)EMPTY, (unsigned long long)key);
            if (cur == EMPTY) { ++new_distinct; cur = key; }
        }
        if (cur == key) {
            uint32_t low = (uint32_t)amount;
            uint64_t hi = amount >> 32;
            uint32_t old = atomicAdd(&t.counts[pos], low);
            if ((uint64_t)old + low > 0xFFFFFFFFULL) ++hi;
            if (hi) ovf_add(t, key, hi << 32);
            return true;
        }
    }
    return table_full(t);
}

// +1 on the hot path of K1: same claim protocol, but the counter add is a NO-RETURN atomic, so a lane never waits for
// it (measured on MI355X: load + returning add chain 12-15 G k-mers/s, load + no-return add 20.7 G/s; the L2 atomic
// units saturate at ~22 G adds/s).  Without the returned value a 32-bit wrap cannot be seen here; the host instead
// guarantees it cannot happen: before the adds launched since the last k_sweep could lift any counter past 2^32-1 it
// runs k_sweep, which moves 2^31 from every counter >= 2^31 into the side table (kg_count.hip: maybe_sweep).
// (Packed tables: table_add_pk<UNIT>, which needs no sweep -- its hand-over to the side table happens in place.)
__device__ __forceinline__ bool table_inc(const DevTable& t, uint64_t key, uint32_t& new_distinct) {
    if (t.cbits) return table_add_pk<true>(t, key, 1ULL, new_distinct);
    if (key == EMPTY) { atomicAdd((unsigned long long*)&t.ctrs[CTR_ONES], 1ULL); return true; }
    Probe pr = probe_start(key, t);
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        const uint64_t pos = pr.pos();
        uint64_t cur = t.keys[pos];
        if (cur == EMPTY) {
            cur = atomicCAS((unsigned long long*)&t.keys[pos], (unsigned long long)EMPTY, (unsigned long long)key);
            if (cur == EMPTY) { ++new_distinct; cur = key; }
        }
        if (cur == key) {
            (void)__hip_atomic_fetch_add(&t.counts[pos], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return table_full(t);
}

// ---- lookup: get_val_for_key (large_hash_array.hpp:358-376) on an immutable table ----
__device__ __forceinline__ uint64_t table_get(const DevTable& t, uint64_t key, uint32_t n_ovf) {
    if (!t.cbits && key == EMPTY) return t.ctrs[CTR_ONES];
    Probe pr = probe_start(key, t);
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        const uint64_t pos = pr.pos();
        const uint64_t cur = t.keys[pos];
        if (t.cbits) {
            if (cur == 0) return 0;
            if ((cur >> t.cbits) == pr.rem) return slot_total(t, pos, key, pk_count(cur, t.cbits), n_ovf);
        } else {
            if (cur == key) return slot_count(t, pos, key, n_ovf);
            if (cur == EMPTY) return 0;
        }
    }
    return 0;
}

// ---- wide keys: 33 <= k <= 63 --------------------------------------------------------------------------------------
// The reference keeps a k-mer in uint64_t[ceil(k/32)] (mer_dna.hpp:235-258) and its table stores arbitrary key widths bit-packed;
// here a k-mer of up to 63 bases is a 2k <= 126-bit word cut into TWO 63-BIT halves, a = bits 125..63, b = bits 62..0, held in two
// parallel arrays keys[] / keys_b[].  Neither half can equal the all-ones EMPTY marker, so a slot is claimed without a 128-bit
// atomic and without anyone waiting for anyone: CAS a into keys[], then CAS b into keys_b[] -- whoever finds its own `a` in a
// slot tries to install its own `b`, and either wins, finds its `b` already there, or finds another k-mer's (same a, other b)
// and moves on.  A slot whose `a` is set always gets a `b` from one of the lanes that saw it, so tables are complete when the
// kernel ends; "distinct" is counted where `b` is installed.  (k = 64 would need the 128th bit and is not supported.)
struct KeyW { uint64_t a, b; };
constexpr uint64_t M63 = 0x7FFFFFFFFFFFFFFFULL;
__device__ __host__ __forceinline__ KeyW keyw_from_words(uint64_t hi, uint64_t lo) { return KeyW{(hi << 1) | (lo >> 63), lo & M63}; }
__device__ __host__ __forceinline__ uint64_t keyw_hi(KeyW x) { return x.a >> 1; }
__device__ __host__ __forceinline__ uint64_t keyw_lo(KeyW x) { return x.b | (x.a << 63); }

__device__ __forceinline__ uint64_t revcomp_word(uint64_t x) {       // all 32 bases of a word
    uint64_t r = __brevll(x);
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
    return ~r;
}
// reverse complement of the 2k-bit word (hi, lo), 33 <= k <= 63: reverse all 64 bases, then drop the 64 - k complemented pad bases
__device__ __forceinline__ void revcomp_words(uint64_t hi, uint64_t lo, uint32_t k, uint64_t& rhi, uint64_t& rlo) {
    const uint64_t H = revcomp_word(lo), L = revcomp_word(hi);
    const uint32_t s = 128 - 2 * k;                                   // 2 .. 62
    rlo = (L >> s) | (H << (64 - s));
    rhi = H >> s;
}
__device__ __forceinline__ KeyW keyw_canonical(KeyW x, uint32_t k) {
    const uint64_t hi = keyw_hi(x), lo = keyw_lo(x);
    uint64_t rhi, rlo;
    revcomp_words(hi, lo, k, rhi, rlo);
    const bool rc_less = rhi < hi || (rhi == hi && rlo < lo);
    return rc_less ? keyw_from_words(rhi, rlo) : x;
}
__device__ __forceinline__ uint32_t keyw_gc(KeyW x, uint32_t k) { return kmer_gc(keyw_lo(x), 32) + kmer_gc(keyw_hi(x), k - 32); }
__device__ __forceinline__ uint64_t keyw_hash(KeyW x) { return mix64(x.b ^ (x.a * 0x9E3779B97F4A7C15ULL)); }
// owner part of a wide k-mer for the multi-GPU merge: a second mix of the CANONICAL form, as owner_of does for one-word k-mers
__device__ __forceinline__ uint32_t owner_of_w(KeyW key, uint32_t k, uint32_t n_parts) {
    const KeyW c = keyw_canonical(key, k);
    return (uint32_t)__umul64hi(mix64(keyw_hash(c) ^ 0x9E3779B97F4A7C15ULL), (uint64_t)n_parts);
}
__device__ __forceinline__ Probe probe_start_w(KeyW key, const DevTable& t) {
    const uint64_t h = keyw_hash(key);
    Probe p;
    p.base = (uint64_t)region_of_hash(h, t.p1, t.p2) * t.region_slots;
    p.s = offset_of_hash(h, t.region_slots);
    p.S = t.region_slots;
    return p;
}

__device__ __forceinline__ bool table_add_w(const DevTable& t, KeyW key, uint64_t amount, uint32_t& new_distinct) {
    Probe pr = probe_start_w(key, t);
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        const uint64_t pos = pr.pos();
        uint64_t a = t.keys[pos];                                   // write-once words: a stale EMPTY only costs the CAS
        if (a == EMPTY) {
            a = atomicCAS((unsigned long long*)&t.keys[pos], (unsigned long long)EMPTY, (unsigned long long)key.a);
            if (a == EMPTY) a = key.a;
        }
        if (a != key.a) continue;
        uint64_t b = t.keys_b[pos];
        if (b == EMPTY) {
            b = atomicCAS((unsigned long long*)&t.keys_b[pos], (unsigned long long)EMPTY, (unsigned long long)key.b);
            if (b == EMPTY) { ++new_distinct; b = key.b; }
        }
        if (b != key.b) continue;
        const uint32_t low = (uint32_t)amount;
        uint64_t hi = amount >> 32;
        const uint32_t old = atomicAdd(&t.counts[pos], low);
        if ((uint64_t)old + low > 0xFFFFFFFFULL) ++hi;
        if (hi) ovf_add(t, pos, hi << 32);
        return true;
    }
    atomicOr((unsigned long long*)&t.ctrs[CTR_FULL], 1ULL);
    return false;
}

__device__ __forceinline__ uint64_t table_get_w(const DevTable& t, KeyW key, uint32_t n_ovf) {
    Probe pr = probe_start_w(key, t);
    for (uint32_t probe = 0; probe < t.region_slots; ++probe, pr.next()) {
        const uint64_t pos = pr.pos();
        const uint64_t a = t.keys[pos];
        if (a == EMPTY) return 0;
        if (a == key.a && t.keys_b[pos] == key.b) return slot_count(t, pos, pos, n_ovf);
    }
    return 0;
}

// A lane's value for the whole wave, the lane wave-uniform (a ballot's first set bit, lane 0 ...): v_readlane_b32, a few cycles --
// __shfl of a uniform lane is a ds_bpermute through the LDS crossbar, ~200 cycles of latency on gfx950 (tools/ubench_valu.hip).
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t lane_value(uint64_t v, int src) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP steps (row_shr 1, 2, 4, 8 inside the rows of sixteen, then row_bcast 15 / 31
// across them): each step is a move through the data-parallel-primitive path of the VALU, a few cycles -- the __shfl_up form is six
// dependent ds_bpermute round trips (~200 cycles each on gfx950), between two barriers of every level-1 and level-2 tile.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);     // row_shr:1 (zeros come in at a row's start)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);     // row_shr:8: every row holds its own scan
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3: lane 15 / 47's total
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3: lane 31's total
    return (uint32_t)x;
}

// flush a per-lane "new distinct" tally: wave reduction, lane 0 adds into one of 64 stripes
__device__ __forceinline__ void flush_distinct(const DevTable& t, uint32_t new_distinct) {
    uint32_t v = new_distinct;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) {
        uint32_t stripe = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (CTR_NSTRIPES - 1);
        atomicAdd((unsigned long long*)&t.ctrs[CTR_DISTINCT0 + stripe], (unsigned long long)v);
    }
}

// ---- counter-based RNG shared with kat_amd/synth.py ----
__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ uint64_t rng2(uint64_t seed, uint64_t idx) {
    return splitmix64(splitmix64(seed) ^ (idx * 0xD1342543DE82EF95ULL));
}

}  // namespace kg
