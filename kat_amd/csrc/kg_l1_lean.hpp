// kg_l1_lean.hpp -- the per-window arithmetic of level 1's ranking sweep (kg_partition.hpp: k_p1v2_scatter) on explicit 32-bit halves.
//
// The sweep is VALU-bound: per k-mer the compiler's 64-bit form spends ~80 instructions, nine of them 64-bit
// shifts (window step, fwd(), the rolling reverse complement, the validity mask) and five exec-mask branches.  Here the window is
// three 32-bit words stepped with two funnel shifts, the k-mer and its reverse complement are register pairs, the validity of all 16
// windows of a lane is ONE 16-bit mask computed once per tile (a dilation of the 48 flag bits), and the level-1 digit comes from the
// k-mer's halves directly.  17 <= k <= 31 (every shift amount in 1 .. 31; other k-mer lengths keep the 64-bit form).
// Every function is plain integer arithmetic and compiles for the host too: tests/l1_lean_check.cc compares it window by window with a
// naive restatement (first base in the MSBs, mer_dna.hpp:46-63; reverse complement mer_dna.hpp:100-108) and with place_digit1_of /
// place_key of kg_device.hpp.
#pragma once
#include <stdint.h>
#include "kg_device.hpp"

#if defined(__HIPCC__)
#define KG_LEAN_FN __host__ __device__ __forceinline__
#else
#define KG_LEAN_FN inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KG_ALIGNBIT(hi, lo, s) __builtin_amdgcn_alignbit((hi), (lo), (s))          /* ((hi:lo) >> (s & 31)) low word */
#define KG_MULHI32(a, b) __umulhi((a), (b))
#define KG_BREV32(x) __brev(x)
#else
#define KG_BREV32(x) kg::lean_brev32_portable(x)
#define KG_ALIGNBIT(hi, lo, s) ((uint32_t)((((uint64_t)(hi) << 32) | (uint32_t)(lo)) >> ((s) & 31)))
#define KG_MULHI32(a, b) ((uint32_t)(((uint64_t)(a) * (uint64_t)(b)) >> 32))
#endif

namespace kg {

inline uint32_t lean_brev32_portable(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}

struct LeanGeom {                 // wave-uniform: from k, the strand mode and the table's placement budget, once per kernel
    uint32_t k;
    uint32_t kshift;              // 64 - 2k: 2 .. 30
    uint32_t top_hi;              // (2k - 2) - 32: where a new base enters the reverse complement's high word
    uint32_t n1m32;               // n1 - 32: bits of r1's high word (the k-mer's low n1 bits are what a level-1 item carries)
    uint32_t mask_lhi;            // low n1m32 bits
    uint32_t canonical;
};
KG_LEAN_FN LeanGeom lean_geom(uint32_t k, bool canonical, uint32_t n1) {
    LeanGeom g;
    g.k = k; g.kshift = 64 - 2 * k; g.top_hi = 2 * k - 34; g.n1m32 = n1 - 32;
    g.mask_lhi = (1u << g.n1m32) - 1;
    g.canonical = canonical ? 1u : 0u;
    return g;
}
// every shift amount in 1 .. 31, and r1 at least one whole word (other shapes keep the 64-bit form)
KG_LEAN_FN bool lean_applies(uint32_t k, uint32_t n1) { return k >= 17 && k <= 31 && n1 >= 32 && n1 < 2 * k; }

// bit (15 - j): window j of the lane (bases j .. j + k - 1 of its 48) holds no flagged base.  b0, b1, b2: the 16 flags (first base in
// bit 15) of bases 0-15, 16-31, 32-47.  A dilation: D[p] = OR of F[p .. p + k - 1]; later bases are lower bits, so "left".
KG_LEAN_FN uint32_t lean_valid16(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t k) {
    uint64_t D = ((uint64_t)(b0 & 0xFFFF) << 32) | ((uint64_t)(b1 & 0xFFFF) << 16) | (b2 & 0xFFFF);   // bit 47 = base 0
    D |= D << 1; D |= D << 2; D |= D << 4; D |= D << 8;                                                 // runs of 16
    D |= D << (k - 16);                                                                                  // ... of k (17 <= k <= 32)
    return (uint32_t)~(D >> 32) & 0xFFFFu;
}

struct LeanWin {
    uint32_t h1, h0, l1;          // bases 0-15, 16-31, 32-47 of the lane as 2-bit codes, first base in the MSBs
    uint32_t rc_hi, rc_lo;        // reverse complement of the current window's k-mer
};
KG_LEAN_FN void lean_step(LeanWin& w) {                                       // one base on
    w.h1 = KG_ALIGNBIT(w.h1, w.h0, 30);
    w.h0 = KG_ALIGNBIT(w.h0, w.l1, 30);
    w.l1 <<= 2;
}
KG_LEAN_FN void lean_fwd(const LeanWin& w, const LeanGeom& g, uint32_t& f_hi, uint32_t& f_lo) {   // the window's k-mer: the top 2k bits of h1:h0
    f_lo = KG_ALIGNBIT(w.h1, w.h0, g.kshift);
    f_hi = w.h1 >> g.kshift;
}
// the reverse complement rolls along: its last base leaves, the complement of the window's new last base enters at the top
KG_LEAN_FN void lean_rc_roll(LeanWin& w, const LeanGeom& g, uint32_t f_lo) {
    w.rc_lo = KG_ALIGNBIT(w.rc_hi, w.rc_lo, 2);
    w.rc_hi = (w.rc_hi >> 2) | ((3u ^ (f_lo & 3u)) << g.top_hi);
}
// canonical form (or the forward k-mer) and its level-1 digit (kg_device.hpp "placement": d1 = (H + g1(L)) mod p1, L = the low n1
// bits = the item, H = the bits above).  The k-mer's high word has 2k - 32 <= 30 bits.
KG_LEAN_FN uint32_t lean_digit1(const LeanWin& w, const LeanGeom& g, const Place& pl, uint32_t f_hi, uint32_t f_lo, uint32_t& key_hi, uint32_t& key_lo, bool& took_rc) {
    const uint64_t rc = ((uint64_t)w.rc_hi << 32) | w.rc_lo, fw = ((uint64_t)f_hi << 32) | f_lo;
    const bool rc_less = g.canonical && rc < fw;
    took_rc = rc_less;
    key_hi = rc_less ? w.rc_hi : f_hi;
    key_lo = rc_less ? w.rc_lo : f_lo;
    const uint32_t d = (key_hi >> g.n1m32) + place_g1(key_lo, key_hi & g.mask_lhi, pl);
    const uint32_t e = d - pl.p1;
    return e < d ? e : d;                                                    // (d < 2 p1: d - p1 wraps exactly when d < p1)
}

// reverse complement of a word of 16 bases (2-bit codes, first base in the MSBs): the tile's other strand is staged as
// rcode[v] = lean_revcomp16(code[511 - v]), so that the reverse complement of the window at tile position p is the window of rcode at
// position T - k - p (T = 8192 bases per tile)
KG_LEAN_FN uint32_t lean_revcomp16(uint32_t x) {
    uint32_t r = KG_BREV32(x);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return ~r;
}
// the k-mer whose window starts at base p of a staged stream (words of 16 bases; two words of padding behind the last)
KG_LEAN_FN uint64_t lean_kmer_at(const uint32_t* code, uint32_t p, uint32_t k) {
    const uint32_t w = p >> 4, o = p & 15;
    const uint32_t c0 = code[w], c1 = code[w + 1], c2 = code[w + 2];
    const uint32_t sh = (32 - 2 * o) & 31;                                     // (v_alignbit shifts by its amount mod 32: o = 0 is the words themselves)
    const uint32_t h1 = o ? KG_ALIGNBIT(c0, c1, sh) : c0, h0 = o ? KG_ALIGNBIT(c1, c2, sh) : c1;
    return (((uint64_t)h1 << 32) | h0) >> (64 - 2 * k);
}

// ---- what a staged entry of level 1 says about where its k-mer is read from (round 5) ----
// The tile is staged as words of 16 bases: code[LEAN_BLOCK + 2], then the flags (16 bits per word of codes: half as many words), then
// rcode[LEAN_BLOCK + 2] (the other strand's stream), one array U to the copy-out, counted from the word IN FRONT of code:
// U[1 + w] = code[w], U[1 + LEAN_RCW + w] = rcode[w].  The k-mer whose
// window starts q + 1 bases into a stream is read from the three words in front of / at / behind that base: with x = q + 16, word index
// W = x >> 4 (+ LEAN_RCW on the other strand's stream) and funnel shift 2 (~q & 15) = 2 ((x ^ 15) & 15) -- one number, entry = (x ^ 15) << 1
// (+ LEAN_RCW << 5): bits 5 .. 15 = W, bits 0 .. 4 = the shift (v_alignbit_b32 reads exactly those).  Window j of lane tid starts
// 16 tid + j bases into the forward stream (q = 16 tid + j - 1) and T - k - (16 tid + j) bases into the other strand's (q = u - 16 tid,
// u = T - 1 - k - j): 16 tid has no low nibble, so both entries are a constant of the window PLUS / MINUS 32 tid.
constexpr uint32_t LEAN_BLOCK = 512, LEAN_RCW = (LEAN_BLOCK + 2) + (LEAN_BLOCK + 2) / 2;
KG_LEAN_FN uint32_t lean_entry_fwd(int j) { return j ? 64u - 2u * (uint32_t)j : 0u; }                                    // + 32 tid
KG_LEAN_FN uint32_t lean_entry_rc(uint32_t u) { return ((LEAN_RCW + (u >> 4) + 1u) << 5) | (2u * (~u & 15u)); }          // - 32 tid;  u = T - 1 - k - j
// the k-mer an entry names, from U (what the copy-out does)
KG_LEAN_FN uint64_t lean_entry_kmer(const uint32_t* U, uint32_t entry, uint32_t k) {
    const uint32_t W = (entry >> 5) & 0x7FFu, sh = entry & 31u;
    const uint32_t h1 = KG_ALIGNBIT(U[W], U[W + 1], sh), h0 = KG_ALIGNBIT(U[W + 1], U[W + 2], sh);
    return (((uint64_t)h1 << 32) | h0) >> (64 - 2 * k);
}

}  // namespace kg
