// Host check of kg_l1_lean.hpp: the 32-bit form of level 1's per-window arithmetic against a naive restatement (k-mer = bases j .. j+k-1,
// first base most significant, A=0 C=1 G=2 T=3: mer_dna.hpp:46-63; reverse complement: mer_dna.hpp:100-108; a window with a flagged base
// yields nothing: mer_iterator.hpp:61-89) and against the placement hash of kg_device.hpp (place_digit1_of / place_key, host-callable).
// Built with hipcc -x hip (so that kg_device.hpp is the real one) and run on the CPU; prints "l1 lean ok".
#include "../../kat_amd/csrc/kg_device.hpp"
#include "../../kat_amd/csrc/kg_l1_lean.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace kg;

static uint64_t naive_revcomp(uint64_t x, uint32_t k) {
    uint64_t r = 0;
    for (uint32_t i = 0; i < k; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
    return r;
}

int main() {
    std::mt19937_64 rng(20260927);
    uint64_t checked = 0;
    for (uint32_t k = 17; k <= 31; ++k)
        for (int canonical = 0; canonical < 2; ++canonical)
            for (uint32_t p1 : {1u, 2u, 7u, 256u, 512u, 773u, 1024u}) {
                const uint32_t n1 = place_n1(k, p1);
                if (!lean_applies(k, n1)) continue;                       // (k = 17 ... 20 with many level-1 digits: r1 shorter than a word)
                const Place pl = place_make(k, p1, n1, 10);
                const LeanGeom g = lean_geom(k, canonical != 0, n1);
                for (int rep = 0; rep < 400; ++rep) {
                    // 48 bases of a lane: codes and flags; flags sparse, dense or none
                    uint8_t code[48], bad[48];
                    const int mode = rep % 4;
                    for (int i = 0; i < 48; ++i) {
                        code[i] = (uint8_t)(rng() & 3);
                        bad[i] = mode == 0 ? 0 : mode == 1 ? (rng() % 37 == 0) : mode == 2 ? (rng() % 5 == 0) : (i == (int)(rng() % 48));
                        if (rep % 16 == 3) code[i] = 3;                      // poly-T: the all-ones k-mer at k = 32
                        if (rep % 16 == 7) code[i] = 0;
                    }
                    uint32_t cw[3] = {0, 0, 0}, bw[3] = {0, 0, 0};
                    for (int i = 0; i < 48; ++i) { cw[i / 16] = (cw[i / 16] << 2) | code[i]; bw[i / 16] = (bw[i / 16] << 1) | bad[i]; }
                    const uint32_t v16 = lean_valid16(bw[0], bw[1], bw[2], k);
                    LeanWin w{cw[0], cw[1], cw[2], 0, 0};
                    uint32_t f_hi, f_lo;
                    lean_fwd(w, g, f_hi, f_lo);
                    { const uint64_t rc0 = naive_revcomp(((uint64_t)f_hi << 32) | f_lo, k); w.rc_hi = (uint32_t)(rc0 >> 32); w.rc_lo = (uint32_t)rc0; }
                    for (int j = 0; j < 16; ++j) {
                        if (j) { lean_step(w); lean_fwd(w, g, f_hi, f_lo); lean_rc_roll(w, g, f_lo); }
                        // naive
                        uint64_t fw = 0; bool ok = true;
                        for (uint32_t i = 0; i < k; ++i) { fw = (fw << 2) | code[j + i]; ok = ok && !bad[j + i]; }
                        const uint64_t rc = naive_revcomp(fw, k);
                        const uint64_t key = canonical && rc < fw ? rc : fw;
                        const bool lean_ok = (v16 >> (15 - j)) & 1;
                        if (lean_ok != ok) { printf("valid mismatch k=%u j=%d\n", k, j); return 1; }
                        if ((((uint64_t)f_hi << 32) | f_lo) != fw) { printf("fwd mismatch k=%u j=%d\n", k, j); return 1; }
                        if ((((uint64_t)w.rc_hi << 32) | w.rc_lo) != rc) { printf("rc mismatch k=%u j=%d\n", k, j); return 1; }
                        uint32_t key_hi, key_lo;
                        bool took_rc;
                        const uint32_t d1 = lean_digit1(w, g, pl, f_hi, f_lo, key_hi, key_lo, took_rc);
                        if (took_rc != (canonical && rc < fw)) { printf("strand mismatch k=%u j=%d\n", k, j); return 1; }
                        if ((((uint64_t)key_hi << 32) | key_lo) != key) { printf("key mismatch k=%u j=%d\n", k, j); return 1; }
                        const uint32_t want = place_digit1_of(key, pl);
                        if (want >= p1) { printf("digit out of range k=%u p1=%u\n", k, p1); return 1; }
                        // what a level-1 item carries (the k-mer's low n1 bits) and the digit give the k-mer back
                        if (place_key(place_base1(want, pl), place_stage2(key & pl.m1, pl), pl) != key) { printf("inverse mismatch k=%u p1=%u\n", k, p1); return 1; }
                        if (d1 != want) { printf("digit mismatch k=%u p1=%u j=%d: %u vs %u\n", k, p1, j, d1, want); return 1; }
                        ++checked;
                    }
                }
            }
    // the other strand's stream: the reverse complement of the window at p is the window of rcode at T - k - p
    {
        constexpr uint32_t T = 8192, W = T / 16;
        static uint32_t code[W + 2], rcode[W + 2];
        static uint8_t base[T];
        for (uint32_t i = 0; i < T; ++i) base[i] = (uint8_t)(rng() & 3);
        for (uint32_t w = 0; w < W; ++w) { uint32_t x = 0; for (int i = 0; i < 16; ++i) x = (x << 2) | base[16 * w + i]; code[w] = x; }
        code[W] = code[W + 1] = 0; rcode[W] = rcode[W + 1] = 0;
        for (uint32_t w = 0; w < W; ++w) rcode[W - 1 - w] = lean_revcomp16(code[w]);
        for (uint32_t k = 17; k <= 31; ++k)
            for (uint32_t p = 0; p + k <= T; p += (p < 64 || p + k + 64 > T) ? 1 : 37) {
                uint64_t fw = 0;
                for (uint32_t i = 0; i < k; ++i) fw = (fw << 2) | base[p + i];
                if (lean_kmer_at(code, p, k) != fw) { printf("kmer_at mismatch k=%u p=%u\n", k, p); return 1; }
                if (lean_kmer_at(rcode, T - k - p, k) != naive_revcomp(fw, k)) { printf("rc stream mismatch k=%u p=%u\n", k, p); return 1; }
                ++checked;
            }
        // what sweep 2 parks per staged k-mer and the copy-out reads the k-mer by (lean_entry_*): every window of every lane, both strands
        static uint32_t U[1 + 3 * (W + 2)];
        for (uint32_t& x : U) x = 0xA5A5A5A5u;                                   // (the word in front of code, the flags between the streams: looked at by nobody)
        for (uint32_t w = 0; w < W + 2; ++w) { U[1 + w] = code[w]; U[1 + LEAN_RCW + w] = rcode[w]; }
        static_assert(LEAN_BLOCK == W && LEAN_RCW == (W + 2) + (W + 2) / 2, "tile");
        for (uint32_t k = 17; k <= 31; ++k)
            for (uint32_t tid = 0; tid < W; ++tid)
                for (int j = 0; j < 16; ++j) {
                    const uint32_t p = 16 * tid + (uint32_t)j;
                    if (p + k > T) continue;
                    uint64_t fw = 0;
                    for (uint32_t i = 0; i < k; ++i) fw = (fw << 2) | base[p + i];
                    const uint32_t ef = (32u * tid + lean_entry_fwd(j)) & 0xFFFFu, er = (lean_entry_rc(T - 1 - k - (uint32_t)j) - 32u * tid) & 0xFFFFu;
                    if (lean_entry_kmer(U, ef, k) != fw) { printf("entry (forward) mismatch k=%u tid=%u j=%d\n", k, tid, j); return 1; }
                    if (lean_entry_kmer(U, er, k) != naive_revcomp(fw, k)) { printf("entry (other strand) mismatch k=%u tid=%u j=%d\n", k, tid, j); return 1; }
                    // and through the arithmetic sweep 2 uses (one expression for both strands)
                    for (uint32_t strand = 0; strand < 2; ++strand) {
                        const uint32_t base_f = 32u * tid, cj = lean_entry_fwd(j), sj = lean_entry_rc(T - 1 - k - (uint32_t)j);
                        const uint32_t e = (base_f + cj + strand * (sj - cj - 2u * base_f)) & 0xFFFFu;
                        if (e != (strand ? er : ef)) { printf("entry arithmetic mismatch k=%u tid=%u j=%d strand=%u\n", k, tid, j, strand); return 1; }
                    }
                    ++checked;
                }
    }
    printf("l1 lean ok: %llu windows\n", (unsigned long long)checked);
    return 0;
}
