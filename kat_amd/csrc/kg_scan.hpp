// kg_scan.hpp -- the record scan of FASTQ / FASTA on the device: raw file bytes in HBM -> the base stream the counter consumes.
//
// Replaces, for large plain files, what Jellyfish's mer_overlap_sequence_parser does on host cores (read_fastq / read_fasta,
// deps/jellyfish-2.2.0/include/jellyfish/mer_overlap_sequence_parser.hpp:189-289) and what kg_ingest.cpp's state machine does for
// every other input: the host only moves bytes (pread into pinned memory, one H2D copy), the device finds the lines, checks the record
// structure and compacts the sequence bytes.  A chunk of the file that begins at a line start (FASTQ: at a record start) becomes
//   FASTQ  for every record: its sequence line, then 'N'          (the reference puts one 'N' between records, :234)
//   FASTA  for every header line one 'N', for every other line its bytes without the newline     (:202, :254-260)
// -- the same k-mer windows, in the same order, as the host state machine's stream.  Anything the device cannot vouch for makes the
// chunk INVALID and the host state machine takes it from the chunk's start, so the stream is the streaming parser's whatever the file
// looks like: a '\r', an empty line, a FASTQ record that is not four lines ('@' / sequence / '+' / quality of the sequence's length),
// more lines than the line arrays hold.
//
// Passes (all streaming; a 1 GiB chunk is ~3 GB of traffic, a millisecond or two):
//   k_nl_count   newlines per 4 KiB tile                                   -> scan -> first line index of every tile
//   k_nl_write   NL[j] = position of the j-th newline
//   k_line_len   per line: what it contributes to the output (and the structure checks)   -> scan -> out_off[j]
//   k_emit       per byte: its place in the output, from its line
#pragma once
#include "kg_kernels.hpp"

namespace kg {

constexpr int SC_BLOCK = 256, SC_BYTES = 16, SC_TILE = SC_BLOCK * SC_BYTES;      // one 16-byte load per lane, 4 KiB per workgroup step
enum ScanType : uint32_t { SCAN_FASTA = 0, SCAN_FASTQ = 1 };
// flags[] (u64 each, device): what the host reads back
constexpr int SCF_BAD = 0, SCF_LINES = 1, SCF_OUT = 2, SCF_WORDS = 4;
constexpr uint64_t SCB_CR = 1, SCB_EMPTY_LINE = 2, SCB_STRUCT = 4, SCB_QLEN = 8, SCB_LINES = 16;   // why a chunk is invalid

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
// the 16 bytes at offset `off` of the chunk (raw is 16-byte aligned; bytes beyond n read as 0) and the mask of those equal to c
__device__ __forceinline__ void sc_load(const uint8_t* __restrict__ raw, uint64_t n, uint64_t off, uint32_t (&w)[4]) {
    if (off + SC_BYTES <= n) { const u32x4v v = *reinterpret_cast<const u32x4v*>(raw + off); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { const uint64_t i = off + q * 4 + b; x |= (i < n ? (uint32_t)raw[i] : 0u) << (8 * b); }
            w[q] = x;
        }
    }
}
__device__ __forceinline__ uint32_t sc_mask(const uint32_t (&w)[4], uint32_t c) {       // bit b: byte b == c
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = w[q] ^ (c * 0x01010101u);                                     // zero bytes where equal
#pragma unroll
        for (int b = 0; b < 4; ++b) m |= (((x >> (8 * b)) & 0xFF) == 0 ? 1u : 0u) << (q * 4 + b);
    }
    return m;
}
// exclusive prefix of v over the 256 lanes of the workgroup; *total = the sum.  s_w: 4 words of LDS.  Two barriers.
__device__ __forceinline__ uint32_t sc_block_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SC_BLOCK / 64; ++w) { const uint32_t x = s_w[w]; if ((uint32_t)w < wave) before += x; tot += x; }
    *total = tot;
    return before + inc - v;
}

// tile_cnt[t] = newlines in tile t; a '\r' anywhere makes the chunk the host's
static __global__ void __launch_bounds__(SC_BLOCK)
k_nl_count(const uint8_t* __restrict__ raw, uint64_t n, uint32_t n_tiles, uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ flags) {
    __shared__ uint32_t s_w[4];
    bool cr = false;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t w[4];
        sc_load(raw, n, (uint64_t)t * SC_TILE + threadIdx.x * SC_BYTES, w);
        cr = cr || sc_mask(w, '\r') != 0;
        uint32_t tot;
        (void)sc_block_scan(__popc(sc_mask(w, '\n')), s_w, &tot);
        if (threadIdx.x == 0) tile_cnt[t] = tot;
    }
    if (__any(cr) && (threadIdx.x & 63) == 0) atomicOr(&flags[SCF_BAD], (unsigned long long)SCB_CR);
}

// NL[j] = offset of the j-th newline of the chunk (tile_off: exclusive scan of tile_cnt)
static __global__ void __launch_bounds__(SC_BLOCK)
k_nl_write(const uint8_t* __restrict__ raw, uint64_t n, uint32_t n_tiles, const uint64_t* __restrict__ tile_off, uint32_t* __restrict__ NL, uint64_t cap_lines) {
    __shared__ uint32_t s_w[4];
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t w[4];
        const uint64_t off = (uint64_t)t * SC_TILE + threadIdx.x * SC_BYTES;
        sc_load(raw, n, off, w);
        uint32_t m = sc_mask(w, '\n'), tot;
        uint64_t at = tile_off[t] + sc_block_scan(__popc(m), s_w, &tot);
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            if (at < cap_lines) NL[at] = (uint32_t)(off + b);
            ++at;
        }
    }
}

// per line j: out_len[j] = bytes it puts into the output; tile_sum[j / 256] = their sum; the structure checks
template <uint32_t TYPE>
static __global__ void __launch_bounds__(SC_BLOCK)
k_line_len(const uint8_t* __restrict__ raw, const uint32_t* __restrict__ NL, uint64_t n_lines, uint32_t* __restrict__ out_len, uint32_t* __restrict__ tile_sum,
           unsigned long long* __restrict__ flags) {
    __shared__ uint32_t s_w[4];
    const uint64_t n_tiles = (n_lines + SC_BLOCK - 1) / SC_BLOCK;
    uint64_t bad = 0;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t j = t * SC_BLOCK + threadIdx.x;
        uint32_t len_out = 0;
        if (j < n_lines) {
            const uint32_t start = j ? NL[j - 1] + 1 : 0, end = NL[j], len = end - start;
            if (len == 0) bad |= SCB_EMPTY_LINE;                           // (the reference reads on past blank lines in ways a line count cannot follow)
            const uint8_t first = len ? raw[start] : 0;
            if (TYPE == SCAN_FASTQ) {
                const uint32_t phase = (uint32_t)(j & 3);
                if (phase == 0 && first != '@') bad |= SCB_STRUCT;
                if (phase == 1) { if (first == '+') bad |= SCB_STRUCT; len_out = len + 1; }       // the sequence line and the record's 'N'
                if (phase == 2 && first != '+') bad |= SCB_STRUCT;
                if (phase == 3 && len != NL[j - 2] - NL[j - 3] - 1) bad |= SCB_QLEN;               // qualities are skipped BY LENGTH (:274-289)
            } else {
                len_out = first == '>' ? 1 : len;                          // a header line: the 'N' between records
            }
            out_len[j] = len_out;
        }
        uint32_t tot;
        (void)sc_block_scan(len_out, s_w, &tot);
        if (threadIdx.x == 0) tile_sum[t] = tot;
    }
    if (bad) atomicOr(&flags[SCF_BAD], (unsigned long long)bad);
}

// out_len[j] -> out_off[j] in place (tile_off: exclusive scan of tile_sum)
static __global__ void __launch_bounds__(SC_BLOCK)
k_line_off(uint64_t n_lines, uint32_t* __restrict__ len_off, const uint64_t* __restrict__ tile_off) {
    __shared__ uint32_t s_w[4];
    const uint64_t n_tiles = (n_lines + SC_BLOCK - 1) / SC_BLOCK;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t j = t * SC_BLOCK + threadIdx.x;
        const uint32_t v = j < n_lines ? len_off[j] : 0;
        uint32_t tot;
        const uint32_t ex = sc_block_scan(v, s_w, &tot);
        if (j < n_lines) len_off[j] = (uint32_t)tile_off[t] + ex;
    }
}

// every byte to its place.  line_tile_off = the exclusive scan of k_nl_count's tile counts: the line the tile's first byte is in.
template <uint32_t TYPE>
static __global__ void __launch_bounds__(SC_BLOCK)
k_emit(const uint8_t* __restrict__ raw, uint64_t n, uint32_t n_tiles, const uint64_t* __restrict__ line_tile_off, const uint32_t* __restrict__ NL,
       const uint32_t* __restrict__ out_off, uint64_t n_lines, uint8_t* __restrict__ out) {
    __shared__ uint32_t s_w[4];
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t w[4];
        const uint64_t off = (uint64_t)t * SC_TILE + threadIdx.x * SC_BYTES;
        sc_load(raw, n, off, w);
        const uint32_t m = sc_mask(w, '\n');
        uint32_t tot;
        uint64_t L = line_tile_off[t] + sc_block_scan(__popc(m), s_w, &tot);     // the line of this lane's first byte
        if (off >= n || L >= n_lines) continue;                                  // (bytes after the last newline belong to no line: the host keeps chunks whole)
        uint32_t ls = L ? NL[L - 1] + 1 : 0, oo = out_off[L];
        bool header = TYPE == SCAN_FASTA && raw[ls] == '>';
        for (int b = 0; b < SC_BYTES; ++b) {
            const uint64_t i = off + b;
            if (i >= n || L >= n_lines) break;
            const uint8_t c = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
            const bool nl = (m >> b) & 1;
            if (TYPE == SCAN_FASTQ) {
                if ((L & 3) == 1) out[(uint64_t)oo + (uint32_t)(i - ls)] = nl ? (uint8_t)'N' : c;
            } else {
                if (header) { if (i == ls) out[oo] = 'N'; }
                else if (!nl) out[(uint64_t)oo + (uint32_t)(i - ls)] = c;
            }
            if (nl) {
                ++L;
                if (L < n_lines) { ls = (uint32_t)i + 1; oo = out_off[L]; header = TYPE == SCAN_FASTA && i + 1 < n && raw[i + 1] == '>'; }
            }
        }
    }
}

}  // namespace kg
