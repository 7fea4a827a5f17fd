"""Synthetic workload of BASELINE.json's configs, host (numpy) edition.

Counter-based SplitMix64 so any shard can be produced independently; bit-identical to the device generator
(kat_amd/csrc/kg_kernels.hpp: k_synth_genome / k_synth_reads), which bench.py uses at full size.  This module is used
at parity scale: it writes the FASTA/FASTQ files the file-level tests feed to both the HIP engine and the oracle.

  genome : n i.i.d. uniform bases
  reads  : PE fragments of `frag_len`; read r = 2*pair + mate; mate 0 reads the fragment's first read_len bases
           forward, mate 1 the reverse complement of its last read_len bases; the fragment's strand is random;
           substitution errors at err_ppm / 1e6 per base.  Layout: read_len bases + 'N' per record.
"""
import numpy as np

U64 = np.uint64
_M32 = U64(0xFFFFFFFF)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _u(x):
    return np.asarray(x, dtype=U64)


def splitmix64(x):
    with np.errstate(over="ignore"):
        z = _u(x) + U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


def rng2(seed, idx):
    with np.errstate(over="ignore"):
        return splitmix64(splitmix64(_u(seed)) ^ (_u(idx) * U64(0xD1342543DE82EF95)))


def mulhi64(a, b):
    """High 64 bits of the 128-bit product (== __umul64hi)."""
    a = _u(a)
    b = _u(b)
    with np.errstate(over="ignore"):
        a0, a1 = a & _M32, a >> U64(32)
        b0, b1 = b & _M32, b >> U64(32)
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> U64(32)) + (p01 & _M32) + (p10 & _M32)
        return p11 + (p01 >> U64(32)) + (p10 >> U64(32)) + (mid >> U64(32))


def genome_codes(n, seed, start=0):
    i = np.arange(start, start + n, dtype=U64)
    w = rng2(seed, i >> U64(5))
    return ((w >> (U64(2) * (i & U64(31)))) & U64(3)).astype(np.uint8)


def genome(n, seed, start=0):
    """ASCII bases [start, start+n) of the genome."""
    return ACGT[genome_codes(n, seed, start)]


def reads(genome_ascii, first_read, n_reads, read_len=150, frag_len=350, err_ppm=2000, seed=1):
    """The base stream of reads [first_read, first_read+n_reads): uint8 array of n_reads*(read_len+1) bytes."""
    g = np.ascontiguousarray(genome_ascii, dtype=np.uint8)
    G = g.size
    assert G >= frag_len >= read_len and read_len < 1024
    r = np.arange(first_read, first_read + n_reads, dtype=U64)
    pair, mate = r >> U64(1), r & U64(1)
    u = rng2(seed, pair)
    start = mulhi64(u, U64(G - frag_len + 1))
    strand = splitmix64(u) >> U64(63)
    fwd = (strand ^ mate) == 0
    j = np.arange(read_len, dtype=U64)
    idx = np.where(fwd[:, None], start[:, None] + j[None, :], start[:, None] + U64(frag_len - 1) - j[None, :])
    gb = g[idx.astype(np.int64)].astype(np.uint32)
    code = ((gb >> 1) & 3) ^ ((gb >> 2) & 1)
    code = np.where(fwd[:, None], code, 3 - code)
    with np.errstate(over="ignore"):
        e = rng2(U64(seed) ^ U64(0x5EED5EED5EED5EED), r[:, None] * U64(1024) + j[None, :])
    thresh = (err_ppm << 32) // 1000000
    hit = (e & _M32) < U64(thresh)
    sub = (code + 1 + ((e >> U64(32)) % U64(3)).astype(np.uint32)) & 3
    code = np.where(hit, sub, code).astype(np.uint8)
    out = np.full((n_reads, read_len + 1), ord("N"), dtype=np.uint8)
    out[:, :read_len] = ACGT[code]
    return out.reshape(-1)


def write_fasta(path, genome_ascii, contig_len=1000000, width=80, name="contig"):
    """Assembly FASTA: contigs of contig_len, `width`-column lines."""
    g = np.ascontiguousarray(genome_ascii, dtype=np.uint8)
    with open(path, "wb") as f:
        for ci, s in enumerate(range(0, g.size, contig_len)):
            f.write((">%s%d\n" % (name, ci)).encode())
            c = g[s:s + contig_len]
            for o in range(0, c.size, width):
                f.write(c[o:o + width].tobytes())
                f.write(b"\n")


def write_fastq_pair(path1, path2, stream, read_len=150, first_pair=0):
    """Split an interleaved read stream (mate 0, mate 1, ...) into R1/R2 4-line FASTQ files."""
    recs = np.ascontiguousarray(stream, dtype=np.uint8).reshape(-1, read_len + 1)[:, :read_len]
    qual = b"I" * read_len
    with open(path1, "wb") as f1, open(path2, "wb") as f2:
        for i in range(recs.shape[0]):
            p = first_pair + i // 2
            (f1 if i % 2 == 0 else f2).write(b"@r%d/%d\n%s\n+\n%s\n" % (p, i % 2 + 1, recs[i].tobytes(), qual))


def stream_of_contigs(genome_ascii, contig_len=1000000):
    """Base stream the parser produces for write_fasta()'s file: contigs joined by 'N'."""
    g = np.ascontiguousarray(genome_ascii, dtype=np.uint8)
    parts = []
    for s in range(0, g.size, contig_len):
        if parts:
            parts.append(np.frombuffer(b"N", dtype=np.uint8))
        parts.append(g[s:s + contig_len])
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def assembly_stream(n_bases, seed, contig_len):
    """Exactly what Engine.synth_genome(n_bases, seed, contig_len) leaves in HBM: 'N' after every contig_len bases."""
    g = genome(n_bases, seed)
    if not contig_len:
        return g
    n_out = n_bases + n_bases // contig_len
    out = np.empty(n_out, np.uint8)
    x = np.arange(n_out, dtype=np.int64)
    c, j = x // (contig_len + 1), x % (contig_len + 1)
    sep = j == contig_len
    out[sep] = ord("N")
    out[~sep] = g[(c * contig_len + j)[~sep]]
    return out
