"""Host edition of the synthetic workload generator: determinism, shard independence, statistics."""
import numpy as np

from kat_amd import synth
from tests import naive


def test_mulhi64_and_rng_reference_values():
    assert int(synth.mulhi64(2**63, 2)) == 1 and int(synth.mulhi64(2**64 - 1, 2**64 - 1)) == 2**64 - 2
    assert int(synth.mulhi64(12345678901234567, 98765432109876543)) == (12345678901234567 * 98765432109876543) >> 64
    # SplitMix64 reference stream for seed 0 (public test vector): first output 0xE220A8397B1DCDAF
    assert int(synth.splitmix64(0)) == 0xE220A8397B1DCDAF


def test_genome_is_shard_independent_and_uniform():
    g = synth.genome(100000, seed=5)
    assert np.array_equal(g[777:5000], synth.genome(5000 - 777, seed=5, start=777))
    assert set(np.unique(g)) == set(b"ACGT")
    assert all(abs((g == c).mean() - 0.25) < 0.01 for c in b"ACGT")
    assert not np.array_equal(g, synth.genome(100000, seed=6))


def test_reads_are_shard_independent_pairs_with_errors():
    G = 50000
    g = synth.genome(G, seed=1)
    a = synth.reads(g, 0, 2000, seed=4)
    assert np.array_equal(a[1000 * 151:], synth.reads(g, 1000, 1000, seed=4))
    recs = a.reshape(-1, 151)
    assert (recs[:, 150] == ord("N")).all()
    gs = g.tobytes().decode()
    clean = synth.reads(g, 0, 2000, seed=4, err_ppm=0).reshape(-1, 151)
    for i in (0, 1, 2, 3, 998, 999):
        r = clean[i, :150].tobytes().decode()
        assert r in gs or naive.revcomp(r) in gs
    # mates of a pair come from opposite strands, 350 bp apart
    r0, r1 = clean[0, :150].tobytes().decode(), clean[1, :150].tobytes().decode()
    f0, f1 = (r0 in gs), (r1 in gs)
    assert f0 != f1
    p0 = gs.find(r0 if f0 else naive.revcomp(r0))
    p1 = gs.find(r1 if f1 else naive.revcomp(r1))
    assert abs(p1 - p0) == 200
    err = (recs[:, :150] != clean[:, :150]).mean()
    assert 0.001 < err < 0.003                      # 0.2 % substitutions


def test_fasta_fastq_writers_roundtrip(ko, tmp_path):
    g = synth.genome(30000, seed=3)
    synth.write_fasta(str(tmp_path / "asm.fa"), g, contig_len=7000, width=80)
    assert ko.parse_file(str(tmp_path / "asm.fa")).tobytes() == synth.stream_of_contigs(g, 7000).tobytes()
    s = synth.reads(g, 0, 200, seed=2)
    synth.write_fastq_pair(str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), s)
    recs = s.reshape(-1, 151)[:, :150]
    assert ko.parse_file(str(tmp_path / "r1.fq")).tobytes() == b"N".join(r.tobytes() for r in recs[0::2])
    assert ko.parse_file(str(tmp_path / "r2.fq")).tobytes() == b"N".join(r.tobytes() for r in recs[1::2])
    a = synth.assembly_stream(28000, 3, 7000)                    # device layout: an 'N' after EVERY contig, the last included
    assert a.tobytes() == synth.stream_of_contigs(g[:28000], 7000).tobytes() + b"N"
