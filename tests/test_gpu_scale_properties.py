"""Larger inputs (generated and counted on the device) checked through size-independent properties."""
import numpy as np
import pytest

import kat_amd
from kat_amd import dist as kdist
from kat_amd import synth

pytestmark = pytest.mark.gpu


def test_checksums_at_scale(engine):
    """20 M reads (2.5 G k-mer instances): sum of counts == number of windows; hist / gcp / comp marginals agree."""
    G, n_reads, k, L = 50_000_000, 20_000_000, 27, 150
    g = engine.synth_genome(G, seed=20260927)
    reads = engine.synth_reads(g, G, 0, n_reads, seed=1)
    asm = engine.synth_genome(G, seed=20260927, contig_len=1_000_000)
    t1 = engine.table(k, True, size_hint=300_000_000).count_bases(reads)
    t2 = engine.table(k, True, size_hint=90_000_000).count_bases(asm)
    s1, s2 = t1.stats(), t2.stats()
    assert s1["total"] == n_reads * (L - k + 1)
    assert s2["total"] == G - (G // 1_000_000) * (k - 1)
    assert 0.99 * G < s2["distinct"] <= s2["total"]                      # random 50 Mbp genome: nearly all 27-mers unique
    h = t1.hist()
    assert int(h.sum()) == s1["distinct"]
    assert int((h * np.arange(1, h.size + 1, dtype=np.uint64))[:-1].sum()) <= s1["total"]
    gm = t1.gcp()
    assert 0 <= s1["distinct"] - int(gm.sum()) < 64                      # only all-G/C 27-mers (GC == k, ~2^-27 of them) are dropped
    mx, cc, sp = kat_amd.comp(t1, t2)
    assert int(cc[0]) == s1["total"] and int(cc[1]) == s2["total"] and int(cc[3]) == s1["distinct"] and int(cc[4]) == s2["distinct"]
    assert int(cc[8]) + int(cc[12]) == int(cc[3]) and int(cc[9]) + int(cc[12]) == int(cc[4])     # only + shared == distinct
    assert int(mx.sum()) == int(cc[3]) + int(cc[9])                      # every hash-1 k-mer once + hash-2-only k-mers
    assert np.array_equal(mx.sum(axis=1)[1:], sp[0][1:])                 # row marginals == spectrum 1 (scale 1.0)
    assert int(sp[0].sum()) == int(cc[3]) and int(sp[1].sum()) == int(cc[4]) and int(sp[2].sum()) == int(cc[12])
    # error k-mers: ~ (1 - 0.998^27) of the instances are singletons absent from the assembly
    assert 0.03 < int(mx[1, 0]) / s1["total"] < 0.07
    # idempotence of the reducers on an immutable table
    assert np.array_equal(h, t1.hist())
    # the three reducers against the documentation-derived restatement (tests/independent.py), from the exported multisets alone
    from tests import independent as ind
    k1, c1 = t1.export()
    k2, c2 = t2.export()
    assert np.array_equal(h, ind.hist(c1)) and np.array_equal(t2.hist(), ind.hist(c2))
    assert np.array_equal(gm, ind.gcp(k1, c1, k))
    assert np.array_equal(mx, ind.comp_matrix(k1, c1, k2, c2))


def test_count_is_order_and_batch_independent(engine):
    G, k = 2_000_000, 31
    g = engine.synth_genome(G, seed=3)
    reads = engine.synth_reads(g, G, 0, 400_000, seed=2)
    whole = engine.table(k, True, size_hint=1 << 22).count_bases(reads)
    parts = engine.table(k, True, size_hint=1 << 12)                     # many regrows, four calls, reverse order
    rec = 151
    cuts = [0, 100_000, 200_001, 333_333, 400_000]
    for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):
        parts.count_bases_device(reads.ptr + a * rec, (b - a) * rec)
    ka, ca = whole.dump_sorted()
    kb, cb = parts.dump_sorted()
    assert np.array_equal(ka, kb) and np.array_equal(ca, cb)


def test_device_owner_matches_host_mirror(engine):
    g = synth.genome(60000, seed=8)
    t = engine.table(23, False).count_bases(synth.reads(g, 0, 3000, seed=6))
    for n_parts in (2, 8):
        sizes = t.partition_sizes(n_parts)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        total = int(sizes.sum())
        dk, dc = engine.alloc(total * 8), engine.alloc(total * 8)
        t.partition(n_parts, offsets, dk.ptr, dc.ptr)
        keys = dk.download(np.uint64)
        part_of = np.repeat(np.arange(n_parts), sizes.astype(np.int64))
        assert np.array_equal(kdist.owner_of(keys, 23, n_parts), part_of)
