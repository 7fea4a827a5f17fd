"""The partitioned counter at the geometry `bench.py` runs -- not the tiny regions and rounds the hooks of
tests/test_gpu_partition.py force -- compared k-mer by k-mer with the direct kernel ON THE DEVICE.

The oracle cannot count 12 G k-mers in seconds, so the checker here is the product's own direct path (one atomic per k-mer,
kg_kernels.hpp: k_count), which the parity suite pins to the oracle table-dump by table-dump at every size the oracle reaches:
the same reads go once through `count_bases_device` as a whole (>= 32 M window starts: partition rounds with production
thresholds -- the 512 x 1024 grid of 9344-slot regions of BASELINE.json's config 4, packed slots, 5-byte level-2 items, two passes of
256 buckets) and once in slices below the threshold (the direct kernel) into a second table of the same grid.  katgpu_comp then joins
the two: every k-mer of one must be in the other with the same count (all mass on the matrix diagonal, shared == distinct,
shared totals == totals), and the histograms must agree bucket by bucket.  A bug that moves counts between k-mers while preserving
the totals -- invisible to tests/test_gpu_scale_properties.py -- cannot pass this."""
import os
import subprocess
import sys

import numpy as np
import pytest

import kat_amd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

L = 150


def _expected_distinct(inst, genome, k, err_ppm=2000):
    p_err = 1.0 - (1.0 - err_ppm / 1e6) ** k
    return int(min(inst, genome + inst * p_err * 1.05))


def _both_ways(engine, k, n_reads, genome, hint):
    g = engine.synth_genome(genome, seed=20260927)
    reads = engine.synth_reads(g, genome, first_read=0, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=2000, seed=1)
    g.free()
    engine.profile_reset()
    tp = engine.table(k, True, size_hint=hint)
    tp.count_bases_device(reads.ptr, reads.nbytes)
    prof = engine.profile()
    # (the direct kernel: at most what level 2's pool of block images had no room for, a millionth of the k-mers -- kg_l2_blocks.hpp)
    assert prof["part_apply"]["launches"] > 0 and prof["part_l1_scatter"]["launches"] > 0 and prof["count"]["units"] <= 1e-6 * n_reads * (L - k + 1), prof
    engine.profile_reset()
    td = engine.table(k, True, size_hint=hint, like=tp)
    rec = L + 1
    step = 100_000                                            # 15.1 M window starts per call: below the partitioned counter's threshold
    for a in range(0, n_reads, step):
        b = min(n_reads, a + step)
        td.count_bases_device(reads.ptr + a * rec, (b - a) * rec)
    prof = engine.profile()
    assert prof["part_apply"]["launches"] == 0 and prof["count"]["launches"] > 0, prof
    reads.free()
    return tp, td


def _same_tables(tp, td, inst):
    sp_, sd = tp.stats(), td.stats()
    assert sp_["total"] == sd["total"] == inst
    assert sp_["distinct"] == sd["distinct"]
    assert sp_["capacity"] == sd["capacity"]
    mx, cc, spec = kat_amd.comp(tp, td)
    assert int(cc[0]) == int(cc[1]) == inst                                          # totals
    assert int(cc[3]) == int(cc[4]) == int(cc[12]) == sp_["distinct"]                # distinct 1 == distinct 2 == shared
    assert int(cc[10]) == int(cc[11]) == inst                                        # shared totals: every instance, on both sides
    assert int(cc[6]) == int(cc[7]) == int(cc[8]) == int(cc[9]) == 0                 # nothing only in one
    off = mx.copy()
    np.fill_diagonal(off, 0)
    assert int(off.sum()) == 0, "k-mers whose counts differ between the partitioned and the direct counter: %d" % int(off.sum())
    assert int(np.trace(mx)) == sp_["distinct"]
    assert np.array_equal(spec[0], spec[1]) and np.array_equal(spec[2], spec[3])
    assert np.array_equal(tp.hist(high=100000), td.hist(high=100000))


def test_config4_full_size_partitioned_equals_direct(engine):
    """The whole of config 4's read library -- 300 M reads, 37.2 G k-mer instances, the three partition rounds of the bench step -- against
    the direct kernel (one global atomic per k-mer, ~3 s).  The reads are generated in slices (the table, its twin and the arena
    leave no room for 45 GB of them at once): a slice is counted both ways, then the next."""
    k, genome, n_reads, step = 27, 1_000_000_000, 300_000_000, 100_000_000
    hint = int(_expected_distinct(n_reads * (L - k + 1), genome, k) / 0.62) + (1 << 20)      # bench.py's hint1
    g = engine.synth_genome(genome, seed=20260927)
    tp = engine.table(k, True, size_hint=hint)
    geo = tp.geometry()
    assert (geo.p1, geo.p2) == (512, 1024) and geo.region_slots > 8192, (geo.p1, geo.p2, geo.region_slots)
    td = engine.table(k, True, size_hint=hint, like=tp)
    rec = L + 1
    for first in range(0, n_reads, step):
        reads = engine.synth_reads(g, genome, first_read=first, n_reads=step, read_len=L, frag_len=350, err_ppm=2000, seed=1)
        engine.profile_reset()
        tp.count_bases_device(reads.ptr, reads.nbytes)
        prof = engine.profile()
        # (the direct kernel takes what level 2's pool of block images had no room for: a tile whose waiting items complete 4.4 sigma more blocks
        # than the mean -- a few hundred k-mers of a round's 12.4 G, kg_l2_blocks.hpp; anything more means the partitioned path is not what ran)
        assert prof["part_apply"]["launches"] > 0 and prof["count"]["units"] <= 1e-6 * step * (L - k + 1), prof
        engine.profile_reset()
        for a in range(0, step, 100_000):                      # 15.1 M window starts per call: below the partitioned counter's threshold
            td.count_bases_device(reads.ptr + a * rec, min(100_000, step - a) * rec)
        prof = engine.profile()
        assert prof["part_apply"]["launches"] == 0 and prof["count"]["launches"] > 0, prof
        reads.free()
    g.free()
    _same_tables(tp, td, n_reads * (L - k + 1))
    tp.free(); td.free()
    engine.release_scratch()


def _count_config(engine, k, genome, n_reads):
    """A whole config of BASELINE.json counted on the device the way `bench.py` does (same generator, seeds and size hint)."""
    hint = int(_expected_distinct(n_reads * (L - k + 1), genome, k) / 0.62) + (1 << 20)
    g = engine.synth_genome(genome, seed=20260927)
    reads = engine.synth_reads(g, genome, first_read=0, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=2000, seed=1)
    g.free()
    engine.profile_reset()
    t = engine.table(k, True, size_hint=hint)
    t.count_bases_device(reads.ptr, reads.nbytes)
    reads.free()
    prof = engine.profile()
    assert prof["part_apply"]["launches"] > 0 and prof["count"]["launches"] == 0, prof
    engine.release_scratch()
    return t


def test_config2_full_size_hist_against_the_second_restatement(engine):
    """BASELINE.json's config 2 at its full size -- `kat hist`, 50 M reads, k = 27 -- : the table's multiset leaves the device, the sum of
    its counts is the number of windows, and the histogram is the one tests/independent.py (written from the user documentation)
    makes of the multiset."""
    from tests import independent as ind
    k, n_reads = 27, 50_000_000
    t = _count_config(engine, k, 100_000_000, n_reads)
    h = t.hist()
    keys, counts = t.export()
    assert keys.size == t.stats()["distinct"] and int(counts.sum(dtype=np.uint64)) == n_reads * (L - k + 1)
    assert np.array_equal(h, ind.hist(counts))
    t.free()


def test_config3_full_size_gcp_against_the_second_restatement(engine):
    """Config 3 at its full size -- `kat gcp`, 100 M reads, k = 27: the GC x coverage matrix against tests/independent.py from the
    exported multiset (GC counts by the popcount form, itself checked against the base-by-base one on a sample)."""
    from tests import independent as ind
    k, n_reads = 27, 100_000_000
    t = _count_config(engine, k, 200_000_000, n_reads)
    gm = t.gcp()
    keys, counts = t.export()
    assert int(counts.sum(dtype=np.uint64)) == n_reads * (L - k + 1)
    assert np.array_equal(ind.gc_count_popcount(keys[:2_000_000], k), ind.gc_count(keys[:2_000_000], k))
    assert np.array_equal(gm, ind.gcp(keys, counts, k, gc=ind.gc_count_popcount))
    t.free()


def test_config4_geometry_prefix_against_the_oracle():
    """The anchor that is not product against product: a 600 K-read prefix of config 4's library through the partitioned counter at
    config 4's table geometry, dump against the CPU oracle's (tests/bench_geometry_oracle_case.py)."""
    env = dict(os.environ, KATGPU_TESTING="1", KATGPU_PART_MIN_STARTS="0", KATGPU_P2_FAST="2")
    r = subprocess.run([sys.executable, os.path.join(HERE, "bench_geometry_oracle_case.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "bench geometry vs oracle ok" in r.stdout


def test_config5_geometry_partitioned_equals_direct(engine):
    """k = 31 (6-byte level-2 items, 21-bit in-slot counters), the per-GPU table of config 5."""
    k, genome, n_reads = 31, 1_000_000_000, 75_000_000
    hint = int(_expected_distinct(n_reads * (L - k + 1), genome, k) / 0.62) + (1 << 20)        # bench.py --workload comp-rr
    tp, td = _both_ways(engine, k, n_reads, genome, hint)
    _same_tables(tp, td, n_reads * (L - k + 1))
    tp.free(); td.free()
    engine.release_scratch()
