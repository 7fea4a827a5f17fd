"""Run in a subprocess by tests/test_gpu_bench_geometry.py::test_config4_geometry_prefix_against_the_oracle with
KATGPU_PART_MIN_STARTS=0 (so that a 600 K-read prefix takes the partitioned counter) and the one-pass level 2 forced: the prefix of
config 4's read library goes into a table of config 4's size -- the 512 x 1024 grid of 9344-slot regions, packed 8-byte slots, 6-byte
level-1 items written as groups by the segmented level 1, 5-byte level-2 items, two passes of 256 buckets -- and the table's dump is
compared record by record with the CPU oracle's count of the same bytes.  This is the anchor of the bench geometry that is NOT a
comparison of the product with itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from oracle import koracle as ko  # noqa: E402

L, K = 150, 27


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
    genome = 1_000_000_000
    eng = kat_amd.Engine(0)
    g = eng.synth_genome(genome, seed=20260927)
    reads = eng.synth_reads(g, genome, first_read=0, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=2000, seed=1)
    g.free()
    p_err = 1.0 - (1.0 - 2000 / 1e6) ** K
    inst300 = 300_000_000 * (L - K + 1)
    hint = int(min(inst300, genome + inst300 * p_err * 1.05) / 0.62) + (1 << 20)           # bench.py's hint for config 4
    eng.profile_reset()
    t = eng.table(K, True, size_hint=hint)
    t.count_bases_device(reads.ptr, reads.nbytes)
    prof = eng.profile()
    geo = t.geometry()
    assert (geo.p1, geo.p2) == (512, 1024) and geo.region_slots > 8192, (geo.p1, geo.p2, geo.region_slots)
    assert prof["part_l1_scatter"]["launches"] > 0 and prof["part_l1_count"]["launches"] <= 1 and prof["part_l2"]["launches"] >= 2 and prof["part_apply"]["launches"] >= 2 and prof["count"]["units"] < 100000, prof     # (the direct kernel: a tail of a few windows at most)
    host = reads.download()
    reads.free()
    o = ko.Table(K, True).count_bases(host)
    st = t.stats()
    assert st["distinct"] == o.distinct and st["total"] == o.total == n_reads * (L - K + 1), (st, o.distinct, o.total)
    gk, gc = t.dump_sorted()
    ok_, oc = o.dump_sorted()
    assert np.array_equal(gk, ok_), "k-mers differ"
    assert np.array_equal(gc, oc), "counts differ at %d k-mers" % int((gc != oc).sum())
    assert np.array_equal(t.hist(), o.hist())
    print("bench geometry vs oracle ok: %d reads, %d distinct %d-mers, launches %s" % (n_reads, gk.size, K, {k: v["launches"] for k, v in prof.items() if v["launches"]}))


if __name__ == "__main__":
    main()
