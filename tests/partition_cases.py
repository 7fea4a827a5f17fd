"""Run in a subprocess by tests/test_gpu_partition.py with the KATGPU_* test hooks set in the environment: every
device-resident count goes through the partitioned counter (tiny regions, tiny rounds), and must equal the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import synth  # noqa: E402
from oracle import koracle as ko  # noqa: E402


def same(gt, ot, what):
    gk, gc = gt.dump_sorted()
    ok_, oc = ot.dump_sorted()
    assert gk.size == ok_.size, (what, "distinct", gk.size, ok_.size)
    assert np.array_equal(gk, ok_), (what, "keys")
    assert np.array_equal(gc, oc), (what, "counts", int((gc != oc).sum()))
    st = gt.stats()
    assert st["distinct"] == ot.distinct and st["total"] == ot.total, (what, st, ot.distinct, ot.total)


def main():
    eng = kat_amd.Engine(0)
    g = synth.genome(200000, seed=31)
    reads = synth.reads(g, 0, 12000, seed=4)                    # 1.8 MB, 1.49 M k-mers at k=27
    messy = np.random.default_rng(7).choice(np.frombuffer(b"ACGTACGTACGTN acgt", np.uint8), size=700001)
    polya = np.concatenate([np.frombuffer(b"A" * 40000 + b"N" + b"T" * 5000 + b"N", np.uint8), reads[:300000]])
    # one k-mer 2.3 M times: more than half the range of a packed slot's counter (20 - 22 bits at k = 27 in these tables): the excess
    # lives in the side table, whether it got there through the apply kernel's sweep, the direct path's hand-over or a regrow
    polya5m = np.concatenate([np.full(2_300_000, ord("A"), np.uint8), np.frombuffer(b"N", np.uint8), reads[:100000]])
    n_cases = 0
    for name, stream in (("reads", reads), ("messy", messy), ("polyA", polya), ("polyA2M", polya5m), ("short", reads[:5000]), ("tiny", reads[:40])):
        buf = eng.alloc(stream.size + 32)
        buf.upload(stream)
        for k, canonical in ((27, True), (31, False), (32, False), (15, True)):
            for hint in (1 << 12, 1 << 21):                     # tiny hint: regrows + spills on the way; roomy hint: clean rounds
                gt = eng.table(k, canonical, size_hint=hint)
                gt.count_bases_device(buf.ptr, stream.size)
                ot = ko.Table(k, canonical).count_bases(stream)
                same(gt, ot, (name, k, canonical, hint))
                assert np.array_equal(gt.hist(), ot.hist()) and np.array_equal(gt.gcp(), ot.gcp())
                n_cases += 1
        # accumulate a second call into a table (rounds must add to existing regions) and compare comp
        a = eng.table(27, True, size_hint=1 << 20)
        a.count_bases_device(buf.ptr, stream.size)
        a.count_bases_device(buf.ptr, stream.size // 2)
        oa = ko.Table(27, True).count_bases(stream).count_bases(stream[: stream.size // 2])
        same(a, oa, (name, "accumulate"))
        b = eng.table(27, True).count_bases(synth.stream_of_contigs(g, 20000))
        ob = ko.Table(27, True).count_bases(synth.stream_of_contigs(g, 20000))
        mx, cc, sp = kat_amd.comp(a, b)
        omx, occ, osp = ko.comp(oa, ob)
        assert np.array_equal(mx, omx) and np.array_equal(cc, occ) and np.array_equal(sp, osp), (name, "comp")
        buf.free()
    prof = eng.profile()
    assert (prof["part_l1_count"]["launches"] > 0 or os.environ.get("KATGPU_L1_FAST") in ("1", "2", None)) and prof["part_apply"]["launches"] > 0, prof
    print("partition cases ok:", n_cases, {k: v["launches"] for k, v in prof.items() if v["launches"]})


if __name__ == "__main__":
    main()
