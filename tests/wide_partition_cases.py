"""Run in a subprocess by tests/test_gpu_wide.py with the KATGPU_* test hooks set in the environment: every device-resident count of a
k > 32 table goes through the wide partitioned counter (kg_partition_wide.hpp: tiny regions, tiny rounds) and must equal the wide oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import synth  # noqa: E402
from oracle import koracle as ko  # noqa: E402


def same(gt, ot, what):
    got, want = gt.dump_sorted_wide(), ot.dump_sorted()
    assert got[0].size == want[0].size, (what, "distinct", got[0].size, want[0].size)
    for a, b, name in zip(got, want, ("hi", "lo", "counts")):
        assert np.array_equal(a, b), (what, name, int((a != b).sum()))
    st = gt.stats()
    assert st["distinct"] == ot.distinct and st["total"] == ot.total, (what, st, ot.distinct, ot.total)


def main():
    eng = kat_amd.Engine(0)
    g = synth.genome(200000, seed=31)
    reads = synth.reads(g, 0, 12000, seed=4)                    # 1.8 MB
    messy = np.random.default_rng(7).choice(np.frombuffer(b"ACGTACGTACGTN acgt", np.uint8), size=700001)
    polya = np.concatenate([np.frombuffer(b"A" * 40000 + b"N" + b"T" * 5000 + b"N", np.uint8), reads[:300000]])
    n_cases = 0
    for name, stream in (("reads", reads), ("messy", messy), ("polyA", polya), ("short", reads[:9000]), ("tiny", reads[:70])):
        buf = eng.alloc(stream.size + 32)
        buf.upload(stream)
        for k, canonical in ((33, True), (45, False), (63, True)):
            for hint in (1 << 12, 1 << 21):                     # tiny hint: regrows + spills on the way; roomy hint: clean rounds
                gt = eng.table(k, canonical, size_hint=hint)
                gt.count_bases_device(buf.ptr, stream.size)
                ot = ko.WideTable(k, canonical).count_bases(stream)
                same(gt, ot, (name, k, canonical, hint))
                assert np.array_equal(gt.hist(), ot.hist()) and np.array_equal(gt.gcp(), ot.gcp())
                gt.free()
                n_cases += 1
        # a second call accumulates into the regions the first one filled; an unaligned stream takes the direct path for its head
        a = eng.table(41, True, size_hint=1 << 20)
        a.count_bases_device(buf.ptr, stream.size)
        a.count_bases_device(buf.ptr, stream.size // 2)
        oa = ko.WideTable(41, True).count_bases(stream).count_bases(stream[: stream.size // 2])
        same(a, oa, (name, "accumulate"))
        a.free()
        buf.free()
    prof = eng.profile()
    assert prof["part_l1_count"]["launches"] > 0 and prof["part_l2"]["launches"] > 0 and prof["part_apply"]["launches"] > 0, prof
    print("wide partition cases ok:", n_cases, {k: v["launches"] for k, v in prof.items() if v["launches"]})


if __name__ == "__main__":
    main()
