"""Run in a subprocess by tests/test_gpu_feeder.py with KATGPU_RING_MB / KATGPU_PART_MIN_STARTS set: host buffers and files go
through pinned staging into small device rings, every ring through the partitioned counter, and the table must equal the oracle."""
import os
import sys
import tempfile

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
    assert np.array_equal(gk, ok_) and np.array_equal(gc, oc), (what, "records")


def main():
    eng = kat_amd.Engine(0)
    g = synth.genome(300000, seed=11)
    reads = synth.reads(g, 0, 60000, seed=9)                      # 9 MB of bases: several rings of 2-3 MiB
    for k, canonical in ((27, True), (31, False), (16, True)):
        for hint in (1 << 12, 1 << 22):
            gt = eng.table(k, canonical, size_hint=hint).count_bases(reads)          # host buffer -> rings
            ot = ko.Table(k, canonical).count_bases(reads, threads=4)
            same(gt, ot, ("host", k, canonical, hint))
    # -g: the table may not grow; a hint with room for everything must not fail (the file path used to bound every staged
    # batch by its window starts and gave up with the table nearly empty), a tiny one must
    ot = ko.Table(27, True).count_bases(reads, threads=4)
    gt = eng.table(27, True, size_hint=int(ot.distinct / 0.5), disable_grow=True).count_bases(reads)
    same(gt, ot, "disable_grow, roomy")
    try:
        eng.table(27, True, size_hint=4096, disable_grow=True).count_bases(reads)
        raise AssertionError("a 4096-slot table took %d distinct k-mers" % ot.distinct)
    except kat_amd.KatGpuError as e:
        assert e.code == 7 and "Hash full" in e.message, e
    # files: two FASTQ files of one group + accumulate a FASTA into the same table
    with tempfile.TemporaryDirectory() as d:
        r = reads.reshape(-1, 151)[:, :150]
        paths = []
        for mate in (0, 1):
            p = os.path.join(d, "r%d.fq" % mate)
            with open(p, "wb") as f:
                for i, row in enumerate(r[mate::2]):
                    f.write(b"@r%d/%d\n" % (i, mate + 1) + row.tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
            paths.append(p)
        fa = os.path.join(d, "g.fa")
        with open(fa, "wb") as f:
            f.write(b">g\n")
            for i in range(0, g.size, 80):
                f.write(g[i:i + 80].tobytes() + b"\n")
        gt = eng.count(paths, 27, True)
        ot = ko.Table(27, True).count_files(paths)
        same(gt, ot, "files")
        gt.count_files([fa])
        ot.count_files([fa])
        same(gt, ot, "files, accumulated")
        assert np.array_equal(gt.hist(), ot.hist())
    prof = eng.profile()
    assert prof["part_apply"]["launches"] > 0, prof
    print("feeder cases ok", {k: v["launches"] for k, v in prof.items() if v["launches"]})


if __name__ == "__main__":
    main()
