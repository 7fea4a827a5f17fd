"""Run in a subprocess by tests/test_gpu_comp_forms.py with KATGPU_* hooks in the environment: `kat comp` through every form of
the comparison kernels -- the fused join (either table resident), the two-pass join with and without the seen bits, the probe
form; packed and KV12 slots -- must equal the oracle (oracle/koracle.c: Comp::compare restated)."""
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import synth  # noqa: E402
from oracle import koracle as ko  # noqa: E402


def main():
    eng = kat_amd.Engine(0)
    g = synth.genome(150000, seed=17)
    a = synth.reads(g, 0, 16000, seed=1)
    b = np.concatenate([synth.stream_of_contigs(g[:90000], 30000), synth.reads(g, 40000, 3000, seed=9, err_ppm=20000)])
    n = 0
    for k, c1, c2 in ((27, True, True), (21, True, True), (15, True, True), (21, False, False), (21, True, False), (21, False, True), (31, True, True)):
        o = {"a": ko.Table(k, c1).count_bases(a), "b": ko.Table(k, c2).count_bases(b)}
        for first, second in (("a", "b"), ("b", "a")):                     # the bigger table first, then second: both residencies of the fused join
            s1, s2 = (a, b) if first == "a" else (b, a)
            cc1, cc2 = (c1, c2) if first == "a" else (c2, c1)
            o1, o2 = (o["a"], o["b"]) if first == "a" else (ko.Table(k, cc1).count_bases(s1), ko.Table(k, cc2).count_bases(s2))
            hint1, hint2 = ((1 << 21, 1 << 19) if first == "a" else (1 << 20, 1 << 21))
            t1 = eng.table(k, cc1, size_hint=hint1).count_bases(s1)
            t2 = eng.table(k, cc2, size_hint=hint2, like=t1).count_bases(s2)
            # (> 64 unscaled bins both ways: the spectra fold into the tile; fewer, or scaled: they do not)
            for bins, scale in (((201, 101), 1.0), ((40, 50), 1.0), ((1001, 1001), 0.5)):
                want = ko.comp(o1, o2, scale, scale, *bins)
                got = kat_amd.comp(t1, t2, scale, scale, *bins)
                for gg, ww, name in zip(got, want, ("main", "counters", "spectra")):
                    assert np.array_equal(gg, ww), (k, cc1, cc2, first, bins, scale, name)
                n += 1
            t1.free(); t2.free()
    # exactness beyond the slot counter (packed: 20-odd bits; KV12: 32) through every form
    for hint in (1 << 21, 1 << 15):
        t1 = eng.table(21, True, size_hint=hint)
        t2 = eng.table(21, True, size_hint=hint, like=t1)
        o1, o2 = ko.Table(21, True), ko.Table(21, True)
        for key, c in ((5, (3 << 32) + 7), (77, 12), (1234567, 1), (999, (1 << 23) + 5), (4242, 1 << 22)):
            key = ko.canonical(key, 21)
            t1.merge_host([key], [c]); o1.add(key, c)
            t2.merge_host([key], [c + (1 << 33)]); o2.add(key, c + (1 << 33))
        got, want = kat_amd.comp(t1, t2), ko.comp(o1, o2)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), hint
        assert np.array_equal(t1.hist(), o1.hist()) and np.array_equal(t2.gcp(), o2.gcp())
        k1, c1_ = t1.dump_sorted()
        ok1, oc1 = o1.dump_sorted()
        assert np.array_equal(k1, ok1) and np.array_equal(c1_, oc1)
        n += 1
    prof = eng.profile()
    print("comp cases ok:", n, {k: v["launches"] for k, v in prof.items() if v["launches"]})


if __name__ == "__main__":
    main()
