import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kat_amd
from kat_amd import synth
eng = kat_amd.Engine(0)
g = synth.genome(150000, seed=17)
a = synth.reads(g, 0, 16000, seed=1)
b = np.concatenate([synth.stream_of_contigs(g[:90000], 30000), synth.reads(g, 40000, 3000, seed=9, err_ppm=20000)])
for k, c1, c2 in ((27, True, True), (21, False, False), (21, True, False), (21, False, True), (32, False, False)):
    for hint1, hint2 in ((1 << 21, 1 << 19), (1 << 21, 1 << 22), (1 << 15, 1 << 14)):
        try:
            t1 = eng.table(k, c1, size_hint=hint1).count_bases(a)
            g1 = t1.geometry()
            t2 = eng.table(k, c2, size_hint=hint2, like=t1)
            g2 = t2.geometry()
            print(k, c1, c2, hint1, hint2, "t1", g1.n_regions, g1.region_slots, g1.p1, g1.p2, "t2", g2.n_regions, g2.region_slots, g2.p1, g2.p2, flush=True)
            t2.count_bases(b)
            g2 = t2.geometry()
            print("   after: t2", g2.n_regions, g2.region_slots, g2.p1, g2.p2, t2.stats(), flush=True)
        except Exception as e:
            print("FAIL", k, c1, c2, hint1, hint2, e, flush=True)
