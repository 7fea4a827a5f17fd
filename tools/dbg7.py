import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kat_amd
from kat_amd import synth
from oracle import koracle as ko
eng = kat_amd.Engine(0)
g = synth.genome(200000, seed=31)
reads = synth.reads(g, 0, 12000, seed=4)
buf = eng.alloc(reads.size + 32); buf.upload(reads)
for k, canonical, hint in ((27, True, 1 << 21), (27, False, 1 << 21), (31, True, 1 << 21), (16, True, 1 << 21), (32, False, 1 << 21), (27, True, 1 << 12)):
    gt = eng.table(k, canonical, size_hint=hint)
    geo = gt.geometry()
    gt.count_bases_device(buf.ptr, reads.size)
    ot = ko.Table(k, canonical).count_bases(reads)
    gk, gc = gt.export()
    okk, oc = ot.dump_sorted()
    u, cnt = np.unique(gk, return_counts=True)
    dup = int((cnt > 1).sum())
    extra = np.setdiff1d(u, okk)
    missing = np.setdiff1d(okk, u)
    print(k, canonical, hint, "geo p1 %x p2 %d R %d S %d" % (geo.p1, geo.p2, geo.n_regions, geo.region_slots), "gpu records", gk.size, "unique", u.size, "oracle", okk.size, "dup keys", dup, "extra", extra.size, "missing", missing.size,
          "sum gpu", int(gc.sum()), "sum oracle", int(oc.sum()), flush=True)
    if extra.size:
        print("   extra sample:", [ko.decode(int(x), k) for x in extra[:3]])
    if dup:
        d = u[cnt > 1][:3]
        print("   dup sample:", [ko.decode(int(x), k) for x in d])
print({k: v["launches"] for k, v in eng.profile().items() if v["launches"]})
