"""Regenerate tests/golden/reference_vectors.json from the REAL reference code (oracle/_ref/jf_ref, built by `make -C oracle ref`
from /root/reference): for the reference's own test data, the sha256 / distinct / total of the "<kmer> <count>" lines its parser +
mer_iterator deliver.  Run in the build container:  python tests/golden/make_reference_vectors.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_oracle_vs_reference as T  # noqa: E402

refdata = os.path.join(ROOT, "tests", "golden", "refdata")
out = {"source": "oracle/_ref/jf_ref kmers <k> <canonical> <files>  (Jellyfish 2.2.0 parser + mer_iterator of TGAC/KAT 2.4.2's deps)", "kmers": {}}
for tag, paths, k, c in T.fixed_cases(refdata):
    b = T.ref_kmers(paths, k, c)
    out["kmers"][tag] = {"sha256": T.digest(b), "distinct": b.count(b"\n"), "total": sum(int(l.split()[1]) for l in b.splitlines())}
json.dump(out, open(T.GOLDEN, "w"), indent=1, sort_keys=True)
print("wrote", T.GOLDEN, len(out["kmers"]), "cases")
