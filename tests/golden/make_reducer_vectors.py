"""Regenerate tests/golden/reducer_vectors.json from the REAL reference code (oracle/_ref, built by `make -C oracle ref` from /root/reference).

What `kat comp` and `kat gcp` compute from two / one table(s), with as little restated as the image allows: the k-mer counts are what
Jellyfish's own parser + mer_iterator deliver for the reference's FASTQ pair (oracle/_ref/jf_ref kmers -- not the oracle's counts); the 13
counters and 4 spectra come out of the reference's own CompCounters::update{Hash1,Shared,Hash2}Counters, driven as Comp::compareSlice drives them
(src/comp.cc:387-484; oracle/_ref/kat_ref_parts compupdate); the matrices out of the reference's own SparseMatrix::inc / getMaxVal /
printMatrix (kat_ref_parts matrix) and, for gcp, its own gcCount (kat_ref_parts strutils).  The only lines restated here are the scalar ones
between them: scaleCounter + the clamp (src/comp.hpp:303-306, src/comp.cc:413-420, 466-471) and gcp's coverage position (src/gcp.cc:188-195).
Run in the build container:  python tests/golden/make_reducer_vectors.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import reducer_vectors as R  # noqa: E402

out = {"source": "oracle/_ref/jf_ref kmers -> oracle/_ref/kat_ref_parts compupdate | matrix | strutils (TGAC/KAT 2.4.2 + Jellyfish 2.2.0 sources, nothing stubbed)",
       "comp": {}, "gcp": {}}
refdata = os.path.join(ROOT, "tests", "golden", "refdata")
for case in R.COMP_CASES:
    counters, spectra, maxval, body = R.reference_comp(refdata, *case)
    out["comp"][R.tag(case)] = {"counters": counters, "spectra_sha256": R.digest(spectra), "maxval": maxval, "matrix_sha256": R.digest(body),
                                "matrix_sum": sum(int(x) for x in body.split())}
for case in R.GCP_CASES:
    maxval, body = R.reference_gcp(refdata, *case)
    out["gcp"][R.tag(case)] = {"maxval": maxval, "matrix_sha256": R.digest(body), "matrix_sum": sum(int(x) for x in body.split())}
json.dump(out, open(R.GOLDEN, "w"), indent=1, sort_keys=True)
print("wrote", R.GOLDEN, len(out["comp"]), "comp cases,", len(out["gcp"]), "gcp cases")
