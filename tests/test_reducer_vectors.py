"""The reducers -- Comp::compareSlice, Gcp::analyseSlice -- pinned to the reference's own classes (tests/golden/reducer_vectors.json, written by
tests/golden/make_reducer_vectors.py in the build container): k-mer counts from Jellyfish's parser + mer_iterator, counters and spectra from the
real CompCounters, matrices from the real SparseMatrix (and gcCount), the scalar lines between them restated in tests/reducer_vectors.py.  Held
to it: the oracle (CPU, here) and the HIP path through the C ABI (-m gpu); and where oracle/_ref exists, the golden file itself is re-derived."""
import json
import os

import numpy as np
import pytest

from tests import reducer_vectors as R

GOLD = json.load(open(R.GOLDEN))
REFDATA = os.path.join(R.ROOT, "tests", "golden", "refdata")
F1, F2 = os.path.join(REFDATA, R.R1), os.path.join(REFDATA, R.R2)


def spectra_text(sp):
    return "".join(" ".join(str(int(v)) for v in row) + "\n" for row in sp)


def check_comp(case, mx, cc, sp):
    g = GOLD["comp"][R.tag(case)]
    assert [int(x) for x in cc] == g["counters"], case
    assert R.digest(spectra_text(sp)) == g["spectra_sha256"], case
    assert int(mx.max()) == g["maxval"] and int(mx.sum()) == g["matrix_sum"], case
    assert R.digest(R.matrix_text(mx)) == g["matrix_sha256"], case


def check_gcp(case, m):
    g = GOLD["gcp"][R.tag(case)]
    assert int(m.max()) == g["maxval"] and int(m.sum()) == g["matrix_sum"], case
    assert R.digest(R.matrix_text(m)) == g["matrix_sha256"], case


@pytest.fixture(scope="module")
def ko():
    from oracle import koracle
    return koracle


@pytest.mark.parametrize("case", R.COMP_CASES, ids=R.tag)
def test_oracle_comp_is_the_reference_classes_result(ko, case):
    k, s1, s2, b1, b2 = case
    t1, t2 = ko.Table(k, True).count_files([F1]), ko.Table(k, True).count_files([F2])
    check_comp(case, *ko.comp(t1, t2, s1, s2, b1, b2))


@pytest.mark.parametrize("case", R.GCP_CASES, ids=R.tag)
def test_oracle_gcp_is_the_reference_classes_result(ko, case):
    k, scale, bins = case
    check_gcp(case, ko.Table(k, True).count_files([F1]).gcp(scale, bins))


@pytest.mark.skipif(not (os.access(R.JF_REF, os.X_OK) and os.access(R.KAT_REF, os.X_OK)), reason="oracle/_ref not built (no /root/reference)")
def test_golden_file_is_what_the_reference_code_says_today():
    case = R.COMP_CASES[4]
    counters, spectra, maxval, body = R.reference_comp(REFDATA, *case)
    g = GOLD["comp"][R.tag(case)]
    assert counters == g["counters"] and R.digest(spectra) == g["spectra_sha256"] and maxval == g["maxval"] and R.digest(body) == g["matrix_sha256"]
    gcase = R.GCP_CASES[4]
    maxval, body = R.reference_gcp(REFDATA, *gcase)
    assert maxval == GOLD["gcp"][R.tag(gcase)]["maxval"] and R.digest(body) == GOLD["gcp"][R.tag(gcase)]["matrix_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", R.COMP_CASES, ids=R.tag)
def test_hip_comp_is_the_reference_classes_result(case):
    import kat_amd
    eng = kat_amd.Engine(0)
    k, s1, s2, b1, b2 = case
    t1, t2 = eng.table(k, True).count_files([F1]), eng.table(k, True).count_files([F2])
    check_comp(case, *kat_amd.comp(t1, t2, s1, s2, b1, b2))


@pytest.mark.gpu
@pytest.mark.parametrize("case", R.GCP_CASES, ids=R.tag)
def test_hip_gcp_is_the_reference_classes_result(case):
    import kat_amd
    eng = kat_amd.Engine(0)
    k, scale, bins = case
    check_gcp(case, eng.table(k, True).count_files([F1]).gcp(scale, bins))
