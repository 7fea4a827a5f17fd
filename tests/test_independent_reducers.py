"""The oracle's hist / gcp / comp against a second restatement written from the reference's documentation (tests/independent.py),
on the reference's own FASTQ pair and on messy synthetic input.  No GPU."""
import os

import numpy as np
import pytest

from tests import independent as ind


@pytest.mark.parametrize("k", [13, 17, 27])
def test_oracle_reducers_agree_with_the_documentation_restatement(ko, refdata, k):
    r1, r2 = (os.path.join(refdata, f) for f in ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq"))
    t1 = ko.Table(k, True).count_files([r1])
    t2 = ko.Table(k, True).count_files([r2])
    both = ko.Table(k, True).count_files([r1, r2])
    for t in (t1, both):
        keys, counts = t.dump_sorted()
        assert np.array_equal(t.hist(), ind.hist(counts))
        assert np.array_equal(t.hist(5, 300, 1), ind.hist(counts, 5, 300))
        assert np.array_equal(t.gcp(), ind.gcp(keys, counts, k))
    mx, cc, sp = ko.comp(t1, t2)
    assert np.array_equal(mx, ind.comp_matrix(*t1.dump_sorted(), *t2.dump_sorted()))
    mx, cc, sp = ko.comp(t1, t2, 1.0, 1.0, 31, 17)                  # small matrices: the catch-all row / column at work
    assert np.array_equal(mx, ind.comp_matrix(*t1.dump_sorted(), *t2.dump_sorted(), 31, 17))


def test_known_rows_of_the_survey(ko, refdata):
    """`kat hist -m27` of the reference's FASTQ pair starts 111200 11696 2063 737 361 ... -- from the multiset alone."""
    r = [os.path.join(refdata, f) for f in ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq")]
    keys, counts = ko.Table(27, True).count_files(r).dump_sorted()
    h = ind.hist(counts)
    assert [int(x) for x in h[:10]] == [111200, 11696, 2063, 737, 361, 128, 72, 45, 28, 5]
    g = ind.gcp(*ko.Table(17, True).count_files(r).dump_sorted(), 17)
    assert int(g.max()) == 21046 and g.shape == (17, 1001)             # `kat gcp -m17`: "# MaxVal:21046", "# Rows:17"
