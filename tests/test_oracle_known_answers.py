"""Pins the CPU oracle (oracle/koracle.c) to every known answer available for the path.

(1) Known answers held by the reference's OWN tests (TGAC/KAT tests/):
      check_jellyfish.cc:38-60   .jf header fields of tests/data/ecoli.header.jf27
      check_jellyfish.cc:62-91   getCount on that hash: 3/1/1/1 and canonical 3/1/0/0
      check_jellyfish.cc:93-116  1889 records
      check_compcounters.cc:30-62 CompCounters arithmetic
(2) Known answers quoted in SURVEY.md 8(c).  PROVENANCE: those were recorded by the survey stage from a hand-built
    reference binary; the reference cannot be rebuilt in this image (it needs an autoconf-generated config.h), so they
    cannot be regenerated here.  They are the only end-to-end hist/gcp/comp numbers available for the reference's
    own tests/data inputs (its CLI tests check exit codes only: tests/test_hist.sh, test_gcp.sh, test_comp.sh).
"""
import json
import os

import numpy as np


def test_jf_header_and_queries(ko, refdata):
    t = ko.Table.from_jf(os.path.join(refdata, "ecoli.header.jf27"))
    hdr = json.loads(t.header_json)
    assert hdr["key_len"] == 54 and hdr["val_len"] == 7 and hdr["counter_len"] == 4        # check_jellyfish.cc:50-53
    assert hdr["max_reprobe"] == 126 and hdr["size"] == 131072 and hdr["format"] == "binary/sorted"
    assert 9 + len(t.header_json) <= 1368 and (1368 % hdr["alignment"]) == 0               # offset 1368 (:54)
    assert t.k == 27 and t.n_records == 1889 and t.distinct == 1889                        # :115
    q = ["AGCTTTTCATTCTGACTGCAACGGGCA", "GCATAGCGCACAGACAGATAAAAATTA", "AATGAAAAAGGCGAACTGGTGGTGCTT", "CTCACCAATGTACATGGCCTTAATCTG"]
    assert [t.get(ko.encode(s)) for s in q] == [3, 1, 1, 1]                               # :82-85
    assert [t.get(ko.canonical(ko.encode(s), 27)) for s in q] == [3, 1, 0, 0]             # :87-90


def test_kmer_codec(ko):
    s = "AGCTTTTCATTCTGACTGCAACGGGCA"
    key = ko.encode(s)
    assert ko.decode(key, 27) == s
    assert ko.decode(ko.revcomp(key, 27), 27) == "TGCCCGTTGCAGTCAGAATGAAAAGCT"
    assert ko.encode("A" * 32) == 0 and ko.encode("T" * 32) == 2**64 - 1 and ko.encode("ACGT") == 0b00011011
    assert ko.canonical(ko.encode("T" * 32), 32) == 0


def test_compcounters_arithmetic(ko):
    """check_compcounters.cc:30-62: two workers each do updateHash1Counters(10,2), (20,4), updateHash2Counters(0,3);
    merged: hash1_distinct 4, hash1_total 60.  Restated through ko.comp on tables holding those counts."""
    t1, t2 = ko.Table(5, False), ko.Table(5, False)
    for key, c1, c2 in ((1, 10, 2), (2, 20, 4), (3, 10, 2), (4, 20, 4)):
        t1.add(key, c1)
        t2.add(key, c2)
    t2.add(100, 3)
    t2.add(101, 3)
    mx, cc, sp = ko.comp(t1, t2)
    assert int(cc[3]) == 4 and int(cc[0]) == 60                  # hash1_distinct, hash1_total
    assert int(cc[4]) == 6 and int(cc[1]) == 18 and int(cc[9]) == 2 and int(cc[7]) == 6
    assert int(mx[10, 2]) == 2 and int(mx[20, 4]) == 2 and int(mx[0, 3]) == 2


def _rows(h, first=1):
    return {i + first: int(v) for i, v in enumerate(h) if v}


def test_survey_known_answers_hist(ko, refdata):
    r12 = [os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")]
    h = ko.Table(27, True).count_files(r12).hist()
    assert _rows(h) == {1: 111200, 2: 11696, 3: 2063, 4: 737, 5: 361, 6: 128, 7: 72, 8: 45, 9: 28, 10: 5, 11: 3, 12: 2,
                        13: 6, 14: 6, 15: 2, 16: 2, 17: 6, 19: 2, 20: 2}
    h17 = ko.Table(17, True).count_files(r12).hist()
    assert len(h17) == 10001 and [int(x) for x in h17[:6]] == [121532, 14124, 2531, 1007, 537, 164]
    assert _rows(ko.Table(27, True).count_files([os.path.join(refdata, "sect_length_test.fa")]).hist()) == {1094: 18, 1095: 16}
    assert _rows(ko.Table(27, True).count_files([os.path.join(refdata, "sect_test.fa")]).hist()) == {1: 26}


HIST_BODY_MD5 = {"sect_length_test.fa": "a915d3c7e7d24ef6de8e049b2f63fe46", "sect_test.fa": "411651418f41c1fb2b7e6f1955900ff4"}


def hist_body_md5(path):
    import hashlib
    return hashlib.md5(b"".join(ln for ln in open(path, "rb").read().splitlines(True) if not ln.startswith(b"#"))).hexdigest()


def test_survey_hist_file_bodies(ko, refdata, tmp_path):
    """md5 of the non-'#' lines of the file `kat hist -m27` wrote for the reference's two FASTA fixtures (recorded by
    the survey stage from a hand-built reference binary): pins count + Histogram::bin + the body Histogram::print writes."""
    for name, want in HIST_BODY_MD5.items():
        p = os.path.join(refdata, name)
        out = str(tmp_path / (name + ".hist"))
        ko.write_hist(out, 27, [p], 1, 10000, 1, ko.Table(27, True).count_files([p]).hist())
        assert hist_body_md5(out) == want, name


def test_survey_known_answers_gcp_comp(ko, refdata, tmp_path):
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    g = ko.Table(17, True).count_files([r1, r2]).gcp()
    assert g.shape == (17, 1001) and int(g.max()) == 21046 and [int(x) for x in g[0, :4]] == [0, 5, 0, 0]
    t1, t2 = ko.Table(13, True).count_files([r1]), ko.Table(13, True).count_files([r2])
    mx, cc, sp = ko.comp(t1, t2)
    assert int(mx.max()) == 62111
    assert [int(x) for x in cc] == [87929, 88000, 0, 80366, 80554, 0, 66743, 67516, 64113, 64301, 21186, 20484, 16253]
    ko.write_comp(str(tmp_path / "c"), 13, [r1], [r2], 1001, 1001, mx, cc, sp)
    stats = (tmp_path / "c.stats").read_text()
    for line in (" - Manhattan distance: 736", " - Euclidean distance: 414.886", " - Cosine distance: 1.48546e-05",
                 " - Canberra distance: 5.56051", " - Jaccard distance: 0.00910576", " - Manhattan distance: 714",
                 " - Euclidean distance: 364.483", " - Cosine distance: 0.000173578", " - Canberra distance: 4.62095",
                 " - Jaccard distance: 0.0429862"):
        assert line + "\n" in stats, line
    # quirk B2 (src/comp.cc:447): with -N -O the reverse lookup still canonicalises
    n1, n2 = ko.Table(21, False).count_files([r1]), ko.Table(21, False).count_files([r2])
    _, cc, _ = ko.comp(n1, n2)
    assert (int(cc[3]), int(cc[4]), int(cc[8]), int(cc[9])) == (76378, 76464, 71337, 68064)


def test_output_formats_match_reference_examples(ko, refdata, tmp_path):
    """Header layout of the writers vs the real KAT outputs kept in the reference tree
    (scripts/test/resources/hist1.hist, gcp1.mx, spectracn1.mx -- restated here as the expected header lines)."""
    p = [os.path.join(refdata, "sect_test.fa")]
    t = ko.Table(27, True).count_files(p)
    ko.write_hist(str(tmp_path / "h"), 27, ["/dev/fd/63"], 1, 10000, 1, t.hist())
    lines = (tmp_path / "h").read_text().split("\n")
    assert lines[:7] == ["# Title:27-mer spectra for: 63", "# XLabel:27-mer frequency", "# YLabel:# distinct 27-mers",
                         "# Kmer value:27", "# Input 1:<pipe>", "###", "1 26"]               # hist1.hist:1-6
    assert len(lines) == 6 + 10001 + 1 and lines[-2].startswith("10001 ")
    ko.write_gcp(str(tmp_path / "g.mx"), 27, ["/dev/fd/63"], 1000, t.gcp())
    lines = (tmp_path / "g.mx").read_text().split("\n")
    assert lines[:11] == ["# Title:K-mer coverage vs GC count plot for: 63", "# XLabel:27-mer frequency", "# YLabel:GC count",
                          "# ZLabel:# distinct 27-mers", "# Columns:1001", "# Rows:27", "# MaxVal:%d" % int(t.gcp().max()),
                          "# Transpose:0", "# Kmer value:27", "# Input 1:<pipe>", "###"]      # gcp1.mx:1-11
    assert len(lines) == 11 + 27 + 1 and all(len(l.split(" ")) == 1001 for l in lines[11:38])
    mx, cc, sp = ko.comp(t, t)
    ko.write_comp(str(tmp_path / "c"), 27, ["a/LIB_R1.fastq.gz"], ["b/LIB_R2.fastq.gz"], 1001, 1001, mx, cc, sp, hists=True)
    lines = (tmp_path / "c-main.mx").read_text().split("\n")
    assert lines[:12] == ["# Title:K-mer comparison plot", "# XLabel:27-mer frequency for: LIB_R1.fastq.gz",
                          "# YLabel:27-mer frequency for: LIB_R2.fastq.gz", "# ZLabel:# distinct 27-mers", "# Columns:1001",
                          "# Rows:1001", "# MaxVal:%d" % int(mx.max()), "# Transpose:1", "# Kmer value:27",
                          "# Input 1:a/LIB_R1.fastq.gz", "# Input 2:b/LIB_R2.fastq.gz", "###"]  # spectracn1.mx:1-12
    assert (tmp_path / "c.stats").read_text().startswith('K-mer statistics for: \n - Hash 1: "a/LIB_R1.fastq.gz"\n - Hash 2: "b/LIB_R2.fastq.gz"\n\nTotal K-mers in: \n')
    assert (tmp_path / "c.1.hist").read_text().startswith("# Title:27-mer spectra for: a/LIB_R1.fastq.gz\n# XLabel:27-mer frequency\n# YLabel:# distinct 27-mers\n###\n0 0\n1 26\n")
