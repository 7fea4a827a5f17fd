"""Oracle (C) vs an independent naive Python counter on messy small inputs -- the check SURVEY.md Appendix D describes
(mixed line widths, lower case, N / IUPAC / '-', '@'-leading quality lines, multi-line FASTQ, gzip)."""
import gzip
import os

import numpy as np
import pytest

from tests import naive


def write_messy_fasta(path, rng, n_rec=40):
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRY-", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n_rec):
            L = int(rng.integers(10, 3000))
            seq = rng.choice(alphabet, size=L).tobytes()
            f.write(b">rec%d some description\n" % i)
            w = int(rng.integers(7, 90))
            for o in range(0, L, w):
                f.write(seq[o:o + w] + b"\n")
            if i % 7 == 0:
                f.write(b"\n")                      # stray blank line after a record


def write_messy_fastq(path, rng, n_rec=300, multiline=False):
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTN", dtype=np.uint8)
    qual = np.frombuffer(b"@+IIIIIIII#5", dtype=np.uint8)     # qualities may start with '@' or '+'
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        for i in range(n_rec):
            L = int(rng.integers(1, 260))
            seq = rng.choice(alphabet, size=L).tobytes()
            q = rng.choice(qual, size=L).tobytes()
            f.write(b"@read%d/1\n" % i)
            if multiline and L > 20:
                h = L // 2
                f.write(seq[:h] + b"\n" + seq[h:] + b"\n+\n" + q[:h] + b"\n" + q[h:] + b"\n")
            else:
                f.write(seq + b"\n+read%d\n" % i + q + b"\n")


def as_dict(table):
    keys, counts = table.dump_sorted()
    return {int(k): int(c) for k, c in zip(keys, counts)}


@pytest.mark.parametrize("k,canonical", [(27, True), (11, False), (31, True), (32, False), (3, True)])
def test_fasta(ko, tmp_path, k, canonical):
    rng = np.random.default_rng(k)
    p = str(tmp_path / "m.fa")
    write_messy_fasta(p, rng)
    want = {naive.pack(w): c for w, c in naive.count_files([p], k, canonical).items()}
    assert as_dict(ko.Table(k, canonical).count_files([p])) == want


@pytest.mark.parametrize("name,multiline", [("a.fq", False), ("b.fastq", True), ("c.fq.gz", False)])
def test_fastq(ko, tmp_path, name, multiline):
    rng = np.random.default_rng(len(name))
    p = str(tmp_path / name)
    write_messy_fastq(p, rng, multiline=multiline)
    for k, canonical in ((21, True), (27, False)):
        want = {naive.pack(w): c for w, c in naive.count_files([p], k, canonical).items()}
        assert as_dict(ko.Table(k, canonical).count_files([p])) == want


def test_group_of_files_never_joins(ko, tmp_path):
    a, b = tmp_path / "a.fa", tmp_path / "b.fa"
    a.write_text(">x\nACGTACGTAC")          # no trailing newline
    b.write_text(">y\nGTACGTACGT\n")
    t = ko.Table(5, False).count_files([str(a), str(b)])
    want = naive.count_string("ACGTACGTAC", 5, False)
    naive.count_string("GTACGTACGT", 5, False, want)
    assert as_dict(t) == {naive.pack(w): c for w, c in want.items()}
    assert t.total == 12


def test_mt_counting_equals_scalar(ko):
    rng = np.random.default_rng(1)
    s = rng.choice(np.frombuffer(b"ACGTACGTACGTN", dtype=np.uint8), size=3_000_000)
    a = ko.Table(25, True).count_bases(s)
    b = ko.Table(25, True).count_bases(s, threads=5)
    ka, ca = a.dump_sorted()
    kb, cb = b.dump_sorted()
    assert np.array_equal(ka, kb) and np.array_equal(ca, cb)


def test_reducers_against_direct_python(ko):
    """hist / gcp / comp reducers restated directly from the reference formulas on a hand-made multiset."""
    k = 7
    rng = np.random.default_rng(3)
    keys = rng.choice(4 ** k, size=500, replace=False)
    c1 = rng.integers(1, 40, size=500)
    t1, t2 = ko.Table(k, False), ko.Table(k, False)
    d1, d2 = {}, {}
    for key, c in zip(keys, c1):
        t1.add(int(key), int(c)); d1[int(key)] = int(c)
    for key in keys[200:]:
        c = int(rng.integers(1, 2000)); t2.add(int(key), c); d2[int(key)] = c
    for key in rng.choice(4 ** k, size=100):
        if int(key) not in d1 and int(key) not in d2:
            t2.add(int(key), 5); d2[int(key)] = 5
    # hist (src/histogram.cc:183-199) low=3 high=20 inc=2
    base, ceil_, nb = ko.hist_geometry(3, 20)
    want = np.zeros(nb, np.uint64)
    for v in d1.values():
        want[0 if v < base else nb - 1 if v > ceil_ else (v - base) // 2] += 1
    assert np.array_equal(t1.hist(3, 20, 2), want)
    # gcp (src/gcp.cc:179-197) scale 0.5 bins 10; rows = k
    want = np.zeros((k, 11), np.uint64)
    for key, v in d1.items():
        s = ko.decode(key, k)
        g = s.count("G") + s.count("C")
        pos = min(int(np.ceil(v * 0.5)), 10)
        if g < k:
            want[g, pos] += 1
    assert np.array_equal(t1.gcp(0.5, 10), want)
    # comp (src/comp.cc:387-484), non-canonical tables: pass 2 still canonicalises its probe
    mx, cc, sp = ko.comp(t1, t2, 1.0, 0.1, 30, 50)
    wmx = np.zeros((30, 50), np.uint64)
    for key, a in d1.items():
        b = d2.get(key, 0)
        wmx[min(a, 29), min(int(np.ceil(b * 0.1)) if b else 0, 49)] += 1
    h2only = 0
    for key, b in d2.items():
        a = d1.get(ko.canonical(key, k), 0)
        if a == 0:
            wmx[0, min(int(np.ceil(b * 0.1)), 49)] += 1
            h2only += 1
    assert np.array_equal(mx, wmx) and int(cc[9]) == h2only
    assert int(cc[0]) == sum(d1.values()) and int(cc[1]) == sum(d2.values()) and int(cc[3]) == len(d1) and int(cc[4]) == len(d2)


def test_three_input_comp_against_direct_python(ko):
    """ends / middle / mixed matrices + hash-3 counters (src/comp.cc:403-433,466-479) restated directly."""
    k = 9
    rng = np.random.default_rng(11)
    pool = rng.choice(4 ** k, size=900, replace=False)
    d = [{}, {}, {}]
    tabs = [ko.Table(k, False), ko.Table(k, True), ko.Table(k, False)]
    for t, dd, keys, hi in ((0, d[0], pool[:600], 30), (1, d[1], pool[200:800], 200), (2, d[2], pool[400:], 60)):
        for key in keys:
            key = int(key)
            if tabs[t].canonical:
                key = ko.canonical(key, k)
            c = int(rng.integers(1, hi))
            tabs[t].add(key, c)
            dd[key] = dd.get(key, 0) + c
    main, ends, middle, mixed, cc, sp = ko.comp3(tabs[0], tabs[1], tabs[2], 1.0, 0.2, 25, 12)
    m2, c2, s2 = ko.comp(tabs[0], tabs[1], 1.0, 0.2, 25, 12)
    assert np.array_equal(main, m2) and np.array_equal(sp, s2)
    sc = lambda c, s, n: min(int(np.ceil(c * s)) if c else 0, n - 1)
    we, wmid, wmix = (np.zeros((25, 12), np.uint64) for _ in range(3))
    for key, c1 in d[0].items():
        c2_ = d[1].get(ko.canonical(key, k), 0)          # input 2 canonical -> probe canonicalised
        c3_ = d[2].get(key, 0)                           # input 3 not canonical -> probe as is
        s1, s2_, s3_ = sc(c1, 1.0, 25), sc(c2_, 0.2, 12), sc(c3_, 0.2, 12)
        (we if s2_ == s3_ else wmix if s3_ > 0 else wmid)[s1, s3_] += 1
    assert np.array_equal(ends, we) and np.array_equal(middle, wmid) and np.array_equal(mixed, wmix)
    assert int(cc[2]) == sum(d[2].values()) and int(cc[5]) == len(d[2])
    assert int(ends.sum() + middle.sum() + mixed.sum()) == len(d[0])
