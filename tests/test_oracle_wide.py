"""oracle/koracle_wide.c (k-mers of up to 64 bases in one 128-bit word) against three independent statements:
koracle.c entry by entry for k <= 32, the naive string counter (tests/naive.py) for k > 32, and -- where oracle/_ref is built --
the reference's own parser + mer_iterator + mer_dna (jf_ref kmers), whose multi-word mer_dna is what KAT runs at k > 32."""
import os

import numpy as np
import pytest

from tests import naive
from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq
from tests.test_oracle_vs_reference import JF_REF, ref

have_jf_ref = pytest.mark.skipif(not os.access(JF_REF, os.X_OK), reason="oracle/_ref not built (no /root/reference)")


def wide_dict(t):
    hi, lo, c = t.dump_sorted()
    return {(int(a) << 64) | int(b): int(n) for a, b, n in zip(hi, lo, c)}


def decode(key, k):
    return "".join("ACGT"[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))


def _files(tmp_path, seed):
    rng = np.random.default_rng(seed)
    fa, fq, fqm = tmp_path / "w.fa", tmp_path / "w.fq", tmp_path / "wm.fq"
    write_messy_fasta(str(fa), rng, n_rec=25)
    write_messy_fastq(str(fq), rng, n_rec=150)
    write_messy_fastq(str(fqm), rng, n_rec=100, multiline=True)
    long_fa = tmp_path / "long.fa"                  # runs long enough for 64-mers, both cases, a few N
    g = rng.choice(list("ACGTacgt"), 6000)
    g[rng.integers(0, 6000, 12)] = "N"
    s = "".join(g)
    long_fa.write_text(">a\n" + "\n".join(s[i:i + 70] for i in range(0, 3000, 70)) + "\n>b\n" + s[3000:] + "\n>c\n" + naive.revcomp(s[3100:3400].upper().replace("N", "A")) + "\n")
    return [str(fa), str(fq), str(fqm), str(long_fa)]


@pytest.mark.parametrize("k,canonical", [(1, True), (5, False), (17, True), (27, True), (31, False), (32, True), (32, False)])
def test_wide_equals_narrow_oracle_up_to_32(ko, tmp_path, k, canonical):
    paths = _files(tmp_path, k)
    w = ko.WideTable(k, canonical).count_files(paths)
    n = ko.Table(k, canonical).count_files(paths)
    keys, counts = n.dump_sorted()
    assert wide_dict(w) == {int(a): int(b) for a, b in zip(keys, counts)}
    assert (w.distinct, w.total) == (n.distinct, n.total)
    assert np.array_equal(w.hist(2, 30, 3), n.hist(2, 30, 3))
    assert np.array_equal(w.gcp(0.7, 40), n.gcp(0.7, 40))
    w2 = ko.WideTable(k, not canonical).count_files(paths[:2])
    n2 = ko.Table(k, not canonical).count_files(paths[:2])
    for a, b in zip(ko.comp(w, w2, 0.5, 2.0, 40, 60), ko.comp(n, n2, 0.5, 2.0, 40, 60)):
        assert np.array_equal(a, b)
    w3, n3 = ko.WideTable(k, canonical).count_files(paths[2:]), ko.Table(k, canonical).count_files(paths[2:])
    for a, b in zip(ko.comp3(w, w2, w3, 1.0, 1.0, 30, 30), ko.comp3(n, n2, n3, 1.0, 1.0, 30, 30)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k,canonical", [(33, True), (33, False), (40, True), (47, False), (63, True), (63, False), (64, True), (64, False)])
def test_wide_against_naive_strings(ko, tmp_path, k, canonical):
    paths = _files(tmp_path, 100 + k)
    want = naive.count_files(paths, k, canonical)
    got = wide_dict(ko.WideTable(k, canonical).count_files(paths))
    assert len(want) > 1000
    assert got == {naive.pack(s): c for s, c in want.items()}


@have_jf_ref
@pytest.mark.parametrize("k,canonical", [(33, True), (48, True), (48, False), (63, True), (64, False)])
def test_wide_against_the_reference_parser_and_mer_dna(ko, tmp_path, refdata, k, canonical):
    """jf_ref kmers = mer_overlap_sequence_parser + mer_iterator + multi-word mer_dna compiled from /root/reference."""
    for paths in (_files(tmp_path, k), [os.path.join(refdata, f) for f in ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq", "sect_length_test.fa")]):
        rc, out = ref(JF_REF, ["kmers", k, int(canonical)] + paths)
        assert rc == 0
        hi, lo, c = ko.WideTable(k, canonical).count_files(paths).dump_sorted()
        got = "".join("%s %d\n" % (decode((int(a) << 64) | int(b), k), int(n)) for a, b, n in zip(hi, lo, c)).encode()
        assert got == out and out.count(b"\n") > 1000


def test_wide_reducers_against_direct_python(ko):
    """hist / gcp / comp at k = 45 restated from the reference formulas on a hand-made multiset (as test_oracle_vs_naive does at k = 7)."""
    k = 45
    rng = np.random.default_rng(8)
    keys = [int.from_bytes(rng.bytes(12), "big") & ((1 << (2 * k)) - 1) for _ in range(600)]
    keys = list(dict.fromkeys(keys))
    t1, t2 = ko.WideTable(k, False), ko.WideTable(k, False)
    d1, d2 = {}, {}
    for key in keys[:450]:
        c = int(rng.integers(1, 40)); t1.add(key, c); d1[key] = c
    for key in keys[200:]:
        c = int(rng.integers(1, 2000)); t2.add(key, c); d2[key] = c
    rc_keys = [naive.pack(naive.revcomp(decode(key, k))) for key in keys[:40]]          # hash-2 k-mers whose canonical form is in hash 1
    for key in rc_keys:
        if key not in d2:
            t2.add(key, 9); d2[key] = 9
    assert t1.get(keys[0]) == d1[keys[0]] and t1.get(keys[-1]) == 0
    base, ceil_, nb = ko.hist_geometry(3, 20)
    want = np.zeros(nb, np.uint64)
    for v in d1.values():
        want[0 if v < base else nb - 1 if v > ceil_ else (v - base) // 2] += 1
    assert np.array_equal(t1.hist(3, 20, 2), want)
    want = np.zeros((k, 11), np.uint64)
    for key, v in d1.items():
        s = decode(key, k)
        g = s.count("G") + s.count("C")
        if g < k:
            want[g, min(int(np.ceil(v * 0.5)), 10)] += 1
    assert np.array_equal(t1.gcp(0.5, 10), want)
    mx, cc, sp = ko.comp(t1, t2, 1.0, 0.1, 30, 50)
    wmx = np.zeros((30, 50), np.uint64)
    for key, a in d1.items():
        b = d2.get(key, 0)
        wmx[min(a, 29), min(int(np.ceil(b * 0.1)) if b else 0, 49)] += 1
    h2only = 0
    for key, b in d2.items():
        s = decode(key, k)
        can = min(key, naive.pack(naive.revcomp(s)))
        if d1.get(can, 0) == 0:
            wmx[0, min(int(np.ceil(b * 0.1)), 49)] += 1
            h2only += 1
    assert np.array_equal(mx, wmx) and int(cc[9]) == h2only
    assert int(cc[0]) == sum(d1.values()) and int(cc[1]) == sum(d2.values()) and int(cc[3]) == len(d1) and int(cc[4]) == len(d2)


def test_wide_k_limits(ko):
    with pytest.raises(ko.OracleError):
        ko.WideTable(65)
    with pytest.raises(ko.OracleError):
        ko.WideTable(0)
    t = ko.WideTable(64, False).count_bases(b"T" * 70 + b"N" + b"A" * 64)
    assert wide_dict(t) == {(1 << 128) - 1: 7, 0: 1}
