"""The oracle (and the product's .jf writer) against the REAL reference code, where that code builds in this image.

oracle/Makefile (`make ref`) compiles, from the sources where they lie under /root/reference and with nothing stubbed,
  oracle/_ref/jf_ref         Jellyfish 2.2.0's parser + mer_iterator + mer_dna + file_header / binary_reader
  oracle/_ref/kat_ref_parts  KAT's CompCounters, distance metrics, SparseMatrix and str_utils
(the tool drivers and Jellyfish's hash array need the autoconf-generated config.h and do not build).  These tests run
wherever the two binaries exist -- in the build container, and on the GPU box, where the prebuilt binaries travel -- and
tests/golden/reference_vectors.json holds what they printed for the fixed inputs (tests/golden/make_reference_vectors.py), so
that the pin survives a checkout without /root/reference."""
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import kat_amd
from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JF_REF = os.path.join(ROOT, "oracle", "_ref", "jf_ref")
KAT_REF = os.path.join(ROOT, "oracle", "_ref", "kat_ref_parts")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_vectors.json")
have_ref = pytest.mark.skipif(not (os.access(JF_REF, os.X_OK) and os.access(KAT_REF, os.X_OK)), reason="oracle/_ref not built (no /root/reference)")


def ref(binary, args, stdin=None):
    r = subprocess.run([binary] + [str(a) for a in args], input=stdin, capture_output=True, timeout=300)
    return r.returncode, r.stdout


def ref_kmers(paths, k, canonical):
    rc, out = ref(JF_REF, ["kmers", k, int(canonical)] + list(paths))
    assert rc == 0, (paths, rc)
    return out


def oracle_kmers(ko, paths, k, canonical):
    if k > 32:                                                   # koracle_wide.c: (hi, lo) = the 2k-bit word
        hi, lo, counts = ko.WideTable(k, canonical).count_files(list(paths)).dump_sorted()
        dec = lambda x: "".join("ACGT"[(x >> (2 * (k - 1 - i))) & 3] for i in range(k))
        return "".join("%s %d\n" % (dec((int(a) << 64) | int(b)), int(c)) for a, b, c in zip(hi, lo, counts)).encode()
    keys, counts = ko.Table(k, canonical).count_files(list(paths)).dump_sorted()
    return "".join("%s %d\n" % (ko.decode(int(a), k), int(b)) for a, b in zip(keys, counts)).encode()


def digest(b):
    return hashlib.sha256(b).hexdigest()


FIXED = [("sect_test.fa",), ("sect_length_test.fa",), ("ecoli_r1.1K.fastq",), ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq")]
KC = [(27, True), (17, False), (5, True), (32, False), (31, True), (33, True), (48, False), (63, True)]


def fixed_cases(refdata):
    for names in FIXED:
        for k, c in KC:
            yield "+".join(names) + ":k%d:%s" % (k, "C" if c else "N"), [os.path.join(refdata, n) for n in names], k, c


def test_golden_reference_vectors(ko, refdata):
    """The committed digests of what the reference's parser + iterator deliver for its own test data: needs no _ref."""
    gold = json.load(open(GOLDEN))
    for tag, paths, k, c in fixed_cases(refdata):
        got = oracle_kmers(ko, paths, k, c)
        g = gold["kmers"][tag]
        assert (digest(got), got.count(b"\n")) == (g["sha256"], g["distinct"]), tag
    # `kat hist -m27` of sect_test.fa has 26 distinct 27-mers, all of count 1
    assert gold["kmers"]["sect_test.fa:k27:C"]["distinct"] == 26


@have_ref
def test_reference_vectors_are_current(refdata):
    gold = json.load(open(GOLDEN))
    for tag, paths, k, c in fixed_cases(refdata):
        out = ref_kmers(paths, k, c)
        assert gold["kmers"][tag] == {"sha256": digest(out), "distinct": out.count(b"\n"), "total": sum(int(l.split()[1]) for l in out.splitlines())}, tag


@have_ref
@pytest.mark.parametrize("seed", range(3))
def test_count_semantics_on_messy_files(ko, tmp_path, seed):
    """Multi-line records, blank lines, IUPAC / lower case / '-', qualities that start with '@' or '+', gzip: the oracle's
    k-mer multiset is the reference parser's."""
    rng = np.random.default_rng(seed)
    fa, fq, fqm, gz = tmp_path / "m.fa", tmp_path / "m.fq", tmp_path / "mm.fq", tmp_path / "z.fq.gz"
    write_messy_fasta(str(fa), rng)
    write_messy_fastq(str(fq), rng)
    write_messy_fastq(str(fqm), rng, multiline=True)
    write_messy_fastq(str(gz), rng)
    for k, c in ((27, True), (11, False), (32, True), (3, False)):
        for paths in ([str(fa)], [str(fq)], [str(fqm)], [str(gz)], [str(fq), str(fa)]):
            assert oracle_kmers(ko, paths, k, c) == ref_kmers(paths, k, c), (paths, k, c)


@have_ref
def test_count_semantics_on_edge_files(ko, tmp_path):
    cases = {
        "header_only.fa": b">x\n",
        "no_newline.fa": b">x\nACGTACGTAC",
        "crlf.fa": b">x\r\nACGTACG\r\nACGTTTT\r\n>y\r\nTTGGAAC\r\n",                  # '\r' breaks k-mers (quirk B8)
        "blank_after_header.fa": b">a\n\n>ACGTACGT\nGGAACC\n>c\nTTGGCCAA\n",       # the line after the blank one is sequence
        "gt_inside.fa": b">a\nACGG>GTAACC\nAATTGG\n",
        "seam.fa": b">x\n" + b"ACGTTGCA" * 1100 + b"\n>y\n" + b"GATTACA" * 700 + b"\n",   # lines longer than the parser's 4096-byte buffers
        "one.fq": b"@r\nACGTACGTT\n+\nIIIIIIIII\n",
        "at_quality.fq": b"@r\nACGTACG\n+\n@IIIIII\n@s\nGGATTCA\n+\n@@@@@@@\n",
        "lower.fa": b">l\nacgtacgtnnACGTacgt\n",
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        for k, c in ((4, True), (7, False)):
            assert oracle_kmers(ko, [str(p)], k, c) == ref_kmers([str(p)], k, c), (name, k, c)


@have_ref
def test_mer_dna_arithmetic(ko):
    rng = np.random.default_rng(9)
    for k in (1, 2, 5, 16, 27, 31, 32):
        mers = ["".join(rng.choice(list("ACGT"), k)) for _ in range(40)] + ["A" * k, "T" * k, "ACGT" * 8][:42]
        mers = [m[:k] for m in mers]
        rc, out = ref(JF_REF, ["merops", k] + mers)
        assert rc == 0
        for m, line in zip(mers, out.decode().splitlines()):
            s, bits, rcs, can, lt = line.split()
            key = ko.encode(m)
            assert s == m and int(bits) == key
            assert ko.decode(ko.revcomp(key, k), k) == rcs and ko.decode(ko.canonical(key, k), k) == can
            assert (key < ko.revcomp(key, k)) == bool(int(lt))


def parse_jfread(out):
    lines = out.decode().splitlines()
    hdr = dict(l.split(" ", 1) for l in lines[:8])
    recs = [(a, int(b), int(c)) for a, b, c in (l.split() for l in lines[8:])]
    return hdr, recs


@have_ref
def test_jf_files_as_the_reference_reads_them(ko, refdata, tmp_path):
    """(a) the reference's fixture: our reader sees what the reference's reader sees; (b) a file written by the PRODUCT's writer is
    read back by the reference's file_header + binary_reader with every field, record and hash position where it must be."""
    fixture = os.path.join(refdata, "ecoli.header.jf27")
    rc, out = ref(JF_REF, ["jfread", fixture])
    assert rc == 0
    hdr, recs = parse_jfread(out)
    t = ko.Table.from_jf(fixture)
    keys, counts = t.dump_sorted()
    assert sorted((ko.encode(a), b) for a, b, _ in recs) == [(int(a), int(b)) for a, b in zip(keys, counts)]
    assert (hdr["key_len"], hdr["counter_len"], hdr["canonical"], hdr["format"]) == ("54", "4", "0", "binary/sorted") and len(recs) == 1889
    k2, can2, pk, pc = kat_amd.jf_read_records(fixture)
    assert k2 == 27 and not can2 and sorted(zip(pk.tolist(), pc.tolist())) == [(int(a), int(b)) for a, b in zip(keys, counts)]
    rng = np.random.default_rng(4)
    for k, canonical, n in ((27, True, 5000), (31, False, 1), (15, True, 70000), (32, False, 300)):
        kk = np.unique(rng.integers(0, 1 << (2 * k) if k < 32 else (1 << 63), n, dtype=np.uint64))
        if canonical:
            kk = np.unique(np.array([ko.canonical(int(x), k) for x in kk], np.uint64))
        cc = rng.integers(1, 1 << 20, kk.size).astype(np.uint64)
        cc[0] = (1 << 40)                                              # saturates at the 4-byte counter (quirk B12)
        p = str(tmp_path / ("w%d.jf" % k))
        kat_amd.jf_write_records(p, k, canonical, kk, cc)
        rc, out = ref(JF_REF, ["jfread", p])
        assert rc == 0, (k, rc)
        hdr, recs = parse_jfread(out)
        assert (hdr["key_len"], hdr["counter_len"], hdr["canonical"], hdr["format"]) == (str(2 * k), "4", str(int(canonical)), "binary/sorted")
        want = {int(a): min(int(b), 0xFFFFFFFF) for a, b in zip(kk, cc)}
        assert {ko.encode(a): b for a, b, _ in recs} == want
        pos = [c for _, _, c in recs]
        assert pos == sorted(pos)                                       # binary/sorted: by hash position, as the reference's readers expect
    # k > 32 (two-word k-mers): the same writer through its (hi, lo) entry point, read by the reference's multi-word mer_dna
    M = (1 << 64) - 1
    pack = lambda s_: sum("ACGT".index(ch) << (2 * (len(s_) - 1 - i)) for i, ch in enumerate(s_))
    for k, canonical, n in ((33, True, 3000), (48, False, 1), (63, True, 40000), (40, False, 500)):
        kk = sorted({int.from_bytes(rng.bytes(16), "big") & ((1 << (2 * k)) - 1) for _ in range(n)})
        cc = rng.integers(1, 1 << 20, len(kk)).astype(np.uint64)
        cc[0] = (1 << 40)
        p = str(tmp_path / ("w%d.jf" % k))
        kat_amd.jf_write_records_wide(p, k, canonical, [x >> 64 for x in kk], [x & M for x in kk], cc)
        rc, out = ref(JF_REF, ["jfread", p])
        assert rc == 0, (k, rc)
        hdr, recs = parse_jfread(out)
        assert (hdr["key_len"], hdr["counter_len"], hdr["canonical"], hdr["format"]) == (str(2 * k), "4", str(int(canonical)), "binary/sorted")
        assert {pack(a): b for a, b, _ in recs} == {x: min(int(b), 0xFFFFFFFF) for x, b in zip(kk, cc)}
        pos = [c for _, _, c in recs]
        assert pos == sorted(pos)
        k2, can2, hi, lo, cnt = kat_amd.jf_read_records_wide(p)
        assert (k2, can2) == (k, canonical)
        assert sorted(((int(a) << 64) | int(b), int(c)) for a, b, c in zip(hi, lo, cnt)) == [(x, min(int(b), 0xFFFFFFFF)) for x, b in zip(kk, cc)]
        with pytest.raises(kat_amd.KatGpuError) as ei:
            kat_amd.jf_read_records(p)                                   # the one-word reader says what it is
        assert ei.value.code == 6


@have_ref
def test_comp_counters_text_and_distances(ko, tmp_path):
    """CompCounters::printCounts (with its five distance metrics, twice) byte for byte, on random counters and spectra."""
    rng = np.random.default_rng(12)
    for trial in range(6):
        n = int(rng.integers(2, 40))
        cc = rng.integers(0, 1 << 40, 13).astype(np.uint64)
        if trial % 2:
            cc[2] = cc[5] = 0                                          # two-input form: no "Hash 3" lines
        sp = rng.integers(0, [5, 1 << 20, 1 << 33][trial % 3], (4, n)).astype(np.uint64)
        paths = ["/data/reads_1.fq", "dir with space/asm\"x&y.fa", "third.jf27"]
        stdin = "\n".join(paths) + "\n%d\n%s\n%s\n" % (n, " ".join(map(str, cc)), "\n".join(" ".join(map(str, r)) for r in sp))
        rc, out = ref(KAT_REF, ["compstats"], stdin.encode())
        assert rc == 0
        p = str(tmp_path / "s.stats")
        L = ko.lib()
        assert L.ko_write_comp_stats3(p.encode(), paths[0].encode(), paths[1].encode(), paths[2].encode(), cc.ctypes.data, sp.ctypes.data, n) == 0
        assert open(p, "rb").read() == out, trial
        for a, b in ((sp[0], sp[1]), (sp[2], sp[3])):
            rc, dout = ref(KAT_REF, ["distance"], ("%d\n%s\n%s\n" % (n, " ".join(map(str, a)), " ".join(map(str, b)))).encode())
            want = [float(x) if x not in (b"-nan", b"nan") else float("nan") for x in dout.split()]
            got = [ko.distance(w, a, b) for w in range(5)]
            assert all((g != g and w != w) or abs(g - w) <= 1e-5 * max(1.0, abs(w)) for g, w in zip(got, want)), (got, want)


@have_ref
def test_comp_counter_arithmetic(ko):
    """update{Hash1,Hash2,Shared}Counters applied as Comp::compareSlice applies them == the oracle's ko_comp on the same tables."""
    rng = np.random.default_rng(3)
    k = 9
    for trial in range(4):
        t1, t2 = ko.Table(k, True), ko.Table(k, True)
        keys = [ko.canonical(int(x), k) for x in rng.integers(0, 1 << (2 * k), 300)]
        for x in keys[:200]:
            t1.add(x, int(rng.integers(1, 60)))
        for x in keys[120:]:
            t2.add(x, int(rng.integers(1, 2000)))
        n = 25
        mx, cc, sp = ko.comp(t1, t2, 1.0, 1.0, n, n)
        k1, c1 = t1.dump_sorted()
        k2, c2 = t2.dump_sorted()
        lines = ["%d" % n] + ["h1 %d %d" % (int(c), t2.get(int(x))) for x, c in zip(k1, c1)] + ["h2 %d %d" % (t1.get(int(x)), int(c)) for x, c in zip(k2, c2)]
        rc, out = ref(KAT_REF, ["compupdate"], ("\n".join(lines) + "\n").encode())
        assert rc == 0
        rows = [list(map(int, l.split())) for l in out.decode().splitlines()]
        assert rows[0] == [int(x) for x in cc]
        assert np.array_equal(np.array(rows[1:5], np.uint64), sp)


@have_ref
def test_sparse_matrix_and_str_utils(ko):
    rc, out = ref(KAT_REF, ["matrix", 4, 6], b"0 0 5\n3 5 11\n3 5 1\n4 2 1000\n2 6 77\n1 3 9\n")     # (4,2) and (2,6) lie outside a 4 x 6 matrix
    assert out == b"12\n5 0 0 0 0 0\n0 0 0 9 0 0\n0 0 0 0 0 0\n0 0 0 0 0 12\n"
    words = ["ACGT", "acgtn", "GGCC", "ACGU", "", "NNNN", "gCgC-"]
    rc, out = ref(KAT_REF, ["strutils"], ("\n".join(words) + "\n").encode())
    for w, line in zip(words, out.decode().splitlines()):
        valid, gc = map(int, line.split())
        assert valid == int(all(ch in "ACGTacgt" for ch in w)) and gc == sum(ch in "GCgc" for ch in w)
    # the same rule decides which windows `kat sect` looks up (oracle/koracle_sect.c: ko_profile)
    t = ko.Table(4, False)
    t.add(ko.encode("ACGT"), 3)
    counts, gcs = ko.profile(t, "ACGTNACGTacgt", False)
    assert gcs.tolist() == [2, -1, -1, -1, -1, 2, 2, 2, 2, 2] and counts.tolist()[0] == 3 and counts.tolist()[5] == 3


@have_ref
def test_matrix_files_as_the_reference_reads_them(ko, refdata, tmp_path):
    """The .mx files of `kat gcp` and `kat comp` (oracle writers == the product's CLI bytes, tests/test_gpu_cli.py) go through the
    reference's own readers: matrix_metadata_extractor finds every key, SparseMatrix(path) loads the body with the right shape,
    MaxVal and cell sum."""
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    t1, t2 = ko.Table(17, True).count_files([r1]), ko.Table(17, True).count_files([r2])
    g = t1.gcp(1.0, 200)
    p = str(tmp_path / "g.mx")
    ko.write_gcp(p, 17, [r1], 200, g)
    rc, out = ref(KAT_REF, ["mxread", p])
    lines = out.decode().split("\n")
    assert rc == 0 and lines[0].split() == ["201", "17", str(int(g.max())), "0", "17"]
    assert lines[1:5] == ["K-mer coverage vs GC count plot for: ecoli_r1.1K.fastq", "17-mer frequency", "GC count", "# distinct 17-mers"] and lines[5] == r1
    assert lines[7].split() == ["17", "201", str(int(g.max())), str(int(g.sum()))]
    mx, cc, sp = ko.comp(t1, t2, 1.0, 1.0, 60, 40)
    ko.write_comp(str(tmp_path / "c"), 17, [r1], [r2], 60, 40, mx, cc, sp)
    rc, out = ref(KAT_REF, ["mxread", str(tmp_path / "c-main.mx")])
    lines = out.decode().split("\n")
    assert rc == 0 and lines[0].split() == ["40", "60", str(int(mx.max())), "1", "17"]
    assert lines[1] == "K-mer comparison plot" and lines[5] == r1 and lines[6] == r2
    assert lines[7].split() == ["60", "40", str(int(mx.max())), str(int(mx.sum()))]


@have_ref
def test_product_ingest_against_the_reference_parser(tmp_path):
    """The PRODUCT's host ingest (kat_amd.parse_file: what katgpu_count ships to the GPU), counted window by window in Python,
    is the k-mer multiset of the reference's parser + iterator -- no oracle in between."""
    from tests import naive
    rng = np.random.default_rng(21)
    fa, fq, fqm = tmp_path / "m.fa", tmp_path / "m.fq", tmp_path / "mm.fq"
    write_messy_fasta(str(fa), rng, n_rec=12)
    write_messy_fastq(str(fq), rng, n_rec=80)
    write_messy_fastq(str(fqm), rng, n_rec=80, multiline=True)
    trap = tmp_path / "trap.fa"
    trap.write_bytes(b">a\n\n>ACGTACGTTG\nGGAACCTT\n>c\r\nTTGGCCAAGT\r\n")
    for p in (fa, fq, fqm, trap):
        stream = kat_amd.parse_file(str(p)).tobytes().decode("latin-1")
        for k, canonical in ((6, True), (13, False)):
            got = naive.count_string(stream, k, canonical)
            text = "".join("%s %d\n" % (w, c) for w, c in sorted(got.items())).encode()
            assert text == ref_kmers([str(p)], k, canonical), (p, k, canonical)


@have_ref
def test_documented_deviations_on_truncated_fastq(ko, tmp_path):
    """DESIGN.md 4, "Known deviations": what the reference really does there.  Its producer thread throws "Invalid fastq sequence"
    and the pool swallows it together with the buffer being filled (here the whole 4 KB file): no k-mers, exit status 0.  The
    product counts the complete records of a file that merely lacks its final newline, and REPORTS a truncated record."""
    nf, tr = tmp_path / "nf.fq", tmp_path / "tr.fq"
    nf.write_bytes(b"@r\nACGTA\n+\nIIIII\n@s\nGGACC\n+\nIIIII")            # complete, no final newline
    tr.write_bytes(b"@r\nACGTA\n+\nIIIII\n@s\nGGACC\n+\nIII")              # quality line cut short
    assert ref_kmers([str(nf)], 3, False) == b"" and ref_kmers([str(tr)], 3, False) == b""
    assert kat_amd.parse_file(str(nf)).tobytes() == b"ACGTANGGACC" == ko.parse_file(str(nf)).tobytes()
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.parse_file(str(tr))
    assert ei.value.code == 4


def ref_kmers_trim(paths, k, canonical, trims):
    rc, out = ref(JF_REF, ["kmerst", k, int(canonical), ",".join(map(str, trims))] + list(paths))
    assert rc == 0
    return out


@have_ref
def test_five_prime_trim_against_the_reference_parser(ko, tmp_path):
    """--5ptrim through the trim5p_list constructor KAT added to the vendored parser.  FASTQ: identical.  FASTA: identical while
    the file fits one of the parser's 4096-byte buffers; beyond that the reference re-applies the trim at EVERY buffer fill, in the
    middle of records (read_fasta sets newread = true on entry, mer_overlap_sequence_parser.hpp:198), and joins the seam to what
    follows the dropped bases -- quirk B7, the one place where the oracle and the product deliberately trim once per record."""
    rng = np.random.default_rng(2)
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    fq = tmp_path / "t.fq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(rnd(int(rng.integers(12, 90))) for _ in range(400))))
    small = tmp_path / "small.fa"
    small.write_text("".join(">c%d\n%s\n" % (i, "\n".join(rnd(60) for _ in range(4))) for i in range(12)))       # 2.9 KB of sequence
    big = tmp_path / "big.fa"
    big.write_text(">a\n" + rnd(9000) + "\n>b\n" + "\n".join(rnd(70) for _ in range(200)) + "\n")
    def oracle(paths, k, c, trims):
        keys, counts = ko.Table(k, c).count_files(paths, trims).dump_sorted()
        return "".join("%s %d\n" % (ko.decode(int(a), k), int(b)) for a, b in zip(keys, counts)).encode()
    for trim in (1, 3, 10):
        for k, c in ((7, True), (21, False)):
            assert oracle([str(fq)], k, c, [trim]) == ref_kmers_trim([str(fq)], k, c, [trim])
            assert oracle([str(small)], k, c, [trim]) == ref_kmers_trim([str(small)], k, c, [trim])
            assert oracle([str(fq), str(small)], k, c, [trim, 0]) == ref_kmers_trim([str(fq), str(small)], k, c, [trim, 0])
    mine, theirs = oracle([str(big)], 7, True, [3]), ref_kmers_trim([str(big)], 7, True, [3])
    assert mine != theirs                                               # the documented deviation, as the real parser shows it
    total = lambda b: sum(int(l.split()[1]) for l in b.splitlines())
    assert total(mine) == (9000 - 3 - 6) + (14000 - 3 - 6) and total(theirs) < total(mine)
    assert kat_amd.parse_file(str(big), 3).size == 9000 - 3 + 1 + 14000 - 3
