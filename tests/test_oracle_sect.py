"""`kat sect` in the C oracle (oracle/koracle_sect.c) against the independent pure-Python statement in tests/naive.py, on
the reference's own sect inputs (tests/data/sect_test.fa, sect_length_test.fa, used by tests/test_sect.sh) and on generated
edge cases.  The reference holds no golden sect outputs, so this cross-check is what pins the restatement."""
import gzip
import os
import random

import pytest

from tests import naive


def oracle_files(ko, table, seq_path, tmp_path, tag, **kw):
    prefix = str(tmp_path / tag)
    ko.sect(table, seq_path, prefix, **kw)
    out = {}
    for suffix in ("-counts.cvg", "-counts.gc", "-non_repetitive.fa", "-repetitive.fa", "-stats.tsv", "-contamination.mx"):
        if os.path.exists(prefix + suffix):
            out[suffix] = open(prefix + suffix, "rb").read()
    return out


def counts_of(ko, table):
    if isinstance(table, ko.WideTable):
        hi, lo, cnts = table.dump_sorted()
        k = table.k
        dec = lambda x: "".join("ACGT"[(x >> (2 * (k - 1 - i))) & 3] for i in range(k))
        return {dec((int(a) << 64) | int(b)): int(c) for a, b, c in zip(hi, lo, cnts)}
    keys, cnts = table.dump_sorted()
    return {ko.decode(int(k), table.k): int(c) for k, c in zip(keys, cnts)}


def test_reference_cli_case(ko, refdata, tmp_path):
    """tests/test_sect.sh: sect_length_test.fa against ecoli.header.jf27 (k = 27, canonical)."""
    t = ko.Table.from_jf(os.path.join(refdata, "ecoli.header.jf27"))
    counts = counts_of(ko, t)
    for name in ("sect_length_test.fa", "sect_test.fa"):
        p = os.path.join(refdata, name)
        want = naive.sect(counts, 27, True, p, output_gc_stats=True, extract_nr=True, extract_r=True, save=True)
        got = oracle_files(ko, t, p, tmp_path, name, output_gc_stats=True, extract_nr=True, extract_r=True, save=True)
        assert got == want
    # every sect_test.fa record is shorter than k = 27 or barely longer: the short-record branch and the uint32 wrap of kmers_in_seq
    stats = want["-stats.tsv"].decode().splitlines()
    assert stats[2].split("\t")[4:6] == ["13", str(2**32 + 13 - 27 + 1)]


@pytest.mark.parametrize("k,canonical", [(5, True), (5, False), (11, True), (31, True)])
def test_self_coverage(ko, refdata, tmp_path, k, canonical):
    """Hash counted from the sequence file itself: non-trivial counts, repeats, N runs."""
    p = os.path.join(refdata, "sect_test.fa")
    t = ko.Table(k, canonical).count_files([p])
    counts = counts_of(ko, t)
    for kw in (dict(), dict(output_gc_stats=True, extract_nr=True, extract_r=True), dict(no_count_stats=True, extract_r=True, min_repeat=1, max_repeat=2),
               dict(extract_nr=True, min_repeat=3, gc_bins=10, cvg_bins=4, save=True)):
        want = naive.sect(counts, k, canonical, p, **kw)
        got = oracle_files(ko, t, p, tmp_path, "s%d%d%d" % (k, canonical, len(kw)), **kw)
        assert got == want


def make_cases(tmp_path):
    rng = random.Random(7)
    unit = "".join(rng.choice("ACGT") for _ in range(40))
    recs = [("r0 plain description", unit * 3), ("lower", unit.lower() + "acgtnnnnACGT" + unit), ("allN", "N" * 30), ("empty", ""),
            ("short", "ACG"), ("odd chars", unit[:20] + "RYK-*" + unit[20:] + " " + unit), ("gc", "GC" * 25), ("at>", "AT" * 25)]
    fa = tmp_path / "cases.fa"
    with open(fa, "w") as f:
        for name, seq in recs:
            f.write(">" + name + "\n")
            for i in range(0, len(seq), 17):
                f.write(seq[i:i + 17] + "\n")
        f.write("\n\n")
    crlf = tmp_path / "crlf.fasta"
    crlf.write_bytes(b"junk before the first record\r\n" + fa.read_bytes().replace(b"\n", b"\r\n"))
    fq = tmp_path / "reads.fq"
    with open(fq, "w") as f:
        for i in range(6):
            s = unit[i:i + 30] + ("N" if i % 2 else "") + unit[:10]
            f.write("@read%d/1\n%s\n+read%d/1\n%s\n" % (i, s, i, "@" * len(s)))      # '@' qualities must not start a record
    gz = tmp_path / "cases.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fa.read_bytes())
    raw = tmp_path / "lines.txt"                      # SeqAn's Raw format: one nameless record per line
    raw.write_bytes(b"\n".join((unit * 2)[i:i + 45].encode() for i in range(0, 60, 7)) + b"\n\nACGTNNACGT" + unit.encode() + b"\r\n")
    upper = tmp_path / "CASES.FA"                     # extensions are matched case-insensitively
    upper.write_bytes(fa.read_bytes())
    return [str(fa), str(crlf), str(fq), str(gz), str(raw), str(upper)], str(fa)


def test_generated_edge_cases(ko, tmp_path):
    paths, fa = make_cases(tmp_path)
    for k, canonical in ((7, True), (21, False)):
        t = ko.Table(k, canonical).count_files([fa])
        counts = counts_of(ko, t)
        for i, p in enumerate(paths):
            kw = dict(output_gc_stats=True, extract_nr=True, extract_r=True, min_repeat=2, max_repeat=5, save=True)
            want = naive.sect(counts, k, canonical, p, **kw)
            got = oracle_files(ko, t, p, tmp_path, "g%d%d" % (k, i), **kw)
            assert got == want, p
    # '>' inside a sequence line ends the record there (SeqAn reads to the next '>' wherever it stands)
    assert [n for n, _ in naive.seqan_records(fa)][-1] == b"at>"
    assert b"-nan" in want["-stats.tsv"]          # the all-N and empty records: 0/0 GC%


def test_reader_errors(ko, tmp_path):
    t = ko.Table(5, True)
    bad = tmp_path / "x.fa"
    bad.write_bytes(b"no record marker here\n")
    with pytest.raises(ko.OracleError):
        ko.sect(t, str(bad), str(tmp_path / "o"))
    with pytest.raises(ValueError):
        naive.seqan_records(str(bad))
    other = tmp_path / "contigs.fna"                  # SeqAn decides on the name alone: UnknownExtensionError (KAT exits with 5)
    other.write_bytes(b">x\nACGTACGT\n")
    with pytest.raises(ko.OracleError):
        ko.sect(t, str(other), str(tmp_path / "o"))
    with pytest.raises(ValueError, match="Unknown file extension"):
        naive.seqan_records(str(other))
    empty = tmp_path / "e.fa"
    empty.write_bytes(b"")
    ko.sect(t, str(empty), str(tmp_path / "e"))
    assert (tmp_path / "e-stats.tsv").read_bytes().count(b"\n") == 1
    assert (tmp_path / "e-counts.cvg").read_bytes() == b""


def test_cold_oracle_vs_naive(ko, refdata, tmp_path):
    """`kat cold`: reads hash + assembly hash, both non-canonical when counted (Cold never sets InputHandler::canonical)."""
    paths, fa = make_cases(tmp_path)
    r1 = os.path.join(refdata, "ecoli_r1.1K.fastq")
    for k, cr, ca in ((7, False, False), (15, True, False), (27, False, True)):
        reads = ko.Table(k, cr).count_files([r1, paths[2]])
        asm = ko.Table(k, ca).count_files([fa, os.path.join(refdata, "sect_test.fa")])
        for p in (paths[0], paths[1], paths[3], os.path.join(refdata, "sect_test.fa")):
            ko.cold(reads, asm, p, str(tmp_path / "c"))
            want = naive.cold(counts_of(ko, reads), cr, counts_of(ko, asm), ca, k, p)
            assert (tmp_path / "c-stats.tsv").read_bytes() == want


@pytest.mark.parametrize("k,canonical", [(33, True), (40, False), (63, True)])
def test_wide_sect_and_cold_vs_naive(ko, refdata, tmp_path, k, canonical):
    """k > 32 (koracle_wide.c behind the same sect / cold code) against the naive statement, which works on strings."""
    paths, fa = make_cases(tmp_path)
    length = os.path.join(refdata, "sect_length_test.fa")
    t = ko.WideTable(k, canonical).count_files([fa, length, os.path.join(refdata, "ecoli_r1.1K.fastq")])
    counts = counts_of(ko, t)
    for i, p in enumerate(paths[:4] + [length]):
        kw = dict(output_gc_stats=True, extract_nr=True, extract_r=True, min_repeat=2, max_repeat=5, save=True)
        assert oracle_files(ko, t, p, tmp_path, "w%d" % i, **kw) == naive.sect(counts, k, canonical, p, **kw), p
    asm = ko.WideTable(k, not canonical).count_files([fa, length])
    for p in (paths[0], paths[3], length):
        ko.cold(t, asm, p, str(tmp_path / "wc"))
        assert (tmp_path / "wc-stats.tsv").read_bytes() == naive.cold(counts, canonical, counts_of(ko, asm), not canonical, k, p)


SEQAN_REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "seqan_ref")


@pytest.mark.skipif(not os.access(SEQAN_REF, os.X_OK), reason="oracle/_ref not built (no /root/reference at build time)")
def test_record_reader_against_the_real_seqan(tmp_path):
    """The records `kat sect` / `kat cold` see: tests/naive.py's statement of the SeqAn 2.0.0 reader against the library itself
    (oracle/_ref/seqan_ref, compiled from the reference's vendored headers), on every edge case used above."""
    import subprocess
    paths, fa = make_cases(tmp_path)
    extra = {"mid.fasta": b"leading junk\r\n>r0 desc\r\nACGT\r\nAC GT\r\n>r1\r\nTT>GG\r\n\r\n", "two.fq": b"@a\nACGT\n+\n@III\n@b\nGG\nTT\n+b\n@@\n@@\n",
             "empty.fa": b"", "x.fna": b">x\nACGT\n", "nomarker.fa": b"no marker\n", "noplus.fq": b"@a\nACGT\n"}
    for name, data in extra.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    for p in paths:
        r = subprocess.run([SEQAN_REF, p], capture_output=True, timeout=60)
        try:
            want = b"".join(b"%d %s\n%d %s\n" % (len(n), n, len(s), s) for n, s in naive.seqan_records(p))
        except ValueError as e:
            assert r.returncode == 5 and r.stdout.startswith(b"EXCEPTION " + str(e).encode()[:20]), (p, r.stdout[:200])
            continue
        assert r.returncode == 0 and r.stdout == want, p
