"""The product's host ingest (kat_amd/csrc/kg_ingest.cpp, reached through the C ABI's katgpu_parse_file -- no GPU needed)
against the oracle's parser and the naive record splitter: the base stream must be the records joined by 'N'."""
import gzip
import os

import numpy as np
import pytest

import kat_amd
from tests import naive
from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq


# Every test runs four times: through the streaming parser, and through the thread team (kg_ingest.hpp:
# parse_file_parallel) with segment sizes that cut these small files into many pieces -- guessed record starts that are
# wrong (blank line after a header, '@' qualities, multi-line FASTQ) must be caught and the output must not change.
@pytest.fixture(autouse=True, params=["stream", "seg37", "seg1000_margin64", "seg200k"])
def ingest_mode(request, monkeypatch):
    mode = request.param
    if mode == "stream":
        monkeypatch.setenv("KATGPU_INGEST_MIN_BYTES", str(1 << 60))
    else:
        seg, margin = {"seg37": (37, 4096), "seg1000_margin64": (1000, 64), "seg200k": (200_000, 1 << 20)}[mode]
        monkeypatch.setenv("KATGPU_INGEST_MIN_BYTES", "0")
        monkeypatch.setenv("KATGPU_INGEST_SEGMENT", str(seg))
        monkeypatch.setenv("KATGPU_INGEST_MARGIN", str(margin))
        monkeypatch.setenv("KATGPU_INGEST_THREADS", "5")
    return mode


def stream_of(path):
    return kat_amd.parse_file(str(path)).tobytes()


def test_reference_data_files(ko, refdata):
    for f in ("sect_test.fa", "sect_length_test.fa", "ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq"):
        p = os.path.join(refdata, f)
        got = stream_of(p)
        assert got == ko.parse_file(p).tobytes()
        assert got == "N".join(naive.records(p)).encode("latin-1")


@pytest.mark.parametrize("seed", range(4))
def test_messy_files(ko, tmp_path, seed):
    rng = np.random.default_rng(seed)
    fa, fq, fqm, gz = tmp_path / "m.fa", tmp_path / "m.fq", tmp_path / "mm.fq", tmp_path / "z.fq.gz"
    write_messy_fasta(str(fa), rng)
    write_messy_fastq(str(fq), rng)
    write_messy_fastq(str(fqm), rng, multiline=True)
    write_messy_fastq(str(gz), rng)
    for p in (fa, fq, fqm, gz):
        got = stream_of(p)
        assert got == ko.parse_file(str(p)).tobytes(), p
        assert got == "N".join(naive.records(str(p))).encode("latin-1"), p


def test_edge_files(ko, tmp_path):
    cases = {
        "empty.fa": b"",
        "header_only.fa": b">x\n",
        "no_newline.fa": b">x\nACGT",
        "crlf.fa": b">x\r\nACGT\r\nACGT\r\n>y\r\nTT\r\n",               # '\r' stays in the stream and breaks k-mers (quirk B8)
        "blank_after_header.fa": b">a\n\n>ACGTACGT\nGG\n>c\nTT\n",       # blank line after a header: next line is sequence
        "gt_inside.fa": b">a\nAC>GT\nAA\n",
        "one.fq": b"@r\nACGT\n+\nIIII\n",
        "no_final_newline.fq": b"@r\nACGT\n+\nIIII\n@s\nGG\n+\nII",       # tolerated (documented deviation)
        "at_quality.fq": b"@r\nACGT\n+\n@III\n@s\nGGA\n+\n@@@\n",
        "long_line.fa": b">x\n" + b"ACGT" * 6_000_000 + b"\n>y\nGATTACA\n",   # one 24 MB line: spans raw blocks
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        got = stream_of(p)
        assert got == ko.parse_file(str(p)).tobytes(), name
    assert stream_of(tmp_path / "crlf.fa") == b"ACGT\rACGT\rNTT\r"
    assert stream_of(tmp_path / "blank_after_header.fa") == b">ACGTACGTGGNTT"
    assert stream_of(tmp_path / "no_final_newline.fq") == b"ACGTNGG"
    assert stream_of(tmp_path / "at_quality.fq") == b"ACGTNGGA"


def test_trim5p(ko, tmp_path):
    p = tmp_path / "t.fq"
    p.write_bytes(b"@a\nACGTACGT\n+\nIIIIIIII\n@b\nTTGGCCAA\n+\nIIIIIIII\n")
    assert kat_amd.parse_file(str(p), 3).tobytes() == b"TACGTNGCCAA" == ko.parse_file(str(p), 3).tobytes()
    f = tmp_path / "t.fa"
    f.write_bytes(b">a\nACGTACGT\nAAAA\n>b\nTTGGCCAA\n")
    assert kat_amd.parse_file(str(f), 2).tobytes() == b"GTACGTAAAANGGCCAA" == ko.parse_file(str(f), 2).tobytes()


def test_errors(ko, tmp_path):
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.parse_file(str(tmp_path / "nope.fa"))
    assert ei.value.code == 2 and "Could not find input file at" in ei.value.message
    bad = tmp_path / "bad.dat"
    bad.write_bytes(b"hello\n")
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.parse_file(str(bad))
    assert ei.value.code == 3 and ei.value.message == "Unsupported format"
    with pytest.raises(ko.OracleError) as oi:
        ko.parse_file(str(bad))
    assert oi.value.code == 2
    for name, data in (("short_q.fq", b"@r\nACGTACGT\n+\nIIII\n@s\nAC\n+\nII\n"), ("trunc.fq", b"@r\nACGT\n+\nII"),
                       ("long_q.fq", b"@r\nACGT\n+\nIIIIIIII\n@s\nAC\n+\nII\n"), ("noplusq.fq", b"@r\nACGT\n+\n"),
                       # an empty read: the blank line after the header makes the reference's parser read the '+' line as
                       # sequence (mer_overlap_sequence_parser.hpp:254-258), and the record then fails the quality-length check
                       ("empty_read.fq", b"@r\n\n+\n\n@s\nGGA\n+\nIII\n")):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(kat_amd.KatGpuError) as ei:
            kat_amd.parse_file(str(p))
        assert ei.value.code == 4 and ei.value.message == "Invalid fastq sequence", name
        with pytest.raises(ko.OracleError) as oi:
            ko.parse_file(str(p))
        assert oi.value.code == 3, name


def test_team_accepts_clean_files_and_falls_back_on_wrong_guesses(ko, tmp_path, ingest_mode, monkeypatch, capfd):
    """What the thread team reports (KATGPU_TRACE): clean files are parsed piece by piece; a file whose guessed record starts
    are wrong is handed to the streaming machine from the last state known to be right."""
    if ingest_mode != "seg1000_margin64":
        pytest.skip("one parallel geometry is enough here")
    monkeypatch.setenv("KATGPU_TRACE", "1")
    monkeypatch.setenv("KATGPU_INGEST_MARGIN", "4096")
    rng = np.random.default_rng(5)
    recs = ["".join(rng.choice(list("ACGTN"), int(rng.integers(30, 200)))) for _ in range(400)]
    fa, fq = tmp_path / "clean.fa", tmp_path / "clean.fq"
    fa.write_text("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(recs)))
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, r, "@" * len(r)) for i, r in enumerate(recs)))      # qualities made of '@'
    for p in (fa, fq):
        capfd.readouterr()
        assert stream_of(p) == "N".join(recs).encode()
        err = capfd.readouterr().err
        assert "all accepted" in err and "streaming from offset" not in err, err
    trap = tmp_path / "trap.fa"                      # every record: header, blank line, then a SEQUENCE line that starts with '>'
    trap.write_text("".join(">h%d\n\n>%s\n" % (i, r) for i, r in enumerate(recs)))
    capfd.readouterr()
    assert stream_of(trap) == ko.parse_file(str(trap)).tobytes() == "N".join(">" + r for r in recs).encode()
    assert "streaming from offset" in capfd.readouterr().err


# ---------------------------------------------------------------- one input group (kg_ingest.hpp: stream_group) ---------

def _group(tmp_path, rng):
    fa, fq, fqm = tmp_path / "g.fa", tmp_path / "g.fq", tmp_path / "gm.fq"
    write_messy_fasta(str(fa), rng)
    write_messy_fastq(str(fq), rng)
    write_messy_fastq(str(fqm), rng, multiline=True)
    z1, z2 = tmp_path / "z1.fq.gz", tmp_path / "z2.fa.gz"
    write_messy_fastq(str(z1), rng)
    with gzip.open(z2, "wb") as f:
        f.write(fa.read_bytes())
    tiny = tmp_path / "tiny.fa"
    tiny.write_bytes(b">t\nACGTACGTACGTTTGACCA\n")
    clean = tmp_path / "c.fq.gz"                                       # reads long enough for a 5' trim (the messy ones are not)
    with gzip.open(clean, "wb") as f:
        for i in range(300):
            r = "".join(rng.choice(list("ACGTN"), int(rng.integers(30, 120)), p=[.24, .24, .24, .24, .04]))
            f.write(("@c%d\n%s\n+\n%s\n" % (i, r, "I" * len(r))).encode())
    return [str(p) for p in (z1, fa, z2, tiny, clean, fqm, z1)]       # a file may be named twice in a group


def _kmer_table(ko, k, stream):
    return ko.Table(k, True).count_bases(stream).dump_sorted()


@pytest.mark.parametrize("files_at_once,block", [(1, 4 << 20), (2, 61), (8, 7), (8, 1000), (3, 4 << 20)])
def test_group_stream_keeps_the_kmer_multiset(ko, tmp_path, monkeypatch, files_at_once, block):
    """Files of a group that have to stream are read concurrently and their blocks interleave (with 'N' + the file's last k-1
    bytes at each switch): the stream's k-mers, counted by the oracle, are those of the files counted one after the other."""
    monkeypatch.setenv("KATGPU_INGEST_FILES", str(files_at_once))
    monkeypatch.setenv("KATGPU_INGEST_BLOCK", str(block))
    paths = _group(tmp_path, np.random.default_rng(11))
    trims = [0, 1, 2, 0, 3, 0, 0]
    for k in (5, 21, 31):
        for tr in (None, trims):
            stream = kat_amd.parse_files(paths, k, tr)
            if tr is None:
                want = ko.Table(k, True).count_files(paths).dump_sorted()
            else:      # the oracle's FASTA 5' trim is the documented per-record one, as the product's: count the product's own per-file streams
                want = _kmer_table(ko, k, b"N".join(kat_amd.parse_file(p, t).tobytes() for p, t in zip(paths, tr)))
            got = _kmer_table(ko, k, stream)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (k, tr)
            if files_at_once == 1:
                assert stream.tobytes() == b"".join(kat_amd.parse_file(p, t).tobytes() + b"N" for p, t in zip(paths, tr or [0] * len(paths)))


def test_group_errors_name_the_first_bad_file(tmp_path, monkeypatch):
    monkeypatch.setenv("KATGPU_INGEST_BLOCK", "50")
    good = tmp_path / "good.fa"
    good.write_bytes(b"".join(b">r%d\n%s\n" % (i, b"ACGT" * 40) for i in range(200)))
    badq = tmp_path / "bad.fq"
    badq.write_bytes(b"".join(b"@r%d\nACGTACGT\n+\nIIIIIIII\n" % i for i in range(100)) + b"@x\nACGT\n+\nII")
    junk = tmp_path / "junk.dat"
    junk.write_bytes(b"hello\n")
    missing = tmp_path / "missing.fq"
    for n in (1, 8):
        monkeypatch.setenv("KATGPU_INGEST_FILES", str(n))
        for order, code, text in (([good, badq, junk, missing], 4, "Invalid fastq sequence"),
                                  ([good, junk, badq, good], 3, "Unsupported format"),
                                  ([missing, badq, junk], 2, "Could not find input file at"),
                                  ([good, good, good, missing], 2, "Could not find input file at")):
            with pytest.raises(kat_amd.KatGpuError) as ei:
                kat_amd.parse_files([str(p) for p in order], 21)
            assert ei.value.code == code and text in ei.value.message, (n, order)
    assert kat_amd.parse_files([], 21).size == 0


# ---------------------------------------------------------------- BGZF (kg_ingest.hpp: parse_bgzf_parallel) -------------

def bgzf_bytes(data, rng, lo=1, hi=3000, eof=True):
    """`data` as a BGZF file: independent gzip members of random sizes, each with the 'BC' extra field, then the empty EOF member."""
    import struct
    import zlib
    out, i = [], 0
    sizes = []
    while i < len(data):
        n = int(rng.integers(lo, hi + 1))
        sizes.append(n)
        chunk = data[i:i + n]
        i += n
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    if eof:
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


@pytest.fixture
def bgzf_env(monkeypatch):
    monkeypatch.setenv("KATGPU_BGZF_MIN_BYTES", "0")
    monkeypatch.setenv("KATGPU_BGZF_THREADS", "6")
    monkeypatch.setenv("KATGPU_BGZF_WINDOW", str(1 << 17))            # the smallest window: many rounds even for these small files
    monkeypatch.setenv("KATGPU_TRACE", "1")


@pytest.mark.parametrize("seed", range(3))
def test_bgzf_files_match_the_streaming_parser(ko, tmp_path, bgzf_env, capfd, monkeypatch, seed):
    """bgzip-style files go through the inflate team; the stream is the one zlib + the streaming parser produce (checked against
    the oracle's parser on the same file, which reads it through zlib)."""
    rng = np.random.default_rng(seed)
    fa, fq, fqm = tmp_path / "a.fa", tmp_path / "a.fq", tmp_path / "am.fq"
    write_messy_fasta(str(fa), rng, n_rec=200)
    write_messy_fastq(str(fq), rng, n_rec=3000)
    write_messy_fastq(str(fqm), rng, n_rec=1500, multiline=True)
    for src in (fa, fq, fqm):
        data = src.read_bytes()
        for tag, lo, hi in (("small", 1, 300), ("big", 20000, 65280)):
            z = tmp_path / (src.name + "." + tag + ".gz")
            z.write_bytes(bgzf_bytes(data, rng, lo, hi))
            assert gzip.decompress(z.read_bytes()) == data
            capfd.readouterr()
            got = stream_of(z)
            assert "BGZF members by the team" in capfd.readouterr().err
            assert got == ko.parse_file(str(z)).tobytes() == stream_of(src), (src, tag)
    clean = tmp_path / "c.fa"
    clean.write_bytes(b"".join(b">r%d\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGT", np.uint8), 90).tobytes()) for i in range(3000)))
    z = tmp_path / "c.fa.gz"
    z.write_bytes(bgzf_bytes(clean.read_bytes(), rng, 100, 5000))
    for trim in (0, 1, 7):
        assert kat_amd.parse_file(str(z), trim).tobytes() == kat_amd.parse_file(str(clean), trim).tobytes()
    monkeypatch.setenv("KATGPU_BGZF", "0")                          # switched off: same bytes through zlib
    capfd.readouterr()
    assert stream_of(z) == stream_of(clean)
    assert "BGZF" not in capfd.readouterr().err


def test_bgzf_edges_and_fallbacks(ko, tmp_path, bgzf_env, capfd):
    rng = np.random.default_rng(9)
    recs = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGTN", np.uint8), 70).tobytes(), b"I" * 70) for i in range(2000))
    plain = tmp_path / "p.fq"
    plain.write_bytes(recs)
    want = stream_of(plain)
    # no EOF marker; only the EOF marker; EOF marker in the middle (two bgzip files concatenated)
    z = tmp_path / "noeof.fq.gz"
    z.write_bytes(bgzf_bytes(recs, rng, 50, 2000, eof=False))
    assert stream_of(z) == want
    z = tmp_path / "empty.fq.gz"
    z.write_bytes(bgzf_bytes(b"", rng))
    assert stream_of(z) == b"" == ko.parse_file(str(z)).tobytes()
    half = recs.index(b"@r1000\n")
    z = tmp_path / "cat.fq.gz"
    z.write_bytes(bgzf_bytes(recs[:half], rng, 50, 2000) + bgzf_bytes(recs[half:], rng, 50, 2000))
    assert stream_of(z) == want
    # a plain gzip member appended: the rest goes through zlib from that offset; then garbage, which zlib ignores
    z = tmp_path / "mixed.fq.gz"
    z.write_bytes(bgzf_bytes(recs[:half], rng, 50, 2000, eof=False) + gzip.compress(recs[half:]) + b"trailing garbage")
    capfd.readouterr()
    assert stream_of(z) == want == ko.parse_file(str(z)).tobytes()
    assert "the rest through zlib" in capfd.readouterr().err
    z = tmp_path / "garbage.fq.gz"
    z.write_bytes(bgzf_bytes(recs, rng, 50, 2000) + b"\x00\x01 not gzip at all")
    assert stream_of(z) == want == ko.parse_file(str(z)).tobytes()
    # damage: a flipped payload byte (CRC), a truncated last member
    good = bgzf_bytes(recs, rng, 500, 2000)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x55
    z = tmp_path / "crc.fq.gz"
    z.write_bytes(bytes(bad))
    with pytest.raises(kat_amd.KatGpuError) as ei:
        stream_of(z)
    assert ei.value.code in (2, 4)                                   # read error (or, if the flip landed in a header, what zlib makes of the rest)
    z = tmp_path / "trunc.fq.gz"
    z.write_bytes(good[:len(good) - 40])
    with pytest.raises(kat_amd.KatGpuError):
        stream_of(z)
    # a BGZF file that holds something else
    z = tmp_path / "junk.gz"
    z.write_bytes(bgzf_bytes(b"hello world\n" * 100, rng, 10, 100))
    with pytest.raises(kat_amd.KatGpuError) as ei:
        stream_of(z)
    assert ei.value.code == 3 and ei.value.message == "Unsupported format"
    # in a group with other files
    paths = [str(plain), str(tmp_path / "cat.fq.gz"), str(tmp_path / "noeof.fq.gz")]
    got = kat_amd.parse_files(paths, 21)
    assert np.array_equal(ko.Table(21, True).count_bases(got).dump_sorted()[1], ko.Table(21, True).count_files(paths).dump_sorted()[1])
