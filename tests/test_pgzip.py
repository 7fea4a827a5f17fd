"""One ordinary gzip stream inflated by a thread team (kat_amd/csrc/kg_pgzip.cpp; katgpu_inflate_file, and katgpu_parse_file /
katgpu_count_files for .gz files of size): the bytes must be zlib's and the base stream the streaming parser's, whatever the stream
is made of -- every block type, several members, entry points that are wrong guesses, data that is not text (no entry point at all),
records that straddle chunks -- and a corrupt or truncated file must be an error, as it is for the reference's gzstream
(deps/jellyfish-2.2.0/include/jellyfish/gzstream.hpp:121 through stream_manager.hpp:133-145).  No GPU needed."""
import gzip
import os
import zlib

import numpy as np
import pytest

import kat_amd
from kat_amd import binding as kb
from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq


def fastq_bytes(n, L=100, seed=1, crlf=False):
    rng = np.random.default_rng(seed)
    g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50000)]
    q = np.frombuffer(b"FFFFF:FF,FFFFFF#@+", np.uint8)       # '@' and '+' in the qualities: what a record-start guess must survive
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        s = int(rng.integers(0, g.size - L))
        out.append(b"@r%d/1 x" % i + nl + g[s:s + L].tobytes() + nl + b"+" + nl + q[rng.integers(0, q.size, L)].tobytes() + nl)
    return b"".join(out)


@pytest.fixture(autouse=True)
def team(monkeypatch):
    monkeypatch.setenv("KATGPU_PGZ_MIN_BYTES", "0")
    monkeypatch.setenv("KATGPU_PGZ_THREADS", "4")
    monkeypatch.setenv("KATGPU_INGEST_MIN_BYTES", str(1 << 60))              # (the plain twin of a file: through the streaming parser)


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush_mode=zlib.Z_SYNC_FLUSH):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for a in range(0, len(data), flush_every):
        out.append(co.compress(data[a:a + flush_every]))
        out.append(co.flush(flush_mode))
    out.append(co.flush())
    return b"".join(out)


SHAPES = {
    "level6": dict(level=6), "level1": dict(level=1), "level9": dict(level=9),
    "stored": dict(level=0),                                               # stored blocks only
    "fixed": dict(level=6, strategy=zlib.Z_FIXED),                         # fixed-Huffman blocks only: no entry point the search accepts
    "huffman_only": dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), "rle": dict(level=6, strategy=zlib.Z_RLE),
    "sync_flushes": dict(level=6, flush_every=70001),                      # pigz's shape: empty stored blocks between the pieces
    "full_flushes": dict(level=4, flush_every=33333, flush_mode=zlib.Z_FULL_FLUSH),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("chunk", [1 << 16, 300_000, 4 << 20])
def test_the_bytes_are_zlibs(tmp_path, monkeypatch, shape, chunk):
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(chunk))
    if shape in ("stored", "fixed"):
        monkeypatch.setenv("KATGPU_PGZ_PROBE", "0")                          # (no way into such a stream: the team is made to take it anyway -- the first decoder goes through it alone)
    data = fastq_bytes(12000, seed=3)
    p = tmp_path / "x.fastq.gz"
    p.write_bytes(deflate(data, **SHAPES[shape]))
    assert zlib.decompress(p.read_bytes(), 31) == data
    assert kb.inflate_file(str(p)) == data
    assert kb.inflate_file(str(p), keep=False) == len(data)


def test_members_headers_and_what_follows_the_last_member(tmp_path, monkeypatch):
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(1 << 16))
    parts = [fastq_bytes(n, seed=s) for n, s in ((3000, 1), (1, 2), (0, 3), (5000, 4), (20, 5))]
    blob = b""
    for i, part in enumerate(parts):                                        # (gzip.compress writes FNAME-less headers; one member gets the works)
        if i == 1:
            raw = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = raw.compress(part) + raw.flush()
            extra = b"AB\x03\x00xyz"
            hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\x03" + len(extra).to_bytes(2, "little") + extra + b"name.fq\0" + b"a comment\0"
            hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
            blob += hdr + body + zlib.crc32(part).to_bytes(4, "little") + (len(part) & 0xFFFFFFFF).to_bytes(4, "little")
        else:
            blob += gzip.compress(part, 6, mtime=0)
    whole = b"".join(parts)
    p = tmp_path / "m.gz"
    p.write_bytes(blob)
    assert gzip.decompress(blob) == whole
    assert kb.inflate_file(str(p)) == whole
    p.write_bytes(blob + b"\0\0\0 trailing bytes that are not gzip")          # zlib ignores them after a complete member
    assert kb.inflate_file(str(p)) == whole


@pytest.mark.parametrize("kind", ["random", "zeros", "runs"])
def test_data_that_is_not_text(tmp_path, monkeypatch, kind):
    """No block of such a file passes for an entry point (its literals are not text): the first chunk's decoder goes through the whole
    file, alone -- slow, and right.  Long runs: copies that overlap their own output, output hundreds of times the input."""
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(1 << 16))
    monkeypatch.setenv("KATGPU_PGZ_PROBE", "0")
    rng = np.random.default_rng(5)
    data = {"random": lambda: rng.integers(0, 256, 700000, dtype=np.uint8).tobytes(), "zeros": lambda: bytes(30_000_000),
            "runs": lambda: b"".join(bytes([int(x)]) * int(n) for x, n in zip(rng.integers(0, 256, 4000), rng.integers(1, 6000, 4000)))}[kind]()
    p = tmp_path / "b.gz"
    p.write_bytes(gzip.compress(data, 6, mtime=0))
    assert kb.inflate_file(str(p)) == data


def test_text_runs_expand_beyond_any_first_guess_of_the_buffers(tmp_path, monkeypatch):
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(1 << 16))
    monkeypatch.setenv("KATGPU_PGZ_PROBE", "0")                              # (blocks of millions of symbols are not entry points the search accepts: the team is made to take the file)
    data = (b"ACGT" * 25 + b"\n") * 600000                                  # 60 MB from ~200 KB: every chunk's output outgrows its buffer
    p = tmp_path / "r.gz"
    p.write_bytes(gzip.compress(data, 6, mtime=0))
    assert kb.inflate_file(str(p)) == data


def test_corrupt_and_truncated_files_are_errors(tmp_path, monkeypatch):
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(1 << 16))
    data = fastq_bytes(8000, seed=9)
    blob = gzip.compress(data, 6, mtime=0)
    p = tmp_path / "c.fastq.gz"
    for what, bad in (("crc", blob[:-8] + bytes([blob[-8] ^ 1]) + blob[-7:]), ("isize", blob[:-4] + bytes([blob[-4] ^ 1]) + blob[-3:]),
                      ("truncated", blob[:len(blob) * 2 // 3]), ("no trailer", blob[:-5]), ("a flipped bit", blob[:len(blob) // 2] + bytes([blob[len(blob) // 2] ^ 0x10]) + blob[len(blob) // 2 + 1:])):
        p.write_bytes(bad)
        with pytest.raises(kat_amd.KatGpuError) as e:
            kb.inflate_file(str(p))
        assert e.value.code == 2 and "read error on" in str(e.value), what
        with pytest.raises(kat_amd.KatGpuError):                            # ... and through the parser, whichever way it reads the file
            kat_amd.parse_file(str(p))
    p.write_bytes(b"@r\nACGT\n+\nFFFF\n")
    with pytest.raises(kat_amd.KatGpuError) as e:
        kb.inflate_file(str(p))
    assert e.value.code == 3


@pytest.mark.parametrize("chunk", [1 << 16, 250_000])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_the_base_stream_is_the_streaming_parsers(ko, tmp_path, monkeypatch, capfd, chunk, seed):
    """FASTA and FASTQ of every shape the parser tests hold -- blank lines, multi-line FASTQ, '@' qualities, CRLF, no final newline --
    gzipped: chunks whose guessed record start is wrong go through the state machine serially; the stream never changes."""
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(chunk))
    monkeypatch.setenv("KATGPU_TRACE", "1")
    files = []
    fa, fq = tmp_path / "m.fa", tmp_path / "m.fq"
    rng = np.random.default_rng(seed)
    fqm = tmp_path / "multi.fq"
    write_messy_fasta(str(fa), rng, n_rec=900)
    write_messy_fastq(str(fq), rng, n_rec=2500)
    write_messy_fastq(str(fqm), rng, n_rec=2500, multiline=True)
    files += [fa, fq, fqm]
    clean = tmp_path / "clean.fq"
    clean.write_bytes(fastq_bytes(20000, seed=seed))
    crlf = tmp_path / "crlf.fq"
    crlf.write_bytes(fastq_bytes(6000, seed=seed, crlf=True)[:-1])            # (and no final newline)
    files += [clean, crlf]
    took = 0
    for f in files:
        plain = f.read_bytes()
        want = kat_amd.parse_file(str(f)).tobytes()
        assert want == ko.parse_file(str(f)).tobytes()
        for level in (1, 6):
            gz = tmp_path / (f.name + ".gz")
            gz.write_bytes(gzip.compress(plain, level, mtime=0))
            capfd.readouterr()
            got = kat_amd.parse_file(str(gz)).tobytes()
            err = capfd.readouterr().err                                    # (the team took it -- or found no way into so small a stream and said so)
            assert "one gzip stream" in err or "one zlib stream" in err, err[-400:]
            took += "one gzip stream" in err
            assert got == want, (f.name, level)
            monkeypatch.setenv("KATGPU_PGZ", "0")
            assert kat_amd.parse_file(str(gz)).tobytes() == want            # zlib's stream: the same
            monkeypatch.delenv("KATGPU_PGZ")
    assert took >= 6                                                        # (most of the ten files are the team's)


def test_which_files_the_team_takes(tmp_path, monkeypatch, capfd):
    monkeypatch.setenv("KATGPU_TRACE", "1")
    data = fastq_bytes(3000, seed=4)
    gz = tmp_path / "a.fastq.gz"
    gz.write_bytes(gzip.compress(data, 6, mtime=0))
    want = kat_amd.parse_file(str(gz)).tobytes()
    assert "one gzip stream" in capfd.readouterr().err
    monkeypatch.setenv("KATGPU_PGZ_STRIP", "0")                             # the chunks' records through the state machine instead of the strip loop: the same stream
    assert kat_amd.parse_file(str(gz)).tobytes() == want and "one gzip stream" in capfd.readouterr().err
    monkeypatch.delenv("KATGPU_PGZ_STRIP")
    monkeypatch.setenv("KATGPU_PGZ_MIN_BYTES", str(8 << 20))                # the default: a small file is not worth a team
    assert kat_amd.parse_file(str(gz)).tobytes() == want and "one gzip stream" not in capfd.readouterr().err
    monkeypatch.setenv("KATGPU_PGZ_MIN_BYTES", "0")
    assert kat_amd.parse_file(str(gz), trim5p=3).tobytes() != want and "one gzip stream" not in capfd.readouterr().err     # 5' trim: the streaming parser
    # a group: the .gz files of size one after the other through their teams, the k-mer multiset that of the files read one by one
    gz2 = tmp_path / "b.fastq.gz"
    gz2.write_bytes(gzip.compress(fastq_bytes(2000, seed=8), 1, mtime=0))
    both = kat_amd.parse_files([str(gz), str(gz2)], 21).tobytes()
    assert both == want + b"N" + kat_amd.parse_file(str(gz2)).tobytes() + b"N"


def test_a_stream_without_a_way_in_is_left_to_zlib(tmp_path, monkeypatch, capfd):
    """Fixed-Huffman or stored blocks only: no block start the search accepts, so the first chunk's decoder would go through the whole file
    alone, its output growing with it.  The team looks for one entry point behind the first chunk before it starts and declines such a file:
    the parser reads it through zlib's stream, the same bytes; katgpu_inflate_file says it is not a file the team takes."""
    monkeypatch.setenv("KATGPU_PGZ_CHUNK", str(1 << 16))
    monkeypatch.setenv("KATGPU_TRACE", "1")
    data = fastq_bytes(9000, seed=6)
    plain = tmp_path / "p.fastq"
    plain.write_bytes(data)
    want = kat_amd.parse_file(str(plain)).tobytes()
    for shape in ("fixed", "stored"):
        gz = tmp_path / (shape + ".fastq.gz")
        gz.write_bytes(deflate(data, **SHAPES[shape]))
        capfd.readouterr()
        assert kat_amd.parse_file(str(gz)).tobytes() == want
        err = capfd.readouterr().err
        assert "one zlib stream" in err and "by a team of" not in err, err[-600:]
        with pytest.raises(kat_amd.KatGpuError) as e:
            kb.inflate_file(str(gz))
        assert e.value.code == 3
