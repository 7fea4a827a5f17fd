"""Run in a subprocess by tests/test_gpu_scan.py with KATGPU_TEST_SCAN_* hooks in the environment (batches of a few KB, so that small
files cross hundreds of batch cuts): katgpu_count_files through the device-side record scan (kg_scan.hip) must build the table the
oracle builds from the HOST parser's base stream of the same file (katgpu_parse_file: the streaming state machine, itself pinned to
the reference's parser in tests/test_oracle_vs_reference.py) -- for well-formed files, where the device does the parsing, and for
files it must hand back to the host machine at some batch: CRLF, blank lines, multi-line FASTQ, a last line without newline.
With KATGPU_TEST_STRIP_SEGMENT set the FASTQ cases take the HOST STRIP instead (kg_scan.hip: read_loop_strip / run_strip -- the readers cut
the file at record starts, keep the sequence lines, the caller's thread commits the segments in file order), with segments of a few KB:
hundreds of cuts per file, the hand-over to the state machine wherever a segment is not plain four-line FASTQ."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import synth  # noqa: E402
from oracle import koracle as ko  # noqa: E402


def fastq(reads, rng, qual_at=True, header=lambda i: b"@r%d some text" % i):
    out = []
    for i, r in enumerate(reads):
        q = bytearray(rng.integers(33, 74, size=len(r), dtype=np.uint8).tobytes())
        if qual_at and len(q) and i % 3 == 0:
            q[0] = ord("@")                                   # a quality line that looks like a header
        if qual_at and len(q) and i % 5 == 0:
            q[0] = ord("+")
        out.append(header(i) + b"\n" + bytes(r) + b"\n+" + (b"" if i % 2 else b"r%d" % i) + b"\n" + bytes(q) + b"\n")
    return b"".join(out)


def fasta(contigs, width):
    out = []
    for i, c in enumerate(contigs):
        out.append(b">contig%d len=%d\n" % (i, len(c)))
        for j in range(0, len(c), width):
            out.append(bytes(c[j:j + width]) + b"\n")
    return b"".join(out)


def main():
    eng = kat_amd.Engine(0)
    rng = np.random.default_rng(5)
    g = synth.genome(120000, seed=21)
    stream = synth.reads(g, 0, 6000, seed=3).reshape(-1, 151)[:, :150]
    reads = [row.tobytes() for row in stream]
    ragged = [r[: int(rng.integers(1, 151))] for r in reads[:3000]]           # 1 .. 150 bases, some shorter than k
    messy = [bytes(rng.choice(np.frombuffer(b"ACGTNacgtRY", np.uint8), size=int(rng.integers(20, 200)))) for _ in range(2000)]
    contigs = [g[a:b].tobytes() for a, b in ((0, 50000), (50000, 50007), (50007, 119000), (119000, 120000))]
    cases = {
        "reads.fq": fastq(reads, rng),
        "ragged.fq": fastq(ragged, rng),
        "messy.fq": fastq(messy, rng),
        "plainhdr.fq": fastq(reads[:500], rng, qual_at=False, header=lambda i: b"@%d" % i),
        "contigs.fa": fasta(contigs, 60),
        "oneline.fa": fasta(contigs, 10 ** 9),                               # a 69 K-base line: longer than a batch and its overlap (-> host)
        "wide.fa": fasta([bytes(rng.choice(np.frombuffer(b"ACGTacgtNn-", np.uint8), size=40000))] * 3, 80),
        # what the device must hand back
        "crlf.fq": fastq(reads[:400], rng).replace(b"\n", b"\r\n"),
        "blank.fa": fasta(contigs, 70).replace(b"\n>contig2", b"\n\n>contig2"),
        "multiline.fq": b"".join(b"@m%d\n" % i + r[:75] + b"\n" + r[75:] + b"\n+\n" + b"I" * 75 + b"\n" + b"I" * 75 + b"\n" for i, r in enumerate(reads[:300])),
        "nonl.fq": fastq(reads[:700], rng)[:-1],
        "nonl.fa": fasta(contigs, 60)[:-1],
        "mixed_tail.fq": fastq(reads[:900], rng) + b"@x\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIII\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n",   # quality wrapped: by length only
    }
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in cases.items():
            path = os.path.join(tmp, name)
            with open(path, "wb") as f:
                f.write(data)
            host = kat_amd.parse_file(path)                               # the streaming state machine's base stream
            for k, canonical in ((27, True), (31, False), (15, True)):
                want = ko.Table(k, canonical).count_bases(host)
                eng.profile_reset()
                got = eng.table(k, canonical, size_hint=1 << 18)
                got.count_files([path])
                gk, gc = got.dump_sorted()
                wk, wc = want.dump_sorted()
                assert np.array_equal(gk, wk) and np.array_equal(gc, wc), (name, k, canonical, gk.size, wk.size)
                stripped = name.endswith(".fq") and os.environ.get("KATGPU_TEST_STRIP_SEGMENT")     # FASTQ through the host strip: no scan kernel at all
                if name not in ("oneline.fa", "multiline.fq", "nonl.fq", "nonl.fa") and not stripped:    # (no line end / no four-line record start inside a batch and its overlap, no final newline in a one-batch file: the host's before any kernel runs)
                    assert eng.profile()["scan"]["launches"] > 0, (name, "the device scan did not run")
                if stripped:
                    assert eng.profile()["scan"]["launches"] == 0, (name, "a scan kernel ran on a FASTQ file the readers strip")
                got.free()
                n += 1
        # k > 32 behind the same scan: the stream goes to the wide counter (batch cuts carry k - 1 = 44 bases for FASTA)
        for name in ("reads.fq", "contigs.fa"):
            path = os.path.join(tmp, name)
            want = ko.WideTable(45, True).count_bases(kat_amd.parse_file(path))
            eng.profile_reset()
            got = eng.table(45, True, size_hint=1 << 18)
            got.count_files([path])
            for a, b in zip(got.dump_sorted_wide(), want.dump_sorted()):
                assert np.array_equal(a, b), (name, "k = 45")
            assert eng.profile()["scan"]["launches"] > 0 or (name.endswith(".fq") and os.environ.get("KATGPU_TEST_STRIP_SEGMENT")), (name, "k = 45: the device scan did not run")
            got.free()
            n += 1
        # a group: scanned files and streamed ones (gzip) into one table
        import gzip
        gz = os.path.join(tmp, "more.fq.gz")
        with gzip.open(gz, "wb") as f:
            f.write(fastq(reads[3000:3600], rng))
        paths = [os.path.join(tmp, "reads.fq"), gz, os.path.join(tmp, "contigs.fa")]
        want = ko.Table(27, True)
        for p in paths:
            want.count_bases(kat_amd.parse_file(p))
        got = eng.table(27, True, size_hint=1 << 18)
        got.count_files(paths)
        for a, b in zip(got.dump_sorted(), want.dump_sorted()):
            assert np.array_equal(a, b), "group"
        n += 1
    print("scan cases ok:", n, {k: v["launches"] for k, v in eng.profile().items() if v["launches"]})


if __name__ == "__main__":
    main()
