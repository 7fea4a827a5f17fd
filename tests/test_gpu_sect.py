"""`kat sect` on the device: katgpu_table_profile_host / _device (k_profile) against the oracle's per-position lookups, and
the C++ host mirror's `katgpu sect` files byte for byte against oracle/koracle_sect.c (tests/test_sect.sh commands plus
option sweeps)."""
import os
import subprocess

import numpy as np
import pytest

from tests.test_oracle_sect import make_cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
SUFFIXES = ("-counts.cvg", "-counts.gc", "-non_repetitive.fa", "-repetitive.fa", "-stats.tsv", "-contamination.mx")


def run(args, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=300, env=e)


def files(prefix):
    return {s: open(prefix + s, "rb").read() for s in SUFFIXES if os.path.exists(prefix + s)}


def random_seq(rng, n, junk=0.01):
    s = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), n)
    bad = rng.random(n) < junk
    s[bad] = rng.choice(np.frombuffer(b"NnRY-\n\x00", np.uint8), int(bad.sum()))
    return s


@pytest.mark.parametrize("k,canonical", [(5, True), (17, False), (27, True), (32, True), (32, False)])
def test_profile_vs_oracle(ko, engine, k, canonical):
    rng = np.random.default_rng(k)
    genome = random_seq(rng, 200_000)
    t = engine.table(k, canonical)
    t.count_bases(genome)
    o = ko.Table(k, canonical).count_bases(genome)
    # a probe that shares most of its windows with the table, a shuffled one that shares few, and ragged lengths
    for probe in (genome[1000:150_000], random_seq(rng, 50_000, 0.0), genome[:k], genome[:k - 1], genome[:0], genome[5:5 + 4064 + k],
                  np.frombuffer(b"N" * 100, np.uint8)):
        for canon in (canonical, True, False):
            got = t.profile(probe, canon)
            want, _ = ko.profile(o, probe.tobytes(), canon)
            assert got.dtype == np.uint64 and got.shape == want.shape
            assert np.array_equal(got, want)
    t.free()


def test_profile_device_unaligned_and_batches(ko, engine):
    """Device-resident form at odd alignments, and the host form across its 32 M-start batch boundary."""
    k = 21
    rng = np.random.default_rng(3)
    genome = random_seq(rng, 100_000, 0.002)
    t = engine.table(k, True)
    t.count_bases(genome)
    o = ko.Table(k, True).count_bases(genome)
    probe = genome[777:90_000]
    want, _ = ko.profile(o, probe.tobytes(), True)
    for shift in (0, 1, 7, 16):
        db = engine.alloc(probe.size + 64)
        dc = engine.alloc(want.size * 8 + 64)
        db.upload(probe, offset=shift)
        t.profile_device(db.ptr + shift, probe.size, dc.ptr + (8 if shift & 1 else 0))
        engine.sync()
        got = dc.download(np.uint64, want.size, offset=8 if shift & 1 else 0)
        assert np.array_equal(got, want)
        db.free(); dc.free()
    big = np.tile(genome, 340)[: (33 << 20) + 12345]                   # > one 32 M batch; periodic, so the oracle stays cheap
    got = t.profile(big, True)
    period, _ = ko.profile(o, np.concatenate([genome, genome[: k - 1]]).tobytes(), True)
    idx = np.arange(got.size) % genome.size
    assert np.array_equal(got, period[idx])
    t.free()


def test_counts_above_32_bits(ko, engine):
    """A k-mer whose count went through the 64-bit side table is profiled with its full count."""
    k = 9
    t = engine.table(k, False)
    key = ko.encode("ACGTTGCAA")
    t.merge_host(np.array([key], np.uint64), np.array([(1 << 33) + 5], np.uint64))
    got = t.profile(b"TTACGTTGCAATT", False)
    assert got.tolist() == [0, 0, (1 << 33) + 5, 0, 0]
    ones = engine.table(4, False)
    ones.count_bases(np.frombuffer(b"TTTTTTT", np.uint8))               # the all-ones key lives in a scalar counter
    assert ones.profile(b"ATTTTTA", False).tolist() == [0, 4, 4, 0]
    t.free(); ones.free()


def test_sect_cli_reference_commands(ko, refdata, tmp_path):
    jf = os.path.join(refdata, "ecoli.header.jf27")
    t = ko.Table.from_jf(jf)
    for tag, fa in (("sect_length", "sect_length_test.fa"), ("sect_test", "sect_test.fa")):      # tests/test_sect.sh
        p = os.path.join(refdata, fa)
        r = run(["sect", "-o", "temp/" + tag, p, jf], tmp_path)
        assert r.returncode == 0, r.stderr
        assert "Running KAT in SECT mode" in r.stdout and "KAT SECT completed." in r.stdout
        ko.sect(t, p, str(tmp_path / ("want_" + tag)))
        got, want = files(str(tmp_path / "temp" / tag)), files(str(tmp_path / ("want_" + tag)))
        assert set(got) == {"-counts.cvg", "-stats.tsv"}        # Sect::main never calls save(): no contamination matrix
        assert got == want


@pytest.mark.parametrize("k,canonical", [(7, True), (21, False)])
def test_sect_cli_options(ko, tmp_path, k, canonical):
    paths, fa = make_cases(tmp_path)
    t = ko.Table(k, canonical).count_files([fa])
    base = ["sect", "-m", str(k), "-H", "10000"] + ([] if canonical else ["-N"])
    for i, p in enumerate(paths):
        r = run(base + ["-o", "o%d" % i, "-g", "-E", "-F", "-M", "2", "-G", "5", "-t", "3", "-x", "50", "-y", "7", p, fa], tmp_path,
                env={"KATGPU_SECT_SAVE": "1"})
        assert r.returncode == 0, r.stderr
        ko.sect(t, p, str(tmp_path / ("w%d" % i)), output_gc_stats=True, extract_nr=True, extract_r=True, min_repeat=2, max_repeat=5,
                gc_bins=50, cvg_bins=7, save=True)
        got, want = files(str(tmp_path / ("o%d" % i))), files(str(tmp_path / ("w%d" % i)))
        assert set(got) == set(SUFFIXES)
        for s in SUFFIXES:
            if s == "-contamination.mx":                         # the title carries the path as given on the command line
                assert got[s] == want[s]
            assert got[s] == want[s], (p, s)
    r = run(base + ["-o", "n", "-n", "-l", paths[0], fa], tmp_path)
    assert r.returncode == 0, r.stderr
    ko.sect(t, paths[0], str(tmp_path / "wn"), no_count_stats=True, cvg_logscale=True)
    assert files(str(tmp_path / "n")) == files(str(tmp_path / "wn")) and set(files(str(tmp_path / "n"))) == {"-stats.tsv"}


def test_sect_cli_errors(refdata, tmp_path):
    jf = os.path.join(refdata, "ecoli.header.jf27")
    r = run(["sect", "-o", "x", "missing.fa", jf], tmp_path)
    assert r.returncode == 4 and "Could not find sequence file at: missing.fa" in r.stderr
    r = run(["sect", "-o", "x", os.path.join(refdata, "sect_test.fa"), "missing.jf27"], tmp_path)
    assert r.returncode == 4
    bad = tmp_path / "bad.fa"
    bad.write_text("no marker\n")
    r = run(["sect", "-o", "x", str(bad), jf], tmp_path)
    assert r.returncode == 5 and "Unexpected end of input." in r.stderr
    fna = tmp_path / "contigs.fna"                                       # SeqAn goes by the file name: seqan::UnknownExtensionError
    fna.write_text(">x\nACGTACGTACGTACGTACGTACGTACGTACGT\n")
    r = run(["sect", "-o", "x", str(fna), jf], tmp_path)
    assert r.returncode == 5 and "Unknown file extension of " + str(fna) in r.stderr
    assert run(["sect"], tmp_path).returncode == 1


def test_cold_cli(ko, refdata, tmp_path):
    """`katgpu cold <assembly> <reads>+`: -stats.tsv byte for byte; counted hashes are non-canonical (Cold never sets the flag)."""
    paths, fa = make_cases(tmp_path)
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    for k, asm_path, reads in ((7, fa, [r1, paths[2]]), (27, paths[3], [r1, r2]), (21, os.path.join(refdata, "sect_length_test.fa"), [fa])):
        r = run(["cold", "-m", str(k), "-H", "100000", "-t", "2", "-o", "c%d" % k, asm_path] + reads, tmp_path)
        assert r.returncode == 0, r.stderr
        assert "Running KAT in Cold mode" in r.stdout and "KAT CoLD completed." in r.stdout
        ko.cold(ko.Table(k, False).count_files(reads), ko.Table(k, False).count_files([asm_path]), asm_path, str(tmp_path / ("w%d" % k)))
        assert (tmp_path / ("c%d-stats.tsv" % k)).read_bytes() == (tmp_path / ("w%d-stats.tsv" % k)).read_bytes()
    # a loaded .jf brings its own canonical flag (reads side), and -d dumps both hashes
    jf = os.path.join(refdata, "ecoli.header.jf27")
    asm_path = os.path.join(refdata, "sect_length_test.fa")
    r = run(["cold", "-o", "cj", "-d", asm_path, jf], tmp_path)
    assert r.returncode == 0, r.stderr
    ko.cold(ko.Table.from_jf(jf), ko.Table(27, False).count_files([asm_path]), asm_path, str(tmp_path / "wj"))
    assert (tmp_path / "cj-stats.tsv").read_bytes() == (tmp_path / "wj-stats.tsv").read_bytes()
    assert os.path.islink(tmp_path / "cj-reads_hash.jf27") and os.path.getsize(tmp_path / "cj-asm_hash.jf27") > 0
