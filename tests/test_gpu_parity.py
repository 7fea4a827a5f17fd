"""HIP path vs CPU oracle, bit-exact, through the C ABI (libkatgpu.so).  Sizes the oracle finishes in seconds."""
import os

import numpy as np
import pytest

import kat_amd
from kat_amd import synth

pytestmark = pytest.mark.gpu


def assert_same_table(gt, ot):
    gk, gc = gt.dump_sorted()
    ok_, oc = ot.dump_sorted()
    assert gk.size == ok_.size, "distinct differs: gpu %d oracle %d" % (gk.size, ok_.size)
    assert np.array_equal(gk, ok_)
    assert np.array_equal(gc, oc)
    st = gt.stats()
    assert st["distinct"] == ot.distinct and st["total"] == ot.total


def assert_same_reducers(gt, ot):
    for low, high, inc in ((1, 10000, 1), (5, 60, 1), (2, 100, 7), (1, 3, 1)):
        assert np.array_equal(gt.hist(low, high, inc), ot.hist(low, high, inc)), (low, high, inc)
    for scale, bins in ((1.0, 1000), (0.37, 50), (3.0, 20)):
        assert np.array_equal(gt.gcp(scale, bins), ot.gcp(scale, bins)), (scale, bins)


@pytest.mark.parametrize("k,canonical", [(27, True), (31, True), (17, True), (13, False), (32, True), (32, False), (1, True), (5, False)])
def test_count_synthetic_reads(engine, ko, k, canonical):
    g = synth.genome(30000, seed=11)
    stream = synth.reads(g, 0, 3000, seed=3)
    gt = engine.table(k, canonical).count_bases(stream)
    ot = ko.Table(k, canonical).count_bases(stream)
    assert_same_table(gt, ot)
    assert_same_reducers(gt, ot)


def test_reference_fastq_files(engine, ko, refdata):
    paths = [os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")]
    for k in (27, 17):
        gt = engine.count(paths, k)
        ot = ko.Table(k, True).count_files(paths)
        assert_same_table(gt, ot)
        assert_same_reducers(gt, ot)
    # SURVEY.md 8(c) known answer (recorded by the survey stage from a hand-built reference binary): kat hist -m27
    h = engine.count(paths, 27).hist()
    assert [int(h[i]) for i in range(6)] == [111200, 11696, 2063, 737, 361, 128]


def test_reference_fasta_files(engine, ko, refdata):
    for name, expect in (("sect_length_test.fa", {1094: 18, 1095: 16}), ("sect_test.fa", {1: 26})):
        p = [os.path.join(refdata, name)]
        gt = engine.count(p, 27)
        assert_same_table(gt, ko.Table(27, True).count_files(p))
        h = gt.hist()
        assert {i + 1: int(v) for i, v in enumerate(h) if v} == expect


def test_comp_reference_reads(engine, ko, refdata):
    p1 = [os.path.join(refdata, "ecoli_r1.1K.fastq")]
    p2 = [os.path.join(refdata, "ecoli_r2.1K.fastq")]
    for k, c1, c2 in ((13, True, True), (21, False, False), (21, True, False), (21, False, True)):
        g1, g2 = engine.count(p1, k, c1), engine.count(p2, k, c2)
        o1, o2 = ko.Table(k, c1).count_files(p1), ko.Table(k, c2).count_files(p2)
        for args in ((1.0, 1.0, 1001, 1001), (0.5, 2.0, 30, 50), (1.0, 1.0, 5, 3)):
            mx, cc, sp = kat_amd.comp(g1, g2, *args)
            omx, occ, osp = ko.comp(o1, o2, *args)
            assert np.array_equal(cc, occ), (k, c1, c2, args, cc, occ)
            assert np.array_equal(sp, osp)
            assert np.array_equal(mx, omx)
    # comp -m13 stats block
    mx, cc, sp = kat_amd.comp(engine.count(p1, 13), engine.count(p2, 13))
    assert list(map(int, cc)) == [87929, 88000, 0, 80366, 80554, 0, 66743, 67516, 64113, 64301, 21186, 20484, 16253]
    assert int(mx.max()) == 62111


def test_edge_inputs(engine, ko):
    k = 27
    cases = [
        b"",                                   # empty
        b"ACGT",                               # shorter than k
        b"A" * 26,                             # one short of a window
        b"A" * 27,                             # exactly one window
        b"N" * 100,
        b"acgtACGTnACGT" * 40,                 # lower case + breaks
        (b"ACGTTGCAAGGCTTAACCGGTTAGCAT" * 3 + b"R") * 5 + b"-" + b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT",
        b"G" * 5000,                           # one k-mer, count 4974 (GC == k: dropped by gcp, quirk B1)
        bytes(np.random.default_rng(5).choice(np.frombuffer(b"ACGTN\r acgt", dtype=np.uint8), size=70001)),
    ]
    for s in cases:
        gt = engine.table(k, True).count_bases(s)
        ot = ko.Table(k, True).count_bases(s)
        assert_same_table(gt, ot)
        assert_same_reducers(gt, ot)


def test_ragged_sizes_and_alignment(engine, ko):
    """Stream lengths around the 4064-start chunk and 16-byte lane granularity, device buffers at odd offsets."""
    g = synth.genome(40000, seed=2)
    k = 31
    for n in (31, 32, 47, 4063, 4064, 4065, 4064 + 30, 4064 + 31, 8128, 8129, 12345, 40000):
        s = g[:n]
        assert_same_table(engine.table(k, True).count_bases(s), ko.Table(k, True).count_bases(s))
    buf = engine.alloc(g.size + 64)
    for off in (0, 1, 7, 16, 33):
        buf.upload(g, offset=off)
        gt = engine.table(k, True)
        gt.count_bases_device(buf.ptr + off, g.size)
        assert_same_table(gt, ko.Table(k, True).count_bases(g))


def test_regrow_and_accumulate(engine, ko):
    """Tiny size hint -> several regrows (hash_counter::double_size); several count calls accumulate into one table."""
    g = synth.genome(200000, seed=9)
    a = synth.reads(g, 0, 6000, seed=1)
    b = synth.reads(g, 6000, 6000, seed=1)
    gt = engine.table(27, True, size_hint=1024)
    gt.count_bases(a)
    gt.count_bases(b)
    ot = ko.Table(27, True).count_bases(a).count_bases(b)
    assert_same_table(gt, ot)
    assert gt.stats()["capacity"] > 1024
    assert engine.profile()["regrow"]["launches"] > 0
    # -g / disable_hash_grow: a full table is an error, as in hash_counter.hpp:198-199
    with pytest.raises(kat_amd.KatGpuError) as ei:
        engine.table(27, True, size_hint=1024, disable_grow=True).count_bases(a)
    assert ei.value.code == 7 and "Hash full" in ei.value.message


def test_host_batching_seams(engine, ko):
    """A host stream larger than one 64 MiB staging buffer: k-mers across the cut are counted exactly once."""
    g = synth.genome(1 << 20, seed=4)
    s = np.tile(np.concatenate([g, np.frombuffer(b"N", np.uint8)]), 70)[: (70 << 20)]   # > 64 MiB, highly repetitive
    gt = engine.table(27, True, size_hint=1 << 22).count_bases(s)
    ot = ko.Table(27, True).count_bases(s, threads=8)
    assert_same_table(gt, ot)


def test_table_get(engine, ko, refdata):
    """JellyfishHelper::getCount semantics, incl. the known answers of the reference's tests/check_jellyfish.cc:62-91."""
    jf = ko.Table.from_jf(os.path.join(refdata, "ecoli.header.jf27"))
    keys, counts = jf.dump_sorted()
    gt = engine.table(27, False)
    gt.merge_host(keys, counts)
    q = [ko.encode(s) for s in ("AGCTTTTCATTCTGACTGCAACGGGCA", "GCATAGCGCACAGACAGATAAAAATTA",
                                "AATGAAAAAGGCGAACTGGTGGTGCTT", "CTCACCAATGTACATGGCCTTAATCTG")]
    assert list(map(int, gt.get(q, canonicalise=False))) == [3, 1, 1, 1]
    assert list(map(int, gt.get(q, canonicalise=True))) == [3, 1, 0, 0]
    assert gt.stats()["distinct"] == 1889


def test_exact_64bit_counts(engine, ko):
    """Counts beyond 32 bits stay exact (Jellyfish chains overflow into 'large' entries, large_hash_array.hpp:668-700)."""
    k = 21
    keys = np.array([5, 77, 123456789, 5], dtype=np.uint64)
    counts = np.array([0xFFFFFFFF, 3, (7 << 32) + 9, 2], dtype=np.uint64)
    gt = engine.table(k, False)
    gt.merge_host(keys, counts)
    gt.count_bases(b"A" * 18 + b"CC" + b"N")          # AAAAAAAAAAAAAAAAAACCC? no: 20 bases -> no window; stays untouched
    gt.merge_host(np.array([77], np.uint64), np.array([0xFFFFFFFE], np.uint64))
    ot = ko.Table(k, False)
    for kk, cc in ((5, 0xFFFFFFFF), (77, 3), (123456789, (7 << 32) + 9), (5, 2), (77, 0xFFFFFFFE)):
        ot.add(kk, cc)
    assert_same_table(gt, ot)
    assert_same_reducers(gt, ot)
    o2 = ko.Table(k, False)
    o2.add(5, 1 << 33)
    g2 = engine.table(k, False)
    g2.merge_host(np.array([5], np.uint64), np.array([1 << 33], np.uint64))
    mx, cc, sp = kat_amd.comp(gt, g2)
    omx, occ, osp = ko.comp(ot, o2)
    assert np.array_equal(cc, occ) and np.array_equal(mx, omx) and np.array_equal(sp, osp)


def test_partition_merge_roundtrip(engine, ko):
    """Owner partition -> merge rebuilds the same table; strands of a k-mer share an owner."""
    g = synth.genome(50000, seed=21)
    s = synth.reads(g, 0, 4000, seed=5)
    k = 25
    src = engine.table(k, False).count_bases(s)
    n_parts = 8
    sizes = src.partition_sizes(n_parts)
    total = int(sizes.sum())
    assert total == src.stats()["distinct"]
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    dk, dc = engine.alloc(total * 8), engine.alloc(total * 8)
    src.partition(n_parts, offsets, dk.ptr, dc.ptr)
    keys, counts = dk.download(np.uint64), dc.download(np.uint64)
    owner = {}
    for p in range(n_parts):
        for key in keys[int(offsets[p]): int(offsets[p] + sizes[p])]:
            can = ko.canonical(int(key), k)
            assert owner.setdefault(can, p) == p
    dst = engine.table(k, False)
    for p in range(n_parts):
        o, n = int(offsets[p]), int(sizes[p])
        dst.merge_device(dk.ptr + 8 * o, dc.ptr + 8 * o, n)
    assert_same_table(dst, ko.Table(k, False).count_bases(s))


def test_device_generator_matches_numpy(engine):
    G, n_reads = 100000, 5000
    gd = engine.synth_genome(G, seed=42)
    assert np.array_equal(gd.download(), synth.genome(G, seed=42))
    rd = engine.synth_reads(gd, G, first_read=1234, n_reads=n_reads, seed=9)
    assert np.array_equal(rd.download(), synth.reads(synth.genome(G, seed=42), 1234, n_reads, seed=9))
    ad = engine.synth_genome(G, seed=42, contig_len=7000)
    assert np.array_equal(ad.download(), synth.assembly_stream(G, 42, 7000))


def test_errors(engine, tmp_path):
    with pytest.raises(kat_amd.KatGpuError) as ei:
        engine.table(64, True)                           # wide tables stop at KATGPU_MAX_K = 63
    assert ei.value.code == 6
    with pytest.raises(kat_amd.KatGpuError) as ei:
        engine.count([str(tmp_path / "missing.fa")], 27)
    assert ei.value.code == 2 and "Could not find input file at" in ei.value.message
    bad = tmp_path / "bad.txt"
    bad.write_text("hello\nworld\n")
    with pytest.raises(kat_amd.KatGpuError) as ei:
        engine.count([str(bad)], 27)
    assert ei.value.code == 3 and "Unsupported format" in ei.value.message
    fq = tmp_path / "bad.fq"
    fq.write_text("@r1\nACGTACGT\n+\nIIII\n@r2\nACGT\n+\nIIII\n")
    with pytest.raises(kat_amd.KatGpuError) as ei:
        engine.count([str(fq)], 3)
    assert ei.value.code == 4 and "Invalid fastq sequence" in ei.value.message
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.comp(engine.table(21, True), engine.table(27, True))
    assert ei.value.code == 9


def test_unchecked_add_sweep_guard(ko, tmp_path):
    """k_count adds with no-return atomics; a host-side guard sweeps large counters into the side table before a 32-bit
    wrap becomes possible.  The test hooks shrink the threshold (64) and the launch size so a small input runs through
    hundreds of sweeps; the table must still equal the oracle's."""
    import subprocess
    import sys
    g = synth.genome(3000, seed=13)
    s = np.concatenate([np.frombuffer(b"A" * 3000 + b"N" + b"ACGT" * 500 + b"N", np.uint8), synth.reads(g, 0, 600, seed=2)])
    np.save(tmp_path / "s.npy", s)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import kat_amd\n"
        "s = np.load(%r)\n"
        "e = kat_amd.Engine(0)\n"
        "b = e.alloc(s.size); b.upload(s)\n"
        "t = e.table(21, True, size_hint=1 << 14).count_bases(b)\n"
        "k, c = t.dump_sorted(); np.savez(%r, k=k, c=c, h=t.hist(1, 5000, 1), regrow=e.profile()['regrow']['launches'])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "s.npy"), str(tmp_path / "out.npz"))
    env = dict(os.environ, KATGPU_TEST_SWEEP_THR="64", KATGPU_TEST_MAX_STARTS="64", KATGPU_NO_PACKED="1")   # (packed tables add checked: nothing to sweep)
    subprocess.run([sys.executable, "-c", code], env=env, check=True, timeout=600)
    got = np.load(tmp_path / "out.npz")
    ot = ko.Table(21, True).count_bases(s)
    ok_, oc = ot.dump_sorted()
    assert np.array_equal(got["k"], ok_) and np.array_equal(got["c"], oc)
    assert int(oc.max()) >= 2980 and np.array_equal(got["h"], ot.hist(1, 5000, 1))
    assert int(got["regrow"]) > 100                     # sweeps are accounted under the regrow class


def test_three_input_comp(engine, ko, refdata):
    p = [os.path.join(refdata, f) for f in ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq", "sect_length_test.fa")]
    for k, flags in ((13, (True, True, True)), (17, (False, True, False))):
        gt = [engine.count([p[i]], k, flags[i]) for i in range(3)]
        ot = [ko.Table(k, flags[i]).count_files([p[i]]) for i in range(3)]
        for args in ((1.0, 1.0, 1001, 1001), (0.3, 4.0, 20, 9)):
            got = kat_amd.comp3(gt[0], gt[1], gt[2], *args)
            want = ko.comp3(ot[0], ot[1], ot[2], *args)
            for g, w, name in zip(got, want, ("main", "ends", "middle", "mixed", "counters", "spectra")):
                assert np.array_equal(g, w), (k, flags, args, name)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.comp3(engine.table(21, True), engine.table(21, True), engine.table(27, True))
    assert ei.value.code == 9


def test_jf_load_query_dump(engine, ko, refdata, tmp_path):
    """The reference's own .jf tests through the device table: load tests/data/ecoli.header.jf27, answer the queries of
    tests/check_jellyfish.cc:62-91, count its records (:93-116), dump and reload (:158-180)."""
    p = os.path.join(refdata, "ecoli.header.jf27")
    t = engine.load_jf(p)
    assert (t.k, t.canonical) == (27, False) and t.stats()["distinct"] == 1889
    q = [ko.encode(s) for s in ("AGCTTTTCATTCTGACTGCAACGGGCA", "GCATAGCGCACAGACAGATAAAAATTA",
                                "AATGAAAAAGGCGAACTGGTGGTGCTT", "CTCACCAATGTACATGGCCTTAATCTG")]
    assert list(map(int, t.get(q, False))) == [3, 1, 1, 1] and list(map(int, t.get(q, True))) == [3, 1, 0, 0]
    ot = ko.Table.from_jf(p)
    assert_same_table(t, ot)
    assert_same_reducers(t, ot)
    out = str(tmp_path / "dump.jf27")
    t.dump_jf(out)
    back = engine.load_jf(out)
    assert_same_table(back, ot)
    assert list(map(int, back.get(q, False))) == [3, 1, 1, 1]
    # a counted table survives dump -> load (counts below the 4-byte saturation)
    g = synth.genome(20000, seed=3)
    c = engine.table(21, True).count_bases(synth.reads(g, 0, 2000, seed=1))
    c.dump_jf(str(tmp_path / "c.jf21"))
    assert_same_table(engine.load_jf(str(tmp_path / "c.jf21")), ko.Table.from_jf(str(tmp_path / "c.jf21")))
    assert_same_table(c, ko.Table.from_jf(str(tmp_path / "c.jf21")))


def test_comp_join_form(engine, ko):
    """Tables that share a region grid (created with like=) are compared by the region-against-region LDS join; tables
    with their own grids, or mixed canonical flags, by HBM probes.  Both must equal the oracle."""
    g = synth.genome(150000, seed=17)
    a = synth.reads(g, 0, 16000, seed=1)
    b = np.concatenate([synth.stream_of_contigs(g[:90000], 30000), synth.reads(g, 40000, 3000, seed=9, err_ppm=20000)])
    for k, c1, c2 in ((27, True, True), (21, False, False), (21, True, False), (21, False, True), (32, False, False)):
        o1, o2 = ko.Table(k, c1).count_bases(a), ko.Table(k, c2).count_bases(b)
        want = ko.comp(o1, o2, 1.0, 1.0, 201, 101)
        for hint1, hint2 in ((1 << 21, 1 << 19), (1 << 21, 1 << 22), (1 << 15, 1 << 14)):
            t1 = engine.table(k, c1, size_hint=hint1).count_bases(a)
            t2 = engine.table(k, c2, size_hint=hint2, like=t1).count_bases(b)
            got = kat_amd.comp(t1, t2, 1.0, 1.0, 201, 101)
            for gg, ww, name in zip(got, want, ("main", "counters", "spectra")):
                assert np.array_equal(gg, ww), (k, c1, c2, hint1, hint2, name)
    # exactness past 32 bits through the join form
    t1 = engine.table(21, True, size_hint=1 << 16)
    t2 = engine.table(21, True, size_hint=1 << 15, like=t1)
    o1, o2 = ko.Table(21, True), ko.Table(21, True)
    for key, c in ((5, (3 << 32) + 7), (77, 12), (1234567, 1)):
        key = ko.canonical(key, 21)
        t1.merge_host([key], [c]); o1.add(key, c)
        t2.merge_host([key], [c + (1 << 33)]); o2.add(key, c + (1 << 33))
    got, want = kat_amd.comp(t1, t2), ko.comp(o1, o2)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))


@pytest.mark.parametrize("seg", [300, 5000])
def test_count_files_through_the_ingest_thread_team(engine, ko, refdata, tmp_path, monkeypatch, seg):
    """katgpu_count_files with the multi-threaded front end (kg_ingest.hpp: parse_file_parallel) feeding the staged host path:
    same table as the oracle, for the reference's FASTQ pair and for a FASTA with k-mers across line and piece cuts."""
    monkeypatch.setenv("KATGPU_INGEST_MIN_BYTES", "0")
    monkeypatch.setenv("KATGPU_INGEST_SEGMENT", str(seg))
    monkeypatch.setenv("KATGPU_INGEST_MARGIN", "4096")
    monkeypatch.setenv("KATGPU_INGEST_THREADS", "7")
    g = synth.genome(30000, seed=4)
    fa = tmp_path / "contigs.fa"
    with open(fa, "wb") as f:
        for i in range(0, g.size, 2500):
            f.write(b">c%d\n" % i)
            c = g[i:i + 2500].tobytes()
            for j in range(0, len(c), 61):
                f.write(c[j:j + 61] + b"\n")
    paths = [os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq"), str(fa)]
    for k, canonical in ((27, True), (17, False)):
        t = engine.count(paths, k, canonical)
        o = ko.Table(k, canonical).count_files(paths)
        gk, gc = t.dump_sorted()
        ok_, oc = o.dump_sorted()
        assert np.array_equal(gk, ok_) and np.array_equal(gc, oc)
        t.free()


@pytest.mark.parametrize("files_at_once,block", [(8, 97), (2, 5000), (1, 4 << 20)])
def test_count_files_group_read_concurrently(engine, ko, refdata, tmp_path, monkeypatch, files_at_once, block):
    """katgpu_count_files on a group of files that must stream (gzip; small; 5' trim): kg_ingest.hpp's stream_group reads them
    concurrently and interleaves their blocks -- the table is the oracle's for the same files read one by one."""
    import gzip
    monkeypatch.setenv("KATGPU_INGEST_FILES", str(files_at_once))
    monkeypatch.setenv("KATGPU_INGEST_BLOCK", str(block))
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    z1, z2 = tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz"
    for src, dst in ((r1, z1), (r2, z2)):
        with open(src, "rb") as f, gzip.open(dst, "wb") as g:
            g.write(f.read())
    g = synth.genome(20000, seed=9)
    fa = tmp_path / "asm.fa.gz"
    with gzip.open(fa, "wb") as f:
        for i in range(0, g.size, 4000):
            f.write(b">c%d\n" % i + g[i:i + 4000].tobytes() + b"\n")
    paths = [str(z1), str(z2), str(fa), r1, str(z2)]
    for k, canonical, trims in ((27, True, None), (21, False, [0, 3, 0, 5, 1])):
        t = engine.count(paths, k, canonical, trim5p=trims)
        o = ko.Table(k, canonical).count_files(paths, trims)
        gk, gc = t.dump_sorted()
        ok_, oc = o.dump_sorted()
        assert np.array_equal(gk, ok_) and np.array_equal(gc, oc)
        assert t.stats()["total"] == int(oc.sum())
        t.free()


def test_count_files_bgzf(engine, ko, refdata, tmp_path, monkeypatch):
    """A bgzip-style input goes through the inflate team (kg_ingest.hpp: parse_bgzf_parallel) on its way to the device table."""
    from tests.test_ingest_parser import bgzf_bytes
    monkeypatch.setenv("KATGPU_BGZF_MIN_BYTES", "0")
    monkeypatch.setenv("KATGPU_BGZF_WINDOW", str(1 << 17))
    rng = np.random.default_rng(3)
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    z1, z2 = tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz"
    z1.write_bytes(bgzf_bytes(open(r1, "rb").read(), rng, 100, 4000))
    z2.write_bytes(bgzf_bytes(open(r2, "rb").read(), rng, 30000, 65280))
    for k, canonical, trims in ((27, True, None), (40, False, [2, 0, 5])):
        paths = [str(z1), r2, str(z2)]
        t = engine.count(paths, k, canonical, trim5p=trims)
        o = (ko.WideTable if k > 32 else ko.Table)(k, canonical).count_files([r1, r2, r2], trims)
        for a, b in zip(t.dump_sorted(), o.dump_sorted()):
            assert np.array_equal(a, b)
        t.free()


JF_REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "jf_ref")


@pytest.mark.skipif(not os.access(JF_REF, os.X_OK), reason="oracle/_ref not built (no /root/reference at build time)")
def test_hip_table_against_the_reference_parser_directly(engine, ko, refdata, tmp_path):
    """No oracle in between: the HIP table's (k-mer, count) dump equals what the REAL Jellyfish 2.2.0 parser + mer_iterator of the
    reference deliver (oracle/_ref/jf_ref, compiled from /root/reference's sources) for the same files."""
    import subprocess
    from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq
    rng = np.random.default_rng(33)
    fa, fq = tmp_path / "m.fa", tmp_path / "mm.fq"
    write_messy_fasta(str(fa), rng)
    write_messy_fastq(str(fq), rng, multiline=True)
    groups = [[os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")], [os.path.join(refdata, "sect_length_test.fa")],
              [str(fa), str(fq)]]
    for paths in groups:
        for k, canonical in ((27, True), (31, False), (11, True)):
            out = subprocess.run([JF_REF, "kmers", str(k), str(int(canonical))] + paths, capture_output=True, timeout=300)
            assert out.returncode == 0
            t = engine.count(paths, k, canonical)
            keys, counts = t.dump_sorted()
            got = "".join("%s %d\n" % (ko.decode(int(a), k), int(b)) for a, b in zip(keys, counts)).encode()
            assert got == out.stdout, (paths, k, canonical)
            t.free()
