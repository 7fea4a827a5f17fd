"""k > 32 on the device ("wide" tables: a k-mer in two 63-bit words, kat_amd/csrc/kg_device.hpp + kg_wide.hpp) against the wide
oracle (oracle/koracle_wide.c, itself checked against the reference's parser + multi-word mer_dna), bit-exact, through the C ABI."""
import os

import numpy as np
import pytest

import kat_amd
from kat_amd import synth
from tests.test_oracle_vs_naive import write_messy_fasta, write_messy_fastq

pytestmark = pytest.mark.gpu


def assert_same_wide(gt, ot):
    g = gt.dump_sorted()
    o = ot.dump_sorted()
    assert g[0].size == o[0].size, "distinct differs: gpu %d oracle %d" % (g[0].size, o[0].size)
    for a, b in zip(g, o):
        assert np.array_equal(a, b)
    st = gt.stats()
    assert st["distinct"] == ot.distinct and st["total"] == ot.total


def assert_same_reducers(gt, ot):
    for low, high, inc in ((1, 10000, 1), (5, 60, 1), (2, 100, 7)):
        assert np.array_equal(gt.hist(low, high, inc), ot.hist(low, high, inc)), (low, high, inc)
    for scale, bins in ((1.0, 1000), (0.37, 50)):
        assert np.array_equal(gt.gcp(scale, bins), ot.gcp(scale, bins)), (scale, bins)


@pytest.mark.parametrize("k,canonical", [(33, True), (33, False), (40, True), (47, False), (48, True), (62, True), (63, True), (63, False)])
def test_count_synthetic_reads_wide(engine, ko, k, canonical):
    g = synth.genome(30000, seed=11)
    stream = synth.reads(g, 0, 3000, seed=3)
    gt = engine.table(k, canonical).count_bases(stream)
    ot = ko.WideTable(k, canonical).count_bases(stream)
    assert ot.distinct > 10000
    assert_same_wide(gt, ot)
    assert_same_reducers(gt, ot)


def test_files_and_messy_inputs_wide(engine, ko, refdata, tmp_path):
    rng = np.random.default_rng(5)
    fa, fq = tmp_path / "m.fa", tmp_path / "m.fq"
    write_messy_fasta(str(fa), rng)
    write_messy_fastq(str(fq), rng)
    ref = [os.path.join(refdata, f) for f in ("ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq", "sect_length_test.fa")]
    for paths, k, canonical in ((ref, 41, True), (ref, 63, False), ([str(fa), str(fq)], 35, True), ([str(fa), str(fq)], 57, False)):
        gt = engine.count(paths, k, canonical)
        ot = ko.WideTable(k, canonical).count_files(paths)
        assert_same_wide(gt, ot)
        assert_same_reducers(gt, ot)


def test_edges_ragged_and_unaligned_wide(engine, ko):
    """Window seams at every chunk offset, inputs shorter than k, non-base bytes at each position of the 80-base register window,
    unaligned device buffers."""
    k = 50
    rng = np.random.default_rng(1)
    base = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), 20000)
    for n in (0, 1, 49, 50, 51, 64, 79, 80, 81, 4031, 4032, 4033, 4095, 4096, 4097, 8064, 8065, 12345):
        s = base[:n]
        assert_same_wide(engine.table(k, True).count_bases(s), ko.WideTable(k, True).count_bases(s))
    for pos in list(range(0, 100)) + [4000, 4031, 4032, 4033, 4080, 4095, 4096]:
        s = base[:9000].copy()
        s[pos] = ord("N")
        s[pos + 60] = ord("-")
        assert_same_wide(engine.table(k, False).count_bases(s), ko.WideTable(k, False).count_bases(s))
    buf = engine.alloc(base.size)
    buf.upload(base)
    for off in (1, 3, 8, 15):
        gt = engine.table(k, True)
        gt.count_bases_device(buf.ptr + off, base.size - off)
        assert_same_wide(gt, ko.WideTable(k, True).count_bases(base[off:]))


def test_regrow_accumulate_and_hash_full_wide(engine, ko):
    g = synth.genome(200000, seed=3)
    s1, s2 = synth.reads(g, 0, 6000, seed=1), synth.reads(g, 6000, 6000, seed=1)
    gt = engine.table(45, True, size_hint=1024)               # grows many times
    gt.count_bases(s1).count_bases(s2)
    ot = ko.WideTable(45, True).count_bases(s1).count_bases(s2)
    assert_same_wide(gt, ot)
    assert gt.stats()["capacity"] > 1_000_000
    small = engine.table(45, True, size_hint=1024, disable_grow=True)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        small.count_bases(s1)
    assert ei.value.code == 7 and "Hash full" in ei.value.message


def test_exact_64bit_counts_and_records_wide(engine, ko):
    """Counts beyond 32 bits stay exact (side table keyed by slot), records in and out through the *_wide entry points."""
    k = 40
    M = (1 << 64) - 1
    keys = [5, (1 << 79) + 77, (0xABCDEF << 50) + 123456789, 5, (1 << 80) - 1]
    counts = [0xFFFFFFFF, 3, (7 << 32) + 9, 2, 1 << 33]
    gt = engine.table(k, False)
    gt.merge_host_wide([x >> 64 for x in keys], [x & M for x in keys], counts)
    gt.merge_host_wide([keys[1] >> 64], [keys[1] & M], [0xFFFFFFFE])
    ot = ko.WideTable(k, False)
    for x, c in list(zip(keys, counts)) + [(keys[1], 0xFFFFFFFE)]:
        ot.add(x, c)
    assert_same_wide(gt, ot)
    assert_same_reducers(gt, ot)
    q = keys[:3] + [12345]
    assert list(map(int, gt.get_wide([x >> 64 for x in q], [x & M for x in q]))) == [ot.get(x) for x in q] == [0x100000001, 0x100000001, (7 << 32) + 9, 0]
    g2 = engine.table(k, False, size_hint=1024)                # regrow carries the 64-bit amounts along
    hi, lo, c = gt.export_wide()
    g2.merge_host_wide(hi, lo, c)
    s = synth.reads(synth.genome(40000, seed=2), 0, 2000, seed=9)
    g2.count_bases(s)
    ot.count_bases(s)
    assert_same_wide(g2, ot)
    with pytest.raises(kat_amd.KatGpuError):
        gt.merge_host_wide([1 << 16], [0], [1])                # wider than 2k = 80 bits


@pytest.mark.parametrize("k,c1,c2", [(33, True, True), (51, True, False), (63, False, True), (44, False, False)])
def test_comp_and_comp3_wide(engine, ko, k, c1, c2):
    g = synth.genome(40000, seed=7)
    reads = synth.reads(g, 0, 5000, seed=2)
    asm = synth.stream_of_contigs(g[:30000], 5000)
    extra = synth.reads(g, 5000, 1500, seed=4)
    g1, g2, g3 = engine.table(k, c1).count_bases(reads), engine.table(k, c2).count_bases(asm), engine.table(k, c1).count_bases(extra)
    o1, o2, o3 = ko.WideTable(k, c1).count_bases(reads), ko.WideTable(k, c2).count_bases(asm), ko.WideTable(k, c1).count_bases(extra)
    for args in ((1.0, 1.0, 1001, 1001), (0.5, 2.0, 40, 70)):
        for a, b in zip(kat_amd.comp(g1, g2, *args), ko.comp(o1, o2, *args)):
            assert np.array_equal(a, b)
    for a, b in zip(kat_amd.comp3(g1, g2, g3, 1.0, 1.0, 60, 60), ko.comp3(o1, o2, o3, 1.0, 1.0, 60, 60)):
        assert np.array_equal(a, b)


def test_narrow_only_entry_points_say_so(engine, tmp_path):
    t = engine.table(40, True).count_bases(b"ACGT" * 30)
    for call in (lambda: t.get(np.array([1], np.uint64)), lambda: t.export(),
                 lambda: t.geometry(), lambda: t.merge_host([1], [1])):
        with pytest.raises(kat_amd.KatGpuError) as ei:
            call()
        assert ei.value.code == 6, ei.value
    n = engine.table(27, True)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        n.export_wide()
    assert ei.value.code == 6
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.comp(t, n)
    assert ei.value.code == 9


def test_cli_hist_gcp_comp_wide(ko, refdata, tmp_path):
    """The C++ host drivers at k = 41: the same files as the oracle's writers produce; sect says why it cannot."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kat_amd", "bin", "katgpu")
    run = lambda args: subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    r = run(["hist", "-m", "41", "-o", "h", r1, r2])
    assert r.returncode == 0, r.stderr
    o = ko.WideTable(41, True).count_files([r1, r2])
    ko.write_hist(str(tmp_path / "h.want"), 41, [r1, r2], 1, 10000, 1, o.hist())
    assert (tmp_path / "h").read_bytes() == (tmp_path / "h.want").read_bytes()
    r = run(["gcp", "-m", "41", "-o", "g", r1, r2])
    assert r.returncode == 0, r.stderr
    ko.write_gcp(str(tmp_path / "g.want"), 41, [r1, r2], 1000, o.gcp())
    assert (tmp_path / "g.mx").read_bytes() == (tmp_path / "g.want").read_bytes()
    r = run(["comp", "-m", "41", "-o", "c", r1, r2])
    assert r.returncode == 0, r.stderr
    o1, o2 = ko.WideTable(41, True).count_files([r1]), ko.WideTable(41, True).count_files([r2])
    mx, cc, sp = ko.comp(o1, o2)
    ko.write_comp(str(tmp_path / "want"), 41, [r1], [r2], 1001, 1001, mx, cc, sp)
    assert (tmp_path / "c-main.mx").read_bytes() == (tmp_path / "want-main.mx").read_bytes()
    assert (tmp_path / "c.stats").read_bytes() == (tmp_path / "want.stats").read_bytes()


@pytest.mark.parametrize("k,canonical", [(33, True), (50, False), (63, True)])
def test_profile_wide(ko, engine, k, canonical):
    """katgpu_table_profile_* (k_profile_w) against the oracle's per-position lookups: shared, shuffled and ragged probes, both
    canonicalisation choices, the device form at odd alignments, a count above 32 bits."""
    from tests.test_gpu_sect import random_seq
    rng = np.random.default_rng(k)
    genome = random_seq(rng, 200_000)
    t = engine.table(k, canonical).count_bases(genome)
    o = ko.WideTable(k, canonical).count_bases(genome)
    for probe in (genome[1000:150_000], random_seq(rng, 50_000, 0.0), genome[:k], genome[:k - 1], genome[:0], genome[5:5 + 4032 + k],
                  genome[3:3 + 4031 + k], np.frombuffer(b"N" * 100, np.uint8)):
        for canon in (canonical, True, False):
            want, _ = ko.profile(o, probe.tobytes(), canon)
            assert np.array_equal(t.profile(probe, canon), want)
    probe = genome[777:90_000]
    want, _ = ko.profile(o, probe.tobytes(), True)
    for shift in (0, 1, 7):
        db, dc = engine.alloc(probe.size + 64), engine.alloc(want.size * 8 + 64)
        db.upload(probe, offset=shift)
        t.profile_device(db.ptr + shift, probe.size, dc.ptr + (8 if shift & 1 else 0), True)
        engine.sync()
        assert np.array_equal(dc.download(np.uint64, want.size, offset=8 if shift & 1 else 0), want)
        db.free(); dc.free()
    big = engine.table(k, False)
    i = next(j for j in range(1000, 5000) if all(c in b"ACGTacgt" for c in genome[j:j + k].tobytes()))
    word = 0
    for c in genome[i:i + k].tobytes().upper():
        word = (word << 2) | b"ACGT".index(c)
    big.merge_host_wide([word >> 64], [word & ((1 << 64) - 1)], [(1 << 33) + 5])
    got = big.profile(genome[i - 2:i + k + 2], False)
    assert got.tolist() == [0, 0, (1 << 33) + 5, 0, 0]


def test_sect_and_cold_cli_wide(ko, refdata, tmp_path):
    """`katgpu sect` / `katgpu cold` at k = 41: the files koracle_sect.c writes from the wide oracle table, byte for byte."""
    from tests.test_gpu_sect import run, files, SUFFIXES
    from tests.test_oracle_sect import make_cases
    paths, fa = make_cases(tmp_path)
    length = os.path.join(refdata, "sect_length_test.fa")
    r1 = os.path.join(refdata, "ecoli_r1.1K.fastq")
    k = 41
    t = ko.WideTable(k, True).count_files([fa, length])
    for i, p in enumerate([paths[0], paths[2], paths[3], length]):
        r = run(["sect", "-m", str(k), "-H", "100000", "-o", "o%d" % i, "-g", "-E", "-F", "-M", "2", "-G", "5", "-x", "50", "-y", "7", p, fa, length],
                tmp_path, env={"KATGPU_SECT_SAVE": "1"})
        assert r.returncode == 0, r.stderr
        ko.sect(t, p, str(tmp_path / ("w%d" % i)), output_gc_stats=True, extract_nr=True, extract_r=True, min_repeat=2, max_repeat=5,
                gc_bins=50, cvg_bins=7, save=True)
        got, want = files(str(tmp_path / ("o%d" % i))), files(str(tmp_path / ("w%d" % i)))
        assert set(got) == set(SUFFIXES) and got == want, p
    r = run(["cold", "-m", str(k), "-H", "100000", "-o", "c", length, r1, fa], tmp_path)
    assert r.returncode == 0, r.stderr
    ko.cold(ko.WideTable(k, False).count_files([r1, fa]), ko.WideTable(k, False).count_files([length]), length, str(tmp_path / "wc"))
    assert (tmp_path / "c-stats.tsv").read_bytes() == (tmp_path / "wc-stats.tsv").read_bytes()


def test_jf_dump_and_load_wide(engine, ko, refdata, tmp_path):
    """-d dumps and .jf inputs at k > 32: dump -> (the reference's reader, when oracle/_ref is there) -> load gives the table back;
    the CLI counts from the .jf what it counts from the reads."""
    import subprocess
    from tests.test_oracle_vs_reference import JF_REF, ref, parse_jfread
    r1 = os.path.join(refdata, "ecoli_r1.1K.fastq")
    for k, canonical in ((45, True), (63, False)):
        gt = engine.count([r1], k, canonical)
        p = str(tmp_path / ("d%d.jf" % k))
        gt.dump_jf(p)
        back = engine.load_jf(p)
        assert (back.k, back.canonical) == (k, canonical)
        ot = ko.WideTable(k, canonical).count_files([r1])
        assert_same_wide(back, ot)
        if os.access(JF_REF, os.X_OK):
            rc, out = ref(JF_REF, ["jfread", p])
            assert rc == 0
            hdr, recs = parse_jfread(out)
            assert hdr["key_len"] == str(2 * k) and len(recs) == ot.distinct
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kat_amd", "bin", "katgpu")
    run = lambda args: subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    r = run(["hist", "-m", "41", "-d", "-o", "a", r1])
    assert r.returncode == 0, r.stderr
    r = run(["hist", "-o", "b", "a-hash.jf41"])
    assert r.returncode == 0, r.stderr
    body = lambda f: [ln for ln in (tmp_path / f).read_text().splitlines() if not ln.startswith("#")]
    assert body("a") == body("b") and len(body("a")) == 10001


def test_wide_checksums_at_scale(engine):
    """4 M reads at k = 51 (400 M k-mer instances) counted on the device: sum of counts == number of windows, the reducers'
    marginals agree, and the table does not depend on batching or order."""
    import time
    G, n_reads, k, L = 20_000_000, 4_000_000, 51, 150
    g = engine.synth_genome(G, seed=20260927)
    reads = engine.synth_reads(g, G, 0, n_reads, seed=1)
    asm = engine.synth_genome(G, seed=20260927, contig_len=1_000_000)
    engine.sync()
    t0 = time.perf_counter()
    t1 = engine.table(k, True, size_hint=200_000_000).count_bases(reads)
    engine.sync()
    dt = time.perf_counter() - t0
    t2 = engine.table(k, True, size_hint=40_000_000).count_bases(asm)
    s1, s2 = t1.stats(), t2.stats()
    print("wide count k=51: %.2f G k-mers/s" % (s1["total"] / dt / 1e9))
    assert s1["total"] == n_reads * (L - k + 1)
    assert s2["total"] == G - (G // 1_000_000) * (k - 1)
    assert 0.999 * s2["total"] < s2["distinct"] <= s2["total"]
    h = t1.hist()
    assert int(h.sum()) == s1["distinct"]
    gm = t1.gcp()
    assert int(gm.sum()) == s1["distinct"]                                # an all-G/C 51-mer is not going to happen
    mx, cc, sp = kat_amd.comp(t1, t2)
    assert int(cc[0]) == s1["total"] and int(cc[1]) == s2["total"] and int(cc[3]) == s1["distinct"] and int(cc[4]) == s2["distinct"]
    assert int(cc[8]) + int(cc[12]) == int(cc[3]) and int(cc[9]) + int(cc[12]) == int(cc[4])
    assert int(mx.sum()) == int(cc[3]) + int(cc[9])
    assert np.array_equal(mx.sum(axis=1)[1:], sp[0][1:])
    # error k-mers: ~ (1 - 0.998^51) = 9.7 % of the instances are singletons absent from the assembly
    assert 0.07 < int(mx[1, 0]) / s1["total"] < 0.12
    # 100 K of the reads, in four unequal calls, in reverse order, into a table that starts tiny: same as one call
    rec = L + 1
    n_q = 100_000
    whole = engine.table(k, True, size_hint=20_000_000)
    whole.count_bases_device(reads.ptr, n_q * rec)
    parts = engine.table(k, True, size_hint=1 << 12)
    cuts = [0, 10_000, 35_001, 77_777, n_q]
    for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):
        parts.count_bases_device(reads.ptr + a * rec, (b - a) * rec)
    for x, y in zip(whole.dump_sorted(), parts.dump_sorted()):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("region_slots,round_items,spill_mod,extra", [
    (512, 100000, 0, {}), (1024, 3000000, 7, {}), (6144, 400000, 0, {}), (256, 150000, 5, {"KATGPU_TEST_GROW_NOMEM": "1"})])
def test_partitioned_counter_wide(region_slots, round_items, spill_mod, extra):
    """The partitioned counter for k > 32 (kg_partition_wide.hpp: 16-byte items through two exact radix levels, regions of 20-byte slots
    applied in LDS) against the wide oracle, with hooks that force it for small inputs: several rounds per call, regions that fill
    and spill, regrows in mid-call, growth that loses the arena."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, KATGPU_PART_MIN_STARTS="0", KATGPU_TEST_REGION_SLOTS=str(region_slots),
               KATGPU_TEST_ROUND_ITEMS=str(round_items), KATGPU_TEST_SPILL_MOD=str(spill_mod))
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(here, "wide_partition_cases.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "wide partition cases ok" in r.stdout
