"""The C++ host mirror (kat_amd/bin/katgpu: InputHandler / Histogram / Gcp / Comp above the C ABI) writes the same bytes
as the oracle's restatement of KAT's writers, on the reference's tests/data inputs and its own CLI test commands
(tests/test_hist.sh, test_gcp.sh, test_comp.sh) and on generated PE reads + assembly."""
import os
import subprocess

import numpy as np
import pytest

from kat_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kat_amd", "bin", "katgpu")


def run(args, cwd):
    r = subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    return r


def test_hist_cli(ko, refdata, tmp_path):
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    r = run(["hist", "-m17", "-o", "temp/hist_test", r1, r2], tmp_path)            # tests/test_hist.sh
    assert r.returncode == 0, r.stderr
    assert "Running KAT in HIST mode" in r.stdout and "KAT HIST completed." in r.stdout
    assert "Warning: Specified hash size" not in r.stdout                   # the default -H is plenty for 2 x 1000 reads
    t = ko.Table(17, True).count_files([r1, r2])
    ko.write_hist(str(tmp_path / "want"), 17, [r1, r2], 1, 10000, 1, t.hist())
    assert (tmp_path / "temp" / "hist_test").read_bytes() == (tmp_path / "want").read_bytes()
    # non-default geometry + non-canonical + default output name
    r = run(["hist", "-m", "21", "-l", "3", "--high=50", "-i", "4", "-N", "-H", "5000", r1], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Warning: Specified hash size insufficent - attempting to double hash size... success!" in r.stdout     # -H 5000 is too small
    t = ko.Table(21, False).count_files([r1])
    ko.write_hist(str(tmp_path / "want2"), 21, [r1], 3, 50, 4, t.hist(3, 50, 4))
    assert (tmp_path / "kat.hist").read_bytes() == (tmp_path / "want2").read_bytes()


def test_hist_cli_survey_md5(refdata, tmp_path):
    """The product's own `hist -m27` files for the reference's FASTA fixtures carry the body the reference binary wrote."""
    from tests.test_oracle_known_answers import HIST_BODY_MD5, hist_body_md5
    for name, want in HIST_BODY_MD5.items():
        r = run(["hist", "-m27", "-o", name + ".hist", os.path.join(refdata, name)], tmp_path)
        assert r.returncode == 0, r.stderr
        assert hist_body_md5(str(tmp_path / (name + ".hist"))) == want, name


def test_gcp_cli(ko, refdata, tmp_path):
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    r = run(["gcp", "-m17", "-o", "temp/gcp_test", r1, r2], tmp_path)              # tests/test_gcp.sh
    assert r.returncode == 0, r.stderr
    t = ko.Table(17, True).count_files([r1, r2])
    ko.write_gcp(str(tmp_path / "want.mx"), 17, [r1, r2], 1000, t.gcp())
    got = (tmp_path / "temp" / "gcp_test.mx").read_bytes()
    assert got == (tmp_path / "want.mx").read_bytes()
    assert b"# Columns:1001\n# Rows:17\n# MaxVal:21046\n" in got
    r = run(["gcp", "-m", "27", "-x", "0.25", "-y", "40", "-o", "g2", r1], tmp_path)
    assert r.returncode == 0, r.stderr
    ko.write_gcp(str(tmp_path / "want2.mx"), 27, [r1], 40, ko.Table(27, True).count_files([r1]).gcp(0.25, 40))
    assert (tmp_path / "g2.mx").read_bytes() == (tmp_path / "want2.mx").read_bytes()


def test_comp_cli(ko, refdata, tmp_path):
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    r = run(["comp", "-m13", "-v", "-n", "-h", "-o", "temp/density_test", r1, r2], tmp_path)   # tests/test_comp.sh (+ -h)
    assert r.returncode == 0, r.stderr
    t1, t2 = ko.Table(13, True).count_files([r1]), ko.Table(13, True).count_files([r2])
    mx, cc, sp = ko.comp(t1, t2)
    ko.write_comp(str(tmp_path / "want"), 13, [r1], [r2], 1001, 1001, mx, cc, sp, hists=True)
    for suffix in ("-main.mx", ".stats", ".1.hist", ".2.hist"):
        assert (tmp_path / "temp" / ("density_test" + suffix)).read_bytes() == (tmp_path / ("want" + suffix)).read_bytes(), suffix
    assert (tmp_path / "want.stats").read_text() in r.stdout                     # stats block echoed to stdout (src/comp.cc:829-833)
    # quoted glob group (tests/test_comp.sh: '${data}/ecoli_r?.1K.fastq') vs a FASTA, mixed canonical flags, scaling, small bins
    asm = os.path.join(refdata, "sect_length_test.fa")
    pattern = os.path.join(refdata, "ecoli_r?.1K.fastq")
    r = run(["comp", "-m21", "-O", "-x", "0.5", "-y", "2", "-i", "40", "-j", "30", "-o", "glob_test", pattern, asm], tmp_path)
    assert r.returncode == 0, r.stderr
    t1, t2 = ko.Table(21, True).count_files([r1, r2]), ko.Table(21, False).count_files([asm])
    mx, cc, sp = ko.comp(t1, t2, 0.5, 2.0, 40, 30)
    ko.write_comp(str(tmp_path / "w2"), 21, [r1, r2], [asm], 40, 30, mx, cc, sp)
    for suffix in ("-main.mx", ".stats"):
        assert (tmp_path / ("glob_test" + suffix)).read_bytes() == (tmp_path / ("w2" + suffix)).read_bytes(), suffix


def test_comp_three_inputs_cli(ko, refdata, tmp_path):
    """`kat comp <reads1> <reads2> <asm>`: -main/-ends/-middle/-mixed .mx + .stats with the Hash 3 lines."""
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    asm = os.path.join(refdata, "sect_length_test.fa")
    r = run(["comp", "-m15", "-P", "-o", "three", r1, r2, asm], tmp_path)
    assert r.returncode == 0, r.stderr
    t = [ko.Table(15, True).count_files([r1]), ko.Table(15, True).count_files([r2]), ko.Table(15, False).count_files([asm])]
    main, ends, middle, mixed, cc, sp = ko.comp3(t[0], t[1], t[2])
    ko.write_comp3(str(tmp_path / "want"), 15, [r1], [r2], [asm], 1001, 1001, (main, ends, middle, mixed), cc, sp)
    for suffix in ("-main.mx", "-ends.mx", "-middle.mx", "-mixed.mx", ".stats"):
        assert (tmp_path / ("three" + suffix)).read_bytes() == (tmp_path / ("want" + suffix)).read_bytes(), suffix
    assert b' - Hash 3: "' in (tmp_path / "three.stats").read_bytes()


def test_jf_inputs_and_dump_cli(ko, refdata, tmp_path):
    """A .jf input (LOAD mode, k from its header) and -d (dump the counted hash, reload it, same answer)."""
    jf = os.path.join(refdata, "ecoli.header.jf27")
    r = run(["hist", "-o", "jf.hist", jf], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Loading hashes into memory..." in r.stdout
    ko.write_hist(str(tmp_path / "want"), 27, [jf], 1, 10000, 1, ko.Table.from_jf(jf).hist())
    assert (tmp_path / "jf.hist").read_bytes() == (tmp_path / "want").read_bytes()
    r1 = os.path.join(refdata, "ecoli_r1.1K.fastq")
    r = run(["hist", "-m21", "-d", "-o", "d.hist", r1], tmp_path)
    assert r.returncode == 0, r.stderr
    dumped = tmp_path / "d.hist-hash.jf21"                                   # src/histogram.cc:105-108
    assert dumped.exists()
    r = run(["hist", "-o", "reload.hist", str(dumped)], tmp_path)
    assert r.returncode == 0, r.stderr
    body = lambda p: [l for l in p.read_text().split("\n") if not l.startswith("#")]
    assert body(tmp_path / "d.hist") == body(tmp_path / "reload.hist")
    # comp of a hash against reads, and of two hashes with different k -> KAT's validateMerLen error (exit 4)
    r = run(["comp", "-m21", "-o", "c1", str(dumped), r1], tmp_path)
    assert r.returncode == 0, r.stderr
    t1, t2 = ko.Table.from_jf(str(dumped)), ko.Table(21, True).count_files([r1])
    mx, cc, sp = ko.comp(t1, t2)
    ko.write_comp(str(tmp_path / "wc1"), 21, [str(dumped)], [r1], 1001, 1001, mx, cc, sp)
    assert (tmp_path / "c1-main.mx").read_bytes() == (tmp_path / "wc1-main.mx").read_bytes()
    assert (tmp_path / "c1.stats").read_bytes() == (tmp_path / "wc1.stats").read_bytes()
    r = run(["comp", "-o", "c2", str(dumped), jf], tmp_path)
    assert r.returncode == 4 and "different K-mer lengths" in r.stderr


def test_generated_pe_reads_vs_assembly(ko, tmp_path):
    """Parity-scale version of BASELINE.json configs[3]: PE FASTQ + assembly FASTA written to disk, both sides read the files."""
    g = synth.genome(300000, seed=20260927)
    synth.write_fasta(str(tmp_path / "asm.fa"), g, contig_len=50000)
    synth.write_fastq_pair(str(tmp_path / "lib_R1.fq"), str(tmp_path / "lib_R2.fq"), synth.reads(g, 0, 40000, seed=1))
    r = run(["comp", "-m27", "-H", "3000000", "-o", "cmp", "lib_R?.fq", "asm.fa"], tmp_path)
    assert r.returncode == 0, r.stderr
    p1, p2 = ["lib_R1.fq", "lib_R2.fq"], ["asm.fa"]
    os.chdir(tmp_path)
    t1, t2 = ko.Table(27, True).count_files(p1), ko.Table(27, True).count_files(p2)
    mx, cc, sp = ko.comp(t1, t2)
    ko.write_comp("want", 27, p1, p2, 1001, 1001, mx, cc, sp)
    assert (tmp_path / "cmp-main.mx").read_bytes() == (tmp_path / "want-main.mx").read_bytes()
    assert (tmp_path / "cmp.stats").read_bytes() == (tmp_path / "want.stats").read_bytes()


def test_cli_exit_codes(refdata, tmp_path):
    """src/kat.cc:286-302: option errors 1, KAT exceptions 4, std::exception 5."""
    assert run(["hist", "--no-such-flag", "x.fa"], tmp_path).returncode == 1
    r = run(["hist", "-m", "27", "does_not_exist.fa"], tmp_path)
    assert r.returncode == 4 and "Could not find input file at: does_not_exist.fa" in r.stderr
    r = run(["hist", "-l", "10", "-h", "5", os.path.join(refdata, "sect_test.fa")], tmp_path)
    assert r.returncode == 4 and "High count value must be >= to low count value" in r.stderr
    (tmp_path / "junk.fa").write_text("not a sequence file\n")
    r = run(["hist", "-m", "27", "junk.fa"], tmp_path)
    assert r.returncode == 5 and "Unsupported format" in r.stderr
    r = run(["comp", "-m", "27", "-g", "-H", "1000", os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "sect_test.fa")], tmp_path)
    assert r.returncode == 5 and "Hash full" in r.stderr
    assert run(["filter", "x"], tmp_path).returncode == 1                        # a mode this build does not carry


KAT_REF = os.path.join(ROOT, "oracle", "_ref", "kat_ref_parts")


@pytest.mark.skipif(not os.access(KAT_REF, os.X_OK), reason="oracle/_ref not built (no /root/reference at build time)")
def test_comp_stats_and_matrix_through_the_reference_code(engine, refdata, tmp_path):
    """`katgpu comp`'s files against the reference's OWN code (oracle/_ref/kat_ref_parts, no oracle in between): the .stats file is
    what KAT's CompCounters::printCounts prints for the device's counters and spectra, and the -main.mx file loads in KAT's
    SparseMatrix / matrix_metadata_extractor with the device matrix's shape, MaxVal and cell sum."""
    import kat_amd
    r1, r2 = os.path.join(refdata, "ecoli_r1.1K.fastq"), os.path.join(refdata, "ecoli_r2.1K.fastq")
    r = run(["comp", "-m15", "-i", "120", "-j", "90", "-o", "refcmp", r1, r2], tmp_path)
    assert r.returncode == 0, r.stderr
    t1, t2 = engine.count([r1], 15, True), engine.count([r2], 15, True)
    mx, cc, sp = kat_amd.comp(t1, t2, 1.0, 1.0, 120, 90)
    n = sp.shape[1]
    stdin = "%s\n%s\n\n%d\n%s\n%s\n" % (r1, r2, n, " ".join(map(str, cc.tolist())), "\n".join(" ".join(map(str, row.tolist())) for row in sp))
    out = subprocess.run([KAT_REF, "compstats"], input=stdin.encode(), capture_output=True, timeout=120)
    assert out.returncode == 0 and out.stdout == (tmp_path / "refcmp.stats").read_bytes()
    out = subprocess.run([KAT_REF, "mxread", str(tmp_path / "refcmp-main.mx")], capture_output=True, timeout=120)
    lines = out.stdout.decode().split("\n")
    assert lines[0].split() == ["90", "120", str(int(mx.max())), "1", "15"]
    assert lines[7].split() == ["120", "90", str(int(mx.max())), str(int(mx.sum()))]
    t1.free(); t2.free()


@pytest.mark.parametrize("gpus,env", [(1, {"KATGPU_COMM_TRANSPORT": "rccl"}), (2, {"KATGPU_COMM_TRANSPORT": "shm"}), (3, {"KATGPU_COMM_TRANSPORT": "shm"}),
                                      (2, {"KATGPU_COMM_TRANSPORT": "shm", "KATGPU_TEST_STRIP_SEGMENT": "131072", "KATGPU_TEST_STRIP_OVERLAP": "8192"})])     # the ranks cut the FASTQ files between themselves on the host-strip path
def test_gpus_switch_writes_the_single_gpu_files(tmp_path, gpus, env):
    """`katgpu <mode> --gpus N` (kat_main.cc forks N ranks; kg_comm.hip merges their tables by owner and sums the reducers' results):
    byte for byte the files of the plain run.  --gpus 1 takes the RCCL branch with one rank; 2 and 3 ranks share this box's one GPU, which
    only the /dev/shm transport carries (RCCL refuses two ranks on a device) -- the protocol above the transport is the same code.
    The FASTQ files are big enough for the device scan with the test's batch size, so the ranks cut them between themselves."""
    g = synth.genome(300000, seed=20260927)
    synth.write_fasta(str(tmp_path / "asm.fa"), g, contig_len=50000)
    synth.write_fastq_pair(str(tmp_path / "lib_R1.fq"), str(tmp_path / "lib_R2.fq"), synth.reads(g, 0, 40000, seed=1))
    base_env = dict(os.environ, KATGPU_TEST_SCAN_BATCH="1048576", KATGPU_TEST_SCAN_SEGMENT="262144", KATGPU_TEST_SCAN_OVERLAP="8192", KATGPU_TRACE="1")

    def go(extra, name, e):
        r = subprocess.run([EXE] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=300, env=e)
        if r.returncode and "did not return within" in r.stderr and "KATGPU_COMM_INIT_TIMEOUT_S" in r.stderr:
            pytest.skip("RCCL's bootstrap did not come back on this box (%s): %s" % (name, r.stderr[-300:]))      # the box's, not the code's (kg_comm.hip: rccl_boot_call)
        assert r.returncode == 0, (name, r.stdout[-1500:], r.stderr[-3000:])
        return r
    go(["comp", "-m27", "-H", "3000000", "-o", "one", "lib_R?.fq", "asm.fa"], "comp", base_env)
    go(["hist", "-m27", "-H", "3000000", "-o", "one.hist", "lib_R1.fq", "lib_R2.fq"], "hist", base_env)
    go(["gcp", "-m27", "-H", "3000000", "-o", "one_gcp", "lib_R1.fq", "lib_R2.fq"], "gcp", base_env)
    e = dict(base_env, **env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("KATGPU_COMM_INIT_TIMEOUT_S", "60")
    r = go(["comp", "--gpus", str(gpus), "-m27", "-H", "3000000", "-o", "many", "lib_R?.fq", "asm.fa"], "comp --gpus", e)
    if gpus > 1:
        assert "Multi-GPU: %d ranks, transport shm" % gpus in r.stdout
    go(["hist", "--gpus=%d" % gpus, "-m27", "-H", "3000000", "-o", "many.hist", "lib_R1.fq", "lib_R2.fq"], "hist --gpus", e)
    go(["gcp", "--gpus", str(gpus), "-m27", "-H", "3000000", "-o", "many_gcp", "lib_R1.fq", "lib_R2.fq"], "gcp --gpus", e)
    for a, b in (("one-main.mx", "many-main.mx"), ("one.hist", "many.hist"), ("one_gcp.mx", "many_gcp.mx")):
        x, y = (tmp_path / a).read_bytes(), (tmp_path / b).read_bytes()
        assert x.replace(b"one", b"many") == y.replace(b"one", b"many"), (a, b)
    sx, sy = (tmp_path / "one.stats").read_text(), (tmp_path / "many.stats").read_text()
    assert sx == sy
    # k > 32: the wide exchange (records all to all, the table refilled) behind the same switch
    go(["comp", "-m41", "-H", "3000000", "-o", "w_one", "lib_R?.fq", "asm.fa"], "comp k=41", base_env)
    go(["comp", "--gpus", str(gpus), "-m41", "-H", "3000000", "-o", "w_many", "lib_R?.fq", "asm.fa"], "comp k=41 --gpus", e)
    assert (tmp_path / "w_one-main.mx").read_bytes().replace(b"w_one", b"w_many") == (tmp_path / "w_many-main.mx").read_bytes()
    assert (tmp_path / "w_one.stats").read_text() == (tmp_path / "w_many.stats").read_text()
    # -d: the ranks' owned k-mers gathered into the one sorted .jf the plain run writes (InputHandler::dump, lib/src/input_handler.cc:221-243)
    go(["hist", "-d", "-m27", "-H", "3000000", "-o", "d_one.hist", "lib_R1.fq"], "hist -d", base_env)
    go(["hist", "--gpus", str(gpus), "-d", "-m27", "-H", "3000000", "-o", "d_many.hist", "lib_R1.fq"], "hist -d --gpus", e)
    def jf(name):                                                           # (header without its "time" field, records)
        import re
        b = (tmp_path / name).read_bytes()
        h = int(b[:9])
        return re.sub(rb'"time":"[^"]*"', b"", b[9:9 + h]), b[9 + h:]
    assert jf("d_one.hist-hash.jf27") == jf("d_many.hist-hash.jf27")
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".part")]
    go(["comp", "-d", "-m41", "-H", "3000000", "-o", "dw_one", "lib_R1.fq", "asm.fa"], "comp -d k=41", base_env)
    go(["comp", "--gpus", str(gpus), "-d", "-m41", "-H", "3000000", "-o", "dw_many", "lib_R1.fq", "asm.fa"], "comp -d k=41 --gpus", e)
    for i in (1, 2):
        assert jf("dw_one-hash%d.jf41" % i) == jf("dw_many-hash%d.jf41" % i)
