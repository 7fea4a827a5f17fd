"""The production ingest at size (what replaces InputHandler::count, lib/src/input_handler.cc:180-202, for large plain files): `katgpu comp`
on > 8 GB of FASTQ (two files) + a FASTA assembly in /dev/shm with the settings a deployment runs -- 16 reader threads, the files read
through a mapping (tmpfs), 8 MiB segments, 512 MiB scan batches, 4 GiB accumulation buffers, katgpu_reserve for the second table, no
test hook in the child's environment.  Its -main.mx and .stats must equal, BYTE FOR BYTE, the files written from the same reads counted
resident in HBM through the C ABI (count_bases_device -- the path bench.py's `value` times); and, so that the comparison is not only
product against product, the first 450 K reads of the same library go through the same device-scan ingest (as a file) and through the
resident path, and both tables must equal the ORACLE's dump."""
import os
import shutil
import subprocess
import tempfile
import time

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, L = 27, 150
N_READS = 26_000_000                       # 2 files x 13 M records x 318 B = 8.27 GB
GENOME = 120_000_000
ERR_PPM = 2000
G_SEED, R_SEED = 4242, 7


def _room(path):
    st = os.statvfs(path)
    return st.f_bavail * st.f_frsize


def test_katgpu_comp_production_ingest_equals_the_resident_path_and_the_oracle(engine, ko):
    import kat_amd
    if not os.path.isdir("/dev/shm") or _room("/dev/shm") < (14 << 30):
        pytest.skip("/dev/shm cannot hold the 8.5 GB of input files")
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    tmp = tempfile.mkdtemp(prefix="katgpu_ingest_", dir="/dev/shm")
    t_all = time.time()
    try:
        g = engine.synth_genome(GENOME, seed=G_SEED)
        paths = [os.path.join(tmp, "lib_R%d.fastq" % m) for m in (1, 2)]
        files = [open(p, "wb") for p in paths]
        first = None
        for lo in range(0, N_READS, 6_500_000):
            m = min(6_500_000, N_READS - lo)
            r = engine.synth_reads(g, GENOME, first_read=lo, n_reads=m, read_len=L, frag_len=350, err_ppm=ERR_PPM, seed=R_SEED)
            h = r.download().reshape(m, L + 1)
            r.free()
            if first is None:
                first = h[:450_000].copy()                                   # the library's first 450 K reads (225 K pairs: two files of 71.5 MB, above the device scan's 64 MiB threshold), 'N'-separated
            for mate in (0, 1):
                bench.write_fastq(files[mate], h[mate::2, :L], lo // 2, mate, L)
        for f in files:
            f.close()
        asm = g.download()
        asm_path = os.path.join(tmp, "asm.fa")
        clen = 1_000_000
        with open(asm_path, "wb") as f:
            for c in range((GENOME + clen - 1) // clen):
                seq = asm[c * clen:(c + 1) * clen]
                f.write(b">contig%d\n" % c)
                pad = (-seq.size) % 80
                lines = np.concatenate([seq, np.full(pad, ord("\n"), np.uint8)]).reshape(-1, 80)
                f.write(np.concatenate([lines, np.full((lines.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes().rstrip(b"\n") + b"\n")
        del asm
        assert sum(os.path.getsize(p) for p in paths) >= 8 * 10 ** 9
        engine.sync()
        engine.release_scratch()                                              # the child needs the device memory this process has parked

        # ---- the CLI with production settings: nothing of the test harness in its environment ----
        inst1 = N_READS * (L - K + 1)
        hint1 = int(bench.expected_distinct(inst1, GENOME, K, ERR_PPM) / 0.62) + (1 << 20)
        hint2 = int(GENOME / 0.62) + (1 << 20)
        env = {k_: v for k_, v in os.environ.items() if not k_.startswith("KATGPU_")}
        env["KATGPU_TIMING"] = "1"
        cmd = [exe, "comp", "-t", "16", "-m", str(K), "-H", str(hint1), "-I", str(hint2), "-o", os.path.join(tmp, "cli"), " ".join(paths), asm_path]
        t0 = time.time()
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        t_cli = time.time() - t0
        assert pr.returncode == 0, pr.stderr[-3000:]
        per_file = [ln for ln in pr.stderr.splitlines() if ln.startswith("katgpu_timing ") and '"file"' in ln]
        assert len(per_file) == 3, pr.stderr[-3000:]                         # all three inputs took the device-scan ingest ...
        assert all('"reader_threads": 16' in ln and "mapping" in ln for ln in per_file[:2]), per_file    # ... 16 readers, through a mapping (tmpfs)

        # ---- the same reads resident (the bench's path), written with the reference's formats ----
        r1 = engine.synth_reads(g, GENOME, first_read=0, n_reads=N_READS, read_len=L, frag_len=350, err_ppm=ERR_PPM, seed=R_SEED)
        t1 = engine.table(K, True, size_hint=hint1)
        t1.count_bases_device(r1.ptr, r1.nbytes)
        r1.free()
        a2 = engine.synth_genome(GENOME, seed=G_SEED, contig_len=clen)
        t2 = engine.table(K, True, size_hint=hint2, like=t1)
        t2.count_bases_device(a2.ptr, a2.nbytes)
        a2.free()
        mx, cc, sp = kat_amd.comp(t1, t2)
        assert int(cc[0]) == inst1 and int(cc[1]) == GENOME - (GENOME // clen) * (K - 1)
        ko.write_comp(os.path.join(tmp, "res"), K, paths, [asm_path], 1001, 1001, mx, cc, sp)
        for suffix in ("-main.mx", ".stats"):
            got, want = open(os.path.join(tmp, "cli" + suffix), "rb").read(), open(os.path.join(tmp, "res" + suffix), "rb").read()
            assert got == want, "cli%s differs from the resident path's" % suffix
        t1.free(); t2.free()

        # ---- the anchor: the library's first 450 K reads, through the same ingest as a file and resident, against the oracle ----
        small = [os.path.join(tmp, "first_R%d.fastq" % m) for m in (1, 2)]
        for mate in (0, 1):
            with open(small[mate], "wb") as f:
                bench.write_fastq(f, first[mate::2, :L], 0, mate, L)
        ot = ko.Table(K, True).count_bases(first.reshape(-1), threads=8)
        want_k, want_c = ot.dump_sorted()
        assert all(os.path.getsize(p) >= (64 << 20) for p in small)            # (KATGPU_SCAN_MIN_BYTES' default: these take the device scan too)
        ft = engine.count(small, K, True)
        gk, gc = ft.dump_sorted()
        assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c), "files -> table differs from the oracle"
        rb = engine.alloc(first.size)
        rb.upload(first.reshape(-1))
        rt = engine.table(K, True, size_hint=1 << 26).count_bases_device(rb.ptr, first.size)
        gk, gc = rt.dump_sorted()
        assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c), "resident table differs from the oracle"
        for x in (ft, rt, rb, g):
            x.free()
        print("production ingest at size: %.1f GB of FASTQ, katgpu comp %.2f s, whole test %.0f s" % (sum(os.path.getsize(p) for p in paths) / 1e9, t_cli, time.time() - t_all))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
