#!/usr/bin/env python3
"""BASELINE.json configs[3] from FILES to FILES at full size: `katgpu comp -m 27` on 300 M x 150 bp PE reads (two FASTQ files, 96 GB) vs the
1 Gbp assembly (FASTA, 1000 contigs) -- the span of the reference's "Total runtime" (src/comp.cc:750).  The files are generated on the
device, written to a RAM-backed directory (they do not fit this image's /tmp), and stay in the page cache: this times parse + PCIe + count
+ reduce + write, not storage.  Prints one JSON object; tools/profile_bench.sh stores it as profiles/rNN_e2e_config4.json."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import kat_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=300_000_000)
    ap.add_argument("--genome", type=int, default=1_000_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--gpus", type=int, default=0, help="pass --gpus N to katgpu (ranks share the devices there are)")
    a = ap.parse_args()
    k, L = 27, 150
    n = a.reads & ~1
    tmp = os.path.join(a.dir, "katgpu_e2e_cfg4_%d" % os.getpid())
    os.makedirs(tmp)
    try:
        t0 = time.perf_counter()
        eng = kat_amd.Engine(0)
        g = eng.synth_genome(a.genome, seed=20260927)
        paths = [os.path.join(tmp, "lib1_R%d.fastq" % m) for m in (1, 2)]
        files = [open(p, "wb") for p in paths]
        for lo in range(0, n, 8_000_000):
            m = min(8_000_000, n - lo)
            r = eng.synth_reads(g, a.genome, first_read=lo, n_reads=m, read_len=L, frag_len=350, err_ppm=2000, seed=1)
            h = r.download().reshape(m, L + 1)[:, :L]
            r.free()
            for mate in (0, 1):
                bench.write_fastq(files[mate], h[mate::2], lo // 2, mate, L)
        for f in files:
            f.close()
        asm = g.download()
        g.free()
        asm_path = os.path.join(tmp, "asm.fa")
        inst2 = 0
        with open(asm_path, "wb") as f:
            clen = 1_000_000
            for c in range((a.genome + clen - 1) // clen):
                seq = asm[c * clen:(c + 1) * clen]
                f.write(b">contig%d\n" % c)
                pad = (-seq.size) % 80
                lines = np.concatenate([seq, np.full(pad, ord("\n"), np.uint8)]).reshape(-1, 80)
                f.write(np.concatenate([lines, np.full((lines.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes().rstrip(b"\n") + b"\n")
                inst2 += max(0, seq.size - k + 1)
        del asm
        eng.close()
        t_gen = time.perf_counter() - t0
        inst = n * (L - k + 1) + inst2
        nbytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
        hint1 = int(bench.expected_distinct(n * (L - k + 1), a.genome, k, 2000) / 0.62) + (1 << 20)
        hint2 = int(a.genome / 0.62) + (1 << 20)
        exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
        cmd = [exe, "comp"] + (["--gpus", str(a.gpus)] if a.gpus else []) + ["-t", "16", "-m", str(k), "-H", str(hint1), "-I", str(hint2), "-o", os.path.join(tmp, "out"),
                                                                               " ".join(paths), asm_path]
        env = dict(os.environ, KATGPU_TRACE="1")
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env)
        dt = time.perf_counter() - t0
        stats = open(os.path.join(tmp, "out.stats")).read() if os.path.exists(os.path.join(tmp, "out.stats")) else ""
        res = {"what": "katgpu comp, files -> files, BASELINE.json configs[3] at full size", "returncode": pr.returncode,
               "command": " ".join(os.path.basename(c) if c.startswith(tmp) or c == exe else c for c in cmd),
               "reads": n, "genome_bp": a.genome, "k": k, "input_bytes": nbytes, "kmer_instances": inst, "seconds": round(dt, 3),
               "input_GB_per_s": round(nbytes / dt / 1e9, 2), "kmers_per_s": round(inst / dt, 1), "files_written_in_s": round(t_gen, 1),
               "span": "process start -> output files closed, inputs in the page cache (" + a.dir + ")",
               "phases": [l.strip() for l in pr.stdout.splitlines() if "Time taken" in l or "Total runtime" in l or "Multi-GPU" in l][:10],
               "trace": [l for l in pr.stderr.splitlines() if "[katgpu +" in l][:24],
               "stats_head": stats.splitlines()[:14], "host_ram_GB": os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") // 10 ** 9, "host_cores": os.cpu_count()}
        if pr.returncode:
            res["stderr_tail"] = pr.stderr[-1500:]
        print(json.dumps(res))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
