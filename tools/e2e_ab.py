#!/usr/bin/env python3
"""Same-box A/B of the files -> files leg: the inputs are written once (FASTQ pair + assembly in /dev/shm), then `katgpu comp` runs once per
environment given on the command line ("NAME=VAL,NAME=VAL" per run; "-" = the defaults).  Prints one line per run: wall seconds, k-mers/s,
and the per-file timing lines.   python tools/e2e_ab.py --reads 150000000 - KATGPU_FASTQ_STRIP=0 KATGPU_SCAN_THREADS=32"""
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
    ap.add_argument("--reads", type=int, default=150_000_000)
    ap.add_argument("--genome", type=int, default=1_000_000_000)
    ap.add_argument("--pause", type=float, default=12.0, help="seconds between runs: a process that starts right after another has freed ~100 GB of HBM waits seconds for the driver to scrub it (measured: 2.4-3.2 s)")
    ap.add_argument("runs", nargs="*", default=["-"])
    a = ap.parse_args()
    k, L = 27, 150
    n = a.reads & ~1
    tmp = "/dev/shm/katgpu_e2e_ab_%d" % os.getpid()
    os.makedirs(tmp)
    try:
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
        inst = n * (L - k + 1) + inst2
        hint1 = int(bench.expected_distinct(n * (L - k + 1), a.genome, k, 2000) / 0.62) + (1 << 20)
        hint2 = int(a.genome / 0.62) + (1 << 20)
        exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
        ref = None
        for i_run, run in enumerate(a.runs):
            if i_run:
                time.sleep(a.pause)
            env = {k_: v for k_, v in os.environ.items() if not k_.startswith("KATGPU_")}
            env.update(KATGPU_TIMING="1", KATGPU_TRACE="1")
            if run != "-":
                env.update(dict(kv.split("=", 1) for kv in run.split(",")))
            out = os.path.join(tmp, "out")
            cmd = [exe, "comp", "-t", "16", "-m", str(k), "-H", str(hint1), "-I", str(hint2), "-o", out, " ".join(paths), asm_path]
            t0 = time.perf_counter()
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
            dt = time.perf_counter() - t0
            stats = open(out + ".stats").read() if os.path.exists(out + ".stats") else ""
            cc = bench.parse_stats(stats)[:5] if stats else None
            if ref is None:
                ref = cc
            print("RUN %-44s rc=%d  %.3f s  %.2f G k-mers/s  counters %s%s" % (run, pr.returncode, dt, inst / dt / 1e9, cc, "" if cc == ref else "  DIFFERENT FROM THE FIRST RUN"))
            for ln in pr.stderr.splitlines():
                if ln.startswith("katgpu_timing ") and '"file"' in ln:
                    d = json.loads(ln[len("katgpu_timing "):])
                    print("    ", {q: d[q] for q in ("setup_ms", "wall_ms", "reader_wait_ms", "scan_ms", "counter_wait_ms", "counting_ms", "pread_ms_per_thread", "h2d_ms_per_thread", "reader_threads")}, d["read_by"][:40])
                elif ln.startswith("katgpu_timing "):
                    print("    ", ln[len("katgpu_timing "):])
                elif ln.startswith("[katgpu") and any(w in ln for w in ("alloc", "arena of", "scan buffers", "context on", "host strip of", "device scan of")):
                    print("       ", ln[:230])
            if pr.returncode:
                print(pr.stderr[-1500:])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
