#!/usr/bin/env python3
"""files -> table through katgpu_count_files' device scan (kg_scan.hip), for a sweep of reader threads / segment sizes / batch sizes.
Writes a FASTQ pair of --reads 150 bp reads to /tmp once, then times `katgpu hist` on it (page cache warm) per setting."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kat_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--settings", default="12:32:1024,24:32:1024,48:32:1024,24:8:1024,24:64:2048,16:32:512")
    a = ap.parse_args()
    eng = kat_amd.Engine(0)
    n, L, k = a.reads & ~1, 150, 27
    gs = max(10_000_000, n * 5)
    g = eng.synth_genome(gs, seed=99)
    tmp = tempfile.mkdtemp(prefix="katgpu_scan_")
    paths = [os.path.join(tmp, "lib_R%d.fastq" % m) for m in (1, 2)]
    files = [open(p, "wb") for p in paths]
    for lo in range(0, n, 8_000_000):
        m = min(8_000_000, n - lo)
        r = eng.synth_reads(g, gs, first_read=lo, n_reads=m, read_len=L, frag_len=350, err_ppm=2000, seed=5)
        h = r.download().reshape(m, L + 1)[:, :L]
        r.free()
        for mate in (0, 1):
            bench.write_fastq(files[mate], h[mate::2], lo // 2, mate, L)
    for f in files:
        f.close()
    g.free()
    eng.close()
    nbytes = sum(os.path.getsize(p) for p in paths)
    hint = int(bench.expected_distinct(n * (L - k + 1), gs, k, 2000) / 0.62) + (1 << 20)
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    for st in a.settings.split(","):
        thr, seg, bat = st.split(":")
        env = dict(os.environ, KATGPU_SCAN_THREADS=thr, KATGPU_SCAN_SEGMENT_MB=seg, KATGPU_SCAN_BATCH_MB=bat, KATGPU_TRACE="1")
        t0 = time.perf_counter()
        pr = subprocess.run([exe, "hist", "-m", str(k), "-H", str(hint), "-o", os.path.join(tmp, "out.hist")] + paths, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        tr = [l for l in pr.stderr.splitlines() if "[katgpu +" in l] + ["wall from spawn to exit: %.0f ms" % (dt * 1e3)]
        print("threads %s segment %s MiB batch %s MiB: %.2f s = %.1f GB/s (rc %d)\n   %s" % (thr, seg, bat, dt, nbytes / dt / 1e9, pr.returncode, "\n   ".join(tr)), flush=True)
    for p in paths + [os.path.join(tmp, "out.hist")]:
        if os.path.exists(p):
            os.unlink(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
