"""The gzip team (kat_amd/csrc/kg_pgzip.cpp) on this host, without the GPU: a FASTQ file of --reads records as ONE gzip member (bench.py's
pigz-shaped writer), inflated through katgpu_inflate_file (bytes checked against the members' CRC-32, not kept) by teams of several sizes,
next to one zlib stream on a bounded sample.  python tools/pgz_host_bench.py [--reads N] [--threads 8,16,...] [--chunks-mb 4]"""
import argparse
import os
import shutil
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kat_amd import binding as kb  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=24_000_000)
    ap.add_argument("--threads", default="1,8,16,24,32,48,64,96")
    ap.add_argument("--chunks-mb", default="4")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--variants", action="store_true")
    a = ap.parse_args()
    L = 150
    for f in ("enabled", "defrag"):
        try:
            print("transparent_hugepage/%s:" % f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
        except OSError:
            pass
    print("cpus shown:", os.cpu_count(), "usable (affinity, cgroup quota):", bench.effective_cpus(), "| cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-", "numa_balancing:", open("/proc/sys/kernel/numa_balancing").read().strip() if os.path.exists("/proc/sys/kernel/numa_balancing") else "?", flush=True)
    d = "/dev/shm/katgpu_pgz_bench_%d" % os.getpid()
    os.makedirs(d, exist_ok=True)
    try:
        rng = np.random.default_rng(1)
        g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 5_000_000)]
        t0 = time.time()
        with open(d + "/w.fastq", "wb") as f:
            for lo in range(0, a.reads, 1_000_000):
                m = min(1_000_000, a.reads - lo)
                st = rng.integers(0, g.size - L, m)
                bench.write_fastq(f, g[st[:, None] + np.arange(L)[None, :]], lo, 0, L)
        print("written in %.1f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        w, t = bench.gzip_records_file(d + "/w.fastq", d + "/w.fastq.gz", a.reads, 2 * L + 18, L, a.level, min(bench.effective_cpus(), 96))
        os.unlink(d + "/w.fastq")
        print("one gzip member: %.2f GB from %.2f GB of FASTQ (level %d) in %.1f s" % (w / 1e9, t / 1e9, a.level, time.time() - t0), flush=True)
        blob = open(d + "/w.fastq.gz", "rb").read(192 << 20)
        t0 = time.time()
        o = zlib.decompressobj(31).decompress(blob)
        dt = time.time() - t0
        one = len(o) / dt / 1e9
        print("one zlib stream (first %d MB): %.3f GB/s of FASTQ, %.3f GB/s compressed" % (len(blob) >> 20, one, len(blob) / dt / 1e9), flush=True)
        del blob, o
        if a.variants:                                              # each in a process of its own: huge pages off, one socket's cores, ...
            import subprocess
            one = ("import os, sys, time; sys.path.insert(0, %r); from kat_amd import binding as kb; t0 = time.time(); "
                   "nb = kb.inflate_file(%r, keep=False); dt = time.time() - t0; import resource; u = resource.getrusage(resource.RUSAGE_SELF); "
                   "print('%%.2f GB/s of FASTQ | %%d minor faults, %%d waits, %%d involuntary switches, %%.1f cpu-s in %%.2f s' %% (nb / dt / 1e9, u.ru_minflt, u.ru_nvcsw, u.ru_nivcsw, u.ru_utime + u.ru_stime, dt))" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), d + "/w.fastq.gz"))
            for T in [int(x) for x in a.threads.split(",")]:
                for tag, env, pre in (("", {}, []), ("no huge pages", {"KATGPU_PGZ_HUGE": "0"}, []), ("cpus 0-63", {}, ["taskset", "-c", "0-63"]), ("cpus 0-63,128-191", {}, ["taskset", "-c", "0-63,128-191"])):
                    r = subprocess.run(pre + [sys.executable, "-c", one],
                                       env=dict(os.environ, KATGPU_PGZ_THREADS=str(T), **env), capture_output=True, text=True, timeout=300)
                    print("team of %3d %-18s: %s | %s" % (T, tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "rc %d" % r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:200] if r.returncode else ""), flush=True)
            a.threads = ""
        for T in [int(x) for x in a.threads.split(",") if x]:
            for mb in [int(x) for x in a.chunks_mb.split(",")]:
                os.environ["KATGPU_PGZ_THREADS"] = str(T)
                os.environ["KATGPU_PGZ_CHUNK"] = str(mb << 20)
                best = 0.0
                for _ in range(2):
                    t0 = time.time()
                    nb = kb.inflate_file(d + "/w.fastq.gz", keep=False)
                    best = max(best, nb / (time.time() - t0) / 1e9)
                assert nb == t
                print("team of %3d, chunks of %2d MB: %.2f GB/s of FASTQ, %.2f GB/s compressed, %.1f x one zlib stream" % (T, mb, best, best * w / t, best / one), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":                                      # (bench.gzip_records_file starts worker processes that import this file)
    main()
