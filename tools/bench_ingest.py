"""Diagnostic: file -> table throughput of katgpu_count with the streaming parser vs the thread team (kg_ingest.hpp).
    python tools/bench_ingest.py [--gb 4] [--threads 32]
Writes a FASTQ of 150 bp reads to /tmp (a 64 MB random block, tiled), counts it both ways and compares the tables' totals."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kat_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=4.0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--k", type=int, default=27)
    a = ap.parse_args()
    path = "/tmp/katgpu_ingest_bench.fq"
    rng = np.random.default_rng(1)
    n_block = 200_000
    seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n_block, 150))
    rec = np.empty((n_block, 8 + 1 + 150 + 1 + 2 + 150 + 1), np.uint8)
    rec[:, :8] = np.frombuffer(b"@read/1 ", np.uint8)
    rec[:, 8] = ord("\n")
    rec[:, 9:159] = seqs
    rec[:, 159] = ord("\n")
    rec[:, 160:162] = np.frombuffer(b"+\n", np.uint8)
    rec[:, 162:312] = ord("I")
    rec[:, 312] = ord("\n")
    block = rec.tobytes()
    reps = max(1, int(a.gb * 1e9 / len(block)))
    with open(path, "wb") as f:
        for _ in range(reps):
            f.write(block)
    size = os.path.getsize(path)
    eng = kat_amd.Engine(0)
    res = {"file_GB": round(size / 1e9, 2), "reads": reps * n_block}
    for mode in ("stream", "team"):
        os.environ["KATGPU_INGEST_MIN_BYTES"] = str(1 << 60) if mode == "stream" else "0"
        os.environ["KATGPU_INGEST_THREADS"] = str(a.threads)
        eng.sync()
        t0 = time.perf_counter()
        t = eng.count([path], a.k, True, size_hint=40_000_000)
        eng.sync()
        dt = time.perf_counter() - t0
        st = t.stats()
        res[mode] = {"seconds": round(dt, 2), "GB_per_s": round(size / 1e9 / dt, 2), "M_kmers_per_s": round(st["total"] / dt / 1e6, 1),
                     "distinct": st["distinct"], "total": st["total"]}
        t.free()
    res["identical"] = res["stream"]["distinct"] == res["team"]["distinct"] and res["stream"]["total"] == res["team"]["total"]
    os.remove(path)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
