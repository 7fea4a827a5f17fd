"""Diagnostic: wall time of `katgpu sect` end to end (count the FASTA, then profile it against its own hash), with the
oracle's ko_sect timed on a sample for scale and byte-compared.    python tools/bench_sect_cli.py [--bases N] [--threads T]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "kat_amd", "bin", "katgpu")


def write_fasta(path, n_bases, contig, seed):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i in range(0, n_bases, contig):
            m = min(contig, n_bases - i)
            s = rng.choice(np.frombuffer(b"ACGT", np.uint8), m)
            s[rng.integers(0, m, max(1, m // 5000))] = ord("N")
            f.write(b">contig%d len=%d\n" % (i // contig, m))
            rows = np.full(((m + 79) // 80, 81), ord("\n"), np.uint8)
            flat = np.zeros(rows.shape[0] * 80, np.uint8)
            flat[:m] = s
            rows[:, :80] = flat.reshape(-1, 80)
            out = rows.reshape(-1)
            tail = rows.shape[0] * 80 - m
            f.write(out[: out.size - tail - 1].tobytes() + b"\n" if tail else out.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=200_000_000)
    ap.add_argument("--sample", type=int, default=5_000_000)
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        big, small = os.path.join(d, "big.fa"), os.path.join(d, "small.fa")
        write_fasta(big, a.bases, 1_000_000, 1)
        write_fasta(small, a.sample, 100_000, 2)
        res = {}
        for tag, fa, extra in (("small", small, []), ("big", big, []), ("big_stats_only", big, ["-n"])):
            t0 = time.perf_counter()
            r = subprocess.run([EXE, "sect", "-m", "27", "-t", str(a.threads), "-H", str(2 * a.bases), "-o", os.path.join(d, tag)] + extra + [fa, fa],
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            res[tag] = {"seconds": round(dt, 2), "stdout": [ln for ln in r.stdout.splitlines() if "Time taken" in ln or "runtime" in ln]}
        from oracle import koracle as ko
        t0 = time.perf_counter()
        t = ko.Table(27, True).count_files([small])
        t1 = time.perf_counter()
        ko.sect(t, small, os.path.join(d, "want"))
        t2 = time.perf_counter()
        same = all(open(os.path.join(d, "small" + s), "rb").read() == open(os.path.join(d, "want" + s), "rb").read() for s in ("-counts.cvg", "-stats.tsv"))
        res["oracle_small"] = {"count_s": round(t1 - t0, 2), "sect_s": round(t2 - t1, 2), "identical": same}
        res["bases"] = a.bases
        res["sample"] = a.sample
        print(json.dumps(res))


if __name__ == "__main__":
    main()
