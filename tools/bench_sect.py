"""Diagnostic: throughput of the `kat sect` lookup kernel (k_profile) on the bench workload's tables.
    python tools/bench_sect.py [--reads N] [--genome G]
Counts synthetic PE reads into a table, then profiles the synthetic assembly (device-resident) against it."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kat_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--genome", type=int, default=1_000_000_000)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    eng = kat_amd.Engine(0)
    def say(m):
        eng.sync()
        print(m, file=sys.stderr, flush=True)
    g = eng.synth_genome(a.genome, seed=20260927)
    reads = eng.synth_reads(g, a.genome, first_read=0, n_reads=a.reads, read_len=150, frag_len=350, err_ppm=5000, seed=1)
    asm = eng.synth_genome(a.genome, seed=20260927, contig_len=100_000)
    g.free()
    say('synth done')
    t = eng.table(a.k, True, size_hint=int(2.2 * a.genome))
    t.count_bases_device(reads.ptr, reads.nbytes)
    say('count done')
    reads.free()
    eng.release_scratch()
    st = t.stats()
    n_out = asm.nbytes - a.k + 1
    out = eng.alloc(n_out * 8)
    say('alloc done %d' % n_out)
    res = []
    for canon in (True,):
        eng.sync()
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            t.profile_device(asm, asm.nbytes, out, canon)
        say('profiled')
        dt = (time.perf_counter() - t0) / a.reps
        p = eng.profile()["profile"]
        res.append({"canonical": canon, "wall_ms": dt * 1e3, "kernel_ms": p["ms"] / max(1, p["launches"]),
                    "G_lookups_per_s": n_out / (p["ms"] / max(1, p["launches"])) / 1e6})
    import numpy as np
    sample = out.download(np.uint64, 1 << 20)
    print(json.dumps({"table": st, "positions": n_out, "runs": res, "sample_mean_count": float(sample.mean()),
                      "sample_zero_frac": float((sample == 0).mean())}))


if __name__ == "__main__":
    main()
