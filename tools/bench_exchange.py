"""Diagnostic: the device-side cost of the multi-GPU exchange at full size on ONE GPU (no wire).
    KATGPU_ARENA_FRACTION=0.75 python tools/bench_exchange.py [--reads N] [--world W] [--chunks C]
Counts the bench's read set, extracts the table for W owners, empties it and applies all W parts back chunk by chunk:
the volume an owner merges in a W-rank run (it receives 1/W of every rank's records)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KATGPU_ARENA_FRACTION", "0.75")
import kat_amd  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=300_000_000)
    ap.add_argument("--genome", type=int, default=1_000_000_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--k", type=int, default=27)
    a = ap.parse_args()
    eng = kat_amd.Engine(0)
    g = eng.synth_genome(a.genome, seed=20260927)
    reads = eng.synth_reads(g, a.genome, first_read=0, n_reads=a.reads, read_len=150, frag_len=350, err_ppm=5000, seed=1)
    g.free()
    inst = a.reads * (150 - a.k + 1)
    t = eng.table(a.k, True, size_hint=int(bench.expected_distinct(inst, a.genome, a.k, 5000) / 0.62) + (1 << 20))
    t.count_bases_device(reads.ptr, reads.nbytes)
    before = t.stats()
    geo = t.geometry()
    R, W, C = geo.n_regions, a.world, a.chunks
    res = {"table": before, "regions": R, "region_slots": geo.region_slots, "free_before": eng.mem_info()[0]}

    def timed(name, fn):
        eng.sync()
        eng.profile_reset()
        t0 = time.perf_counter()
        out = fn()
        eng.sync()
        p = eng.profile()
        res[name] = {"wall_ms": round((time.perf_counter() - t0) * 1e3, 1), "kernel_ms": {k: round(v["ms"], 1) for k, v in p.items() if v["launches"]}}
        return out

    cnt = eng.alloc(4 * W * R)
    sizes = timed("extract_sizes", lambda: t.extract_sizes(W, cnt.ptr))
    total = int(sizes.sum())
    sc = eng.scratch(0)
    res["arena_GB"] = round(sc.capacity / 1e9, 1)
    need = 12 * total + 4096
    res["send_list_GB"] = round(need / 1e9, 1)
    base_ptr = eng.scratch(need).ptr
    keys_ptr, counts_ptr = base_ptr, base_ptr + (8 * total + 255) // 256 * 256
    big = timed("extract", lambda: t.extract(W, cnt.ptr, keys_ptr, counts_ptr))
    timed("clear", t.clear)
    m = cnt.download(np.uint32, W * R).reshape(W, R).astype(np.int64)
    off = np.concatenate([np.zeros((W, 1), np.int64), np.cumsum(m, axis=1)], axis=1)
    pbase = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    bounds = (np.arange(C + 1) * R) // C

    def merge_all():
        for c in range(C):
            lo, hi = int(bounds[c]), int(bounds[c + 1])
            t.merge_regions(lo, hi, [(keys_ptr + 8 * int(pbase[p] + off[p][lo]), counts_ptr + 4 * int(pbase[p] + off[p][lo]), cnt.ptr + 4 * (p * R + lo),
                                      int(off[p][hi] - off[p][lo]), geo.p1, geo.p2) for p in range(W)])
    timed("merge_regions", merge_all)
    t.merge_host(*big)
    after = t.stats()
    res["restored"] = after["distinct"] == before["distinct"] and after["total"] == before["total"]
    res["records"] = total
    timed("clear2", t.clear)

    def merge_direct():
        for p in range(W):
            t.merge_device32(keys_ptr + 8 * int(pbase[p]), counts_ptr + 4 * int(pbase[p]), int(sizes[p]))
    timed("merge_direct_atomics", merge_direct)
    t.merge_host(*big)
    res["restored_direct"] = t.stats()["distinct"] == before["distinct"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
