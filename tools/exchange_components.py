"""What katgpu_exchange_merge costs on the GPU side, measured at full size on ONE device: a communicator of one rank runs the whole protocol
on the rank's own send list -- extraction (two sweeps of the table + the send list), the table emptied, the region-by-region merge of every
record -- with nothing on the wire.  These are the two terms of DESIGN.md section 7's model that a single GPU can measure; the wire term stays
arithmetic.  python tools/exchange_components.py [--reads 150000000] [--k 31] (config 5's per-GPU shard at the defaults: one of its two tables)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kat_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=75_000_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--genome", type=int, default=1_000_000_000)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--err-ppm", type=int, default=2000)
    a = ap.parse_args()
    eng = kat_amd.Engine(0)
    comm = kat_amd.Comm(eng, 0, 1, kat_amd.Comm.unique_id())
    g = eng.synth_genome(a.genome, seed=20260927)
    reads = eng.synth_reads(g, a.genome, first_read=0, n_reads=a.reads, read_len=150, frag_len=350, err_ppm=a.err_ppm, seed=1)
    g.free()
    for packed in ("1", "0"):
        os.environ["KATGPU_COMM_PACKED_RECORDS"] = packed      # (read when the library is loaded: the second pass only says so if it could not)
        import bench
        t = eng.table(a.k, True, size_hint=int(bench.expected_distinct(a.reads * (150 - a.k + 1), a.genome, a.k, a.err_ppm) / 0.62) + (1 << 20))     # (as bench.py sizes config 5's tables)
        t.count_bases_device(reads.ptr, reads.nbytes)
        eng.sync()
        st = t.stats(want_total=False)
        geo = t.geometry()
        print("table: k = %d, %d distinct in %d slots of %d bytes (%.1f GB), %d regions of %d slots" % (a.k, st["distinct"], st["capacity"], t.slot_bytes(), st["capacity"] * t.slot_bytes() / 1e9,
                                                                                                      geo.n_regions, geo.region_slots), flush=True)
        for r in range(a.repeats):
            b = comm.stats()
            eng.profile_reset()
            t0 = time.perf_counter()
            comm.exchange_merge(t)
            eng.sync()
            dt = (time.perf_counter() - t0) * 1e3
            c = comm.stats()
            print("  exchange of one rank's own records (%s-byte records asked for: %s): %.1f ms -- extract %.1f, merge %.1f, wire calls %.1f; %d merge calls" % (
                "9" if packed == "1" else "12", "packed" if c["records_packed"] else "key + count", dt, c["extract_ms"] - b["extract_ms"], c["merge_ms"] - b["merge_ms"],
                c["exchange_ms"] - b["exchange_ms"], c["merge_calls"] - b["merge_calls"]), flush=True)
            pr = eng.profile()
            print("      of that in kernels (HIP events): " + ", ".join("%s %.1f ms in %d launches" % (q, pr[q]["ms"], pr[q]["launches"]) for q in pr if pr[q]["launches"]), flush=True)
        assert t.stats(want_total=False)["distinct"] == st["distinct"]
        t.free()
        eng.release_scratch()
        break                                                   # (the switch is read once per process: run again with KATGPU_COMM_PACKED_RECORDS=0 for the other form)
    comm.free()
    eng.close()


if __name__ == "__main__":
    main()
