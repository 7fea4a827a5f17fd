"""One rank of tests/test_gpu_comm.py: count a shard, katgpu_exchange_merge (native: kg_comm.hip), reduce, katgpu_allreduce_u64.
argv: rank world id_file out_dir mode"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kat_amd  # noqa: E402
from kat_amd import dist as kdist  # noqa: E402
from kat_amd import synth  # noqa: E402

K, G, N_READS, CONTIG = 27, 400000, 60000, 50000


def main():
    rank, world, id_file, out_dir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    eng = kat_amd.Engine(0)
    if rank == 0:
        cid = kat_amd.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(cid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 120, "no id from rank 0"
            time.sleep(0.01)
        cid = open(id_file, "rb").read()
    comm = kat_amd.Comm(eng, rank, world, cid)
    if os.environ.get("KATGPU_TEST_STALL_RANK") == str(rank):                    # (test_a_dead_peer_is_told_from_its_heartbeat: alive, but not coming)
        open(os.path.join(out_dir, "stalled.%d" % rank), "w").close()
        time.sleep(float(os.environ.get("KATGPU_TEST_STALL_S", "600")))
    k = 31 if mode == "rr31" else 45 if mode == "wide45" else K
    wide = k > 32
    g = synth.genome(G, seed=11)
    lo, hi = kdist.shard_range(N_READS // 2, rank, world)
    reads = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=1)
    if mode == "rr31":
        asm = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=2)
    else:
        c_lo, c_hi = kdist.shard_range(G // CONTIG, rank, world)
        asm = synth.stream_of_contigs(g[c_lo * CONTIG:c_hi * CONTIG], CONTIG)
    rb, ab = eng.alloc(reads.size), eng.alloc(max(asm.size, 16))
    rb.upload(reads)
    ab.upload(asm)
    hint = (1 << 22) if not (mode == "mixed" and rank == 1) else (1 << 24)      # "mixed": rank 1's grid differs: its records take the direct path
    split = os.environ.get("KATGPU_TEST_SPLIT_EXCHANGE") == "1"                  # katgpu_exchange_begin ... count the second input ... katgpu_exchange_finish
    t1 = eng.table(k, True, size_hint=hint).count_bases_device(rb.ptr, reads.size)
    if split:
        if wide:
            t1.merge_host_wide([0], [12345], [(1 << 33) + rank])
        else:
            t1.merge_host(np.array([12345], np.uint64), np.array([(1 << 33) + rank], np.uint64))
        g1 = None if wide else t1.geometry()
        comm.exchange_begin(t1)                                                  # table 1's records travel ...
    t2 = eng.table(k, True, size_hint=1 << 20, like=None if wide else t1).count_bases_device(ab.ptr, asm.size)     # ... while input 2 is counted (in the arena)
    if split:
        eng.sync()
        if os.environ.get("KATGPU_TEST_SPLIT_TWO") == "1":                       # ... and table 2's records travel while table 1's are applied
            comm.exchange_begin(t2)
            comm.exchange_finish(t1)
            comm.exchange_finish(t2)
        else:
            comm.exchange_finish(t1)
            comm.exchange_merge(t2)
        if not wide:
            assert (t1.geometry().p1, t1.geometry().p2) == (g1.p1, g1.p2)
            keys, counts = t1.dump_sorted()
            assert (kdist.owner_of(keys, k, world) == rank).all()
    elif wide:                                                                   # k > 32: records (hi, lo, count) all to all, the table refilled
        t1.merge_host_wide([0], [12345], [(1 << 33) + rank])
        before = t1.dump_sorted_wide() if world == 1 else None
        comm.exchange_merge(t1)
        comm.exchange_merge(t2)
        hi, lo, counts = t1.dump_sorted_wide()
        assert (kdist.owner_of_wide(hi, lo, k, world) == rank).all()
        if world == 1:
            assert all(np.array_equal(a, b) for a, b in zip(before, (hi, lo, counts)))
    else:
        t1.merge_host(np.array([12345], np.uint64), np.array([(1 << 33) + rank], np.uint64))      # travels out of band
        g1 = t1.geometry()
        before = t1.dump_sorted() if world == 1 else None
        comm.exchange_merge(t1)
        w1 = comm.stats()
        if rank == 0:                                                            # (the first table's records alone: test_records_travel_in_nine_bytes_...)
            print("wire after table 1:", {q: w1[q] for q in ("records_sent", "record_bytes_sent", "records_packed")})
        comm.exchange_merge(t2)
        assert (t1.geometry().p1, t1.geometry().p2) == (g1.p1, g1.p2)
        keys, counts = t1.dump_sorted()
        assert (kdist.owner_of(keys, k, world) == rank).all()
        if world == 1:                                                           # the protocol on the rank's own records: the table it found
            assert np.array_equal(before[0], keys) and np.array_equal(before[1], counts)
    assert eng.profile()["merge"]["launches"] > 0
    mx, cc, sp = kat_amd.comp(t1, t2, 1.0, 1.0, 201, 101)
    h, gm = t1.hist(1, 300, 1), t1.gcp(1.0, 100)
    mx, cc, sp, h, gm = comm.allreduce_u64([mx, cc, sp, h, gm])
    st = comm.stats()
    assert st["merge_calls"] > 0 and (world == 1 or st["bytes_sent"] > 0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mx=mx, cc=cc, sp=sp, h=h, gm=gm)
        print("transport:", comm.transport, "|", comm.transport_note, "| devices:", comm.distinct_devices, "|", st)
    comm.barrier()
    comm.free()
    eng.close()


if __name__ == "__main__":
    main()
