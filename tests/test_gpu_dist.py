"""Two processes, ONE GPU: the multi-GPU path end to end on the device tables (count shard -> owner partition ->
exchange -> merge -> comp on owned shards -> all-reduce), with gloo carrying the records through host memory because a
single GPU cannot host two RCCL ranks.  Only the transport differs from `bench.py --gpus N` (there: RCCL send/recv on
the arena-backed buffers).  The sharded result must be bit-identical to the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

K, G, N_READS, CONTIG = 27, 400000, 60000, 50000
K_RR = 31                                            # "rr31": BASELINE.json configs[4] -- k = 31, reads library 1 vs reads library 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode):
    import kat_amd
    from kat_amd import dist as kdist
    from kat_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    eng = kat_amd.Engine(0)
    k = K_RR if mode == "rr31" else K
    g = synth.genome(G, seed=11)
    lo, hi = kdist.shard_range(N_READS // 2, rank, world)
    reads = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=1)
    if mode == "rr31":                                                      # the second input is a read library too, sharded like the first
        asm = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=2)
    else:
        c_lo, c_hi = kdist.shard_range(G // CONTIG, rank, world)
        asm = synth.stream_of_contigs(g[c_lo * CONTIG:c_hi * CONTIG], CONTIG)
    rb, ab = eng.alloc(reads.size), eng.alloc(asm.size)
    rb.upload(reads)
    ab.upload(asm)
    # "same": every rank's table has the same region grid (the bench's case: runs applied region by region in LDS);
    # "mixed": rank 1 sized its table differently, so its records reach rank 0 through the direct path and vice versa
    hint = (1 << 22) if not (mode == "mixed" and rank == 1) else (1 << 24)
    t1 = eng.table(k, True, size_hint=hint).count_bases(rb)
    t2 = eng.table(k, True, size_hint=1 << 20, like=t1).count_bases(ab)
    t1.merge_host(np.array([12345], np.uint64), np.array([(1 << 33) + rank], np.uint64))      # travels out of band
    g1 = t1.geometry()
    o1 = kdist.exchange_merge(kdist.HipShard(t1, staged=True), min_chunks=5)                  # in place: o1.table is t1
    o2 = kdist.exchange_merge(kdist.HipShard(t2, staged=True), min_chunks=2)
    assert o1.table is t1 and (t1.geometry().p1, t1.geometry().p2) == (g1.p1, g1.p2)
    prof = eng.profile()
    assert prof["merge"]["launches"] > 0
    keys, _ = o1.table.dump_sorted()
    assert (kdist.owner_of(keys, k, world) == rank).all()
    mx, cc, sp = kat_amd.comp(o1.table, o2.table, 1.0, 1.0, 201, 101)
    h, gm = o1.table.hist(1, 300, 1), o1.table.gcp(1.0, 100)
    mx, cc, sp, h, gm = kdist.allreduce_u64([mx, cc, sp, h, gm], torch.device("cpu"))
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mx=mx, cc=cc, sp=sp, h=h, gm=gm)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("world,mode", [(2, "same"), (2, "mixed"), (3, "same"), (2, "rr31")])
def test_ranks_sharing_one_gpu_match_single_process(engine, ko, tmp_path, world, mode):
    from kat_amd import synth
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    k = K_RR if mode == "rr31" else K
    g = synth.genome(G, seed=11)
    o1 = ko.Table(k, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
    o1.add(12345, world * (1 << 33) + world * (world - 1) // 2)
    o2 = ko.Table(k, True).count_bases(synth.reads(g, 0, N_READS, seed=2) if mode == "rr31" else synth.stream_of_contigs(g, CONTIG))
    mx, cc, sp = ko.comp(o1, o2, 1.0, 1.0, 201, 101)
    assert np.array_equal(got["cc"], cc) and np.array_equal(got["mx"], mx) and np.array_equal(got["sp"], sp)
    assert np.array_equal(got["h"], o1.hist(1, 300, 1)) and np.array_equal(got["gm"], o1.gcp(1.0, 100))


def _nccl_worker(rank, port, out_dir):
    """One rank, backend nccl (= RCCL): every line of the non-staged branch -- torch views of the arena and of the region-count
    matrix as collective operands, the device-side sets, the in-place protocol end to end -- runs on device tensors."""
    import kat_amd
    from kat_amd import dist as kdist
    from kat_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = kat_amd.Engine(0)
    g = synth.genome(G, seed=11)
    res = {}
    for k, canonical in ((27, True), (31, True), (32, False)):
        stream = np.concatenate([synth.reads(g, 0, N_READS, seed=1), np.frombuffer(b"N" + b"T" * 400 + b"N", np.uint8)])
        t1 = eng.table(k, canonical, size_hint=1 << 22).count_bases(stream)
        t1.merge_host(np.array([12345], np.uint64), np.array([(1 << 33) + 5], np.uint64))      # travels out of band
        before = t1.dump_sorted()
        geo = t1.geometry()
        sh = kdist.exchange_merge(kdist.HipShard(t1, staged=False), min_chunks=5, force=True)
        assert sh.table is t1 and (t1.geometry().p1, t1.geometry().p2, t1.geometry().n_regions) == (geo.p1, geo.p2, geo.n_regions)
        after = t1.dump_sorted()
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1]), ("in-place exchange changed the table", k)
        h = t1.hist(1, 300, 1)
        (h2,) = kdist.allreduce_u64([h], dev)                                                    # ncclAllReduce on a device tensor
        assert np.array_equal(h, h2)
        res["h%d" % k] = h2
    # wide tables: partition -> (no peers) -> rebuild, on device tensors
    tw = eng.table(45, True, size_hint=1 << 21).count_bases(synth.reads(g, 0, 8000, seed=3))
    bw = tw.dump_sorted()
    tw2 = kdist.exchange_merge_wide(kdist.HipWideShard(tw, staged=False), force=True).table
    aw = tw2.dump_sorted()
    assert all(np.array_equal(x, y) for x, y in zip(bw, aw))
    x = torch.arange(8, dtype=torch.int64, device=dev)
    dist.all_reduce(x)
    dist.barrier()
    assert eng.profile()["merge"]["launches"] > 0
    np.savez(os.path.join(out_dir, "nccl1.npz"), **res)
    dist.destroy_process_group()
    eng.close()


def test_nccl_backend_single_rank_runs_the_device_branch(ko, tmp_path):
    """The RCCL transport cannot have two ranks on one GPU; what one GPU can show is the whole non-staged branch of
    kat_amd/dist.py with backend "nccl" and world_size 1 (force=True): all_gather / all_reduce / barrier on device tensors that
    are views of katgpu's arena, extraction -> clear -> region-by-region merge of the rank's own send list, the out-of-band
    records -- and the table it leaves must be the table it started from."""
    from kat_amd import synth
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl1.npz")
    g = synth.genome(G, seed=11)
    stream = np.concatenate([synth.reads(g, 0, N_READS, seed=1), np.frombuffer(b"N" + b"T" * 400 + b"N", np.uint8)])
    for k, canonical in ((27, True), (31, True), (32, False)):
        o = ko.Table(k, canonical).count_bases(stream, threads=4)
        o.add(12345, (1 << 33) + 5)
        assert np.array_equal(got["h%d" % k], o.hist(1, 300, 1))


def _records(t, n_parts):
    """(sizes, region-count matrix, keys, counts32, big) of a table's extraction, downloaded."""
    eng = t.engine
    R = t.geometry().n_regions
    cnt = eng.alloc(4 * n_parts * R)
    sizes = t.extract_sizes(n_parts, cnt.ptr)
    total = int(sizes.sum())
    dk, dc = eng.alloc(8 * max(total, 1)), eng.alloc(4 * max(total, 1))
    big = t.extract(n_parts, cnt.ptr, dk.ptr, dc.ptr)
    return sizes, cnt, dk, dc, big, total


@pytest.mark.parametrize("k,canonical,n_parts", [(27, True, 3), (32, False, 8), (15, True, 1)])
def test_extract_clear_merge_roundtrip(engine, ko, k, canonical, n_parts):
    """katgpu_table_extract_* / _clear / _merge_regions on one GPU: the records are grouped by owner, ordered by region and
    complete; applying every part's runs to the emptied table restores it bit for bit."""
    from kat_amd import dist as kdist
    from kat_amd import synth
    g = synth.genome(300000, seed=5)
    stream = np.concatenate([synth.reads(g, 0, 20000, seed=2), np.frombuffer(b"N" + b"T" * 600 + b"N", np.uint8)])
    t = engine.table(k, canonical, size_hint=1 << 21).count_bases(stream)
    t.merge_host(np.array([77, 78], np.uint64), np.array([(1 << 32) + 9, 0xFFFFFFFF], np.uint64))     # one above 32 bits, one just inside
    o = ko.Table(k, canonical).count_bases(stream)
    o.add(77, (1 << 32) + 9)
    o.add(78, 0xFFFFFFFF)
    want_k, want_c = o.dump_sorted()
    geo = t.geometry()
    R = geo.n_regions
    sizes, cnt, dk, dc, (bk, bc), total = _records(t, n_parts)
    keys, c32 = dk.download(np.uint64, total), dc.download(np.uint32, total)
    m = cnt.download(np.uint32, n_parts * R).reshape(n_parts, R).astype(np.int64)
    assert np.array_equal(m.sum(1), sizes.astype(np.int64))
    base = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    for p in range(n_parts):                                               # grouped by owner
        kk = keys[base[p]:base[p + 1]]
        assert (kdist.owner_of(kk, k, n_parts) == p).all() if n_parts > 1 else True
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    full = {int(a): int(b) for a, b in zip(keys, c32)}
    for a, b in zip(bk, bc):
        assert int(a) == int(ones) or full[int(a)] == 0                    # the in-band record of a big count carries 0
        full[int(a)] = int(b)
    assert 77 in [int(x) for x in bk] and 78 not in [int(x) for x in bk]
    got = sorted(full.items())
    assert [a for a, _ in got] == [int(x) for x in want_k] and [b for _, b in got] == [int(x) for x in want_c]
    # restore: every part is one source ordered by this table's regions
    t.clear()
    assert t.stats()["distinct"] == 0
    srcs = [(dk.ptr + 8 * int(base[p]), dc.ptr + 4 * int(base[p]), cnt.ptr + 4 * p * R, int(sizes[p]), geo.p1, geo.p2) for p in range(n_parts)]
    half = R // 2
    if half:                                                               # two chunks of consecutive regions
        off = np.concatenate([np.zeros((n_parts, 1), np.int64), np.cumsum(m, axis=1)], axis=1)
        for lo, hi in ((0, half), (half, R)):
            t.merge_regions(lo, hi, [(dk.ptr + 8 * int(base[p] + off[p][lo]), dc.ptr + 4 * int(base[p] + off[p][lo]), cnt.ptr + 4 * (p * R + lo),
                                      int(off[p][hi] - off[p][lo]), geo.p1, geo.p2) for p in range(n_parts)])
    else:
        t.merge_regions(0, R, srcs)
    t.merge_host(bk, bc)
    gk, gc = t.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    assert engine.profile()["merge"]["launches"] > 0
    # a second application doubles every count (adds into occupied regions), through the direct path this time
    t.merge_regions(0, R, [(s_[0], s_[1], None, s_[3], 0, 0) for s_ in srcs])
    t.merge_host(bk, bc)
    gk, gc = t.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c * np.uint64(2))
    for b in (cnt, dk, dc):
        b.free()
    t.free()


def test_merge_regions_overflowing_regions_take_the_direct_path(engine, ko):
    """Owner regions too small for what arrives: k_merge_apply defers them, the host makes room and inserts directly."""
    from kat_amd import synth
    k = 27
    stream = synth.reads(synth.genome(200000, seed=9), 0, 15000, seed=3)
    a = engine.table(k, True, size_hint=1 << 21).count_bases(stream)
    geo = a.geometry()
    sizes, cnt, dk, dc, big, total = _records(a, 1)
    b = engine.table(k, True, size_hint=1 << 16, like=a)                   # same grid, regions of 256 slots (the smallest a common grid is kept for)
    gb = b.geometry()
    assert (gb.p1, gb.p2) == (geo.p1, geo.p2) and gb.region_slots < geo.region_slots
    b.merge_regions(0, geo.n_regions, [(dk.ptr, dc.ptr, cnt.ptr, total, geo.p1, geo.p2)])
    want_k, want_c = ko.Table(k, True).count_bases(stream).dump_sorted()
    gk, gc = b.dump_sorted()
    assert np.array_equal(gk, want_k) and np.array_equal(gc, want_c)
    for x in (cnt, dk, dc):
        x.free()
    a.free(); b.free()
