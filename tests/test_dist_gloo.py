"""world_size 2 / 3 / 4 gloo runs of the multi-GPU path on CPU.

The product's exchange is native (kg_comm.hip; tests/test_gpu_comm.py runs it on the GPU).  Here its protocol runs as the Python model
of kat_amd/dist.py over gloo, the per-rank table an oracle-backed stand-in (the HIP table needs a GPU): read sharding, owner
routing, the region-ordered chunked exchange with its ordering claims CHECKED, out-of-band records, result reduction -- and the
sharded answer must be bit-identical to the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kat_amd import dist as kdist
from kat_amd import synth
from kat_amd.dist import owner_of_wide  # noqa: F401



class OracleShard:
    """The duck type kat_amd.dist.exchange_merge is written against, backed by the CPU oracle table.  Its "region grid" is R buckets of a hash
    of the k-mer, so the region-ordered protocol (per-region counts, chunks of consecutive regions, runs per region) is
    exercised and CHECKED here: merge_chunk asserts that every aligned source really is ordered the way it claims."""

    def __init__(self, table, n_regions=8):
        self.table = table
        self.device = torch.device("cpu")
        self.R = n_regions
        self.chunks_seen = 0

    def _region(self, keys):
        return (kdist._mix64(keys) % np.uint64(self.R)).astype(np.int64)

    def geometry(self):
        return np.array([self.table.k, int(self.table.canonical), self.R, 1 << 20, self.R, 1], dtype=np.int64)

    def begin_exchange(self, n_parts):
        keys, counts = self.table.dump_sorted()
        part = kdist.owner_of(keys, self.table.k, n_parts)
        reg = self._region(keys)
        order = np.lexsort((keys, reg, part))                    # by owner, then region
        self._rec = (keys[order], counts[order])
        cnt = np.zeros((n_parts, self.R), np.int32)
        np.add.at(cnt, (part, reg), 1)
        return cnt.sum(1).astype(np.int64), torch.from_numpy(cnt)

    def exchange_buffers(self, total_send, set_records):
        mk = lambda n, dt: torch.empty(max(int(n), 1), dtype=dt)
        return {"send_keys": mk(total_send, torch.int64), "send_counts": mk(total_send, torch.int32),
                "recv": [(mk(set_records, torch.int64), mk(set_records, torch.int32)) for _ in range(2)]}

    def extract(self, n_parts, bufs):
        keys, counts = self._rec
        big = counts > np.uint64(0xFFFFFFFF)
        c32 = np.where(big, 0, counts).astype(np.uint32)
        bufs["send_keys"][:keys.size] = torch.from_numpy(keys.view(np.int64).copy())
        bufs["send_counts"][:keys.size] = torch.from_numpy(c32.view(np.int32).copy())
        return keys[big], counts[big]

    def clear(self):
        from oracle import koracle as ko
        self.table = ko.Table(self.table.k, self.table.canonical)

    def merge_chunk(self, g_lo, g_hi, sources, set_index):
        self.chunks_seen += 1
        for s in sources:
            n = int(s["n"])
            kk = s["keys"][:n].numpy().view(np.uint64)
            cc = s["counts"][:n].numpy().view(np.uint32)
            if s["rcnt"] is not None:                             # an aligned source: runs per region, in region order
                assert (int(s["p1"]), int(s["p2"])) == (self.R, 1)
                r = s["rcnt"].numpy().astype(np.int64)
                assert r.size == g_hi - g_lo and int(r.sum()) == n
                assert np.array_equal(self._region(kk), np.repeat(np.arange(g_lo, g_hi), r))
            for i in range(n):
                if cc[i]:
                    self.table.add(int(kk[i]), int(cc[i]))

    def merge_big(self, keys, counts):
        for k_, c_ in zip(keys, counts):
            self.table.add(int(k_), int(c_))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


K, G, N_READS, CONTIG = 21, 40000, 3000, 5000


def _worker(rank, world, port, out_dir, mixed_grids):
    from oracle import koracle as ko
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.genome(G, seed=11)
    lo, hi = kdist.shard_range(N_READS // 2, rank, world)                 # shard by read PAIR
    reads = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=1)
    n_contigs = G // CONTIG
    c_lo, c_hi = kdist.shard_range(n_contigs, rank, world)
    asm = synth.stream_of_contigs(g[c_lo * CONTIG:c_hi * CONTIG], CONTIG)
    t1 = ko.Table(K, True).count_bases(reads)
    t2 = ko.Table(K, True).count_bases(asm)
    t1.add(ko.encode("ACGT" * 5 + "A"), (1 << 33) + rank)                 # a count above 32 bits travels out of band
    regions = 8 if not (mixed_grids and rank == 1) else 5                  # mixed: rank 1's table has another grid
    s1 = kdist.exchange_merge(OracleShard(t1, regions), min_chunks=3)
    s2 = kdist.exchange_merge(OracleShard(t2, regions), min_chunks=1)
    assert s1.chunks_seen == 3 and s2.chunks_seen == 1
    o1, o2 = s1.table, s2.table
    # every key this rank now owns really is its own
    keys, _ = o1.dump_sorted()
    assert (kdist.owner_of(keys, K, world) == rank).all()
    mx, cc, sp = ko.comp(o1, o2, 1.0, 1.0, 101, 101)
    h = o1.hist(1, 200, 1)
    gm = o1.gcp(1.0, 100)
    mx, cc, sp, h, gm = kdist.allreduce_u64([mx, cc, sp, h, gm], torch.device("cpu"))
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mx=mx, cc=cc, sp=sp, h=h, gm=gm)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mixed_grids", [(2, False), (2, True), (3, False), (4, True)])
def test_sharded_exchange_matches_single_process(ko, tmp_path, world, mixed_grids):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mixed_grids), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    g = synth.genome(G, seed=11)
    t1 = ko.Table(K, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
    t1.add(ko.encode("ACGT" * 5 + "A"), world * (1 << 33) + world * (world - 1) // 2)      # every rank's out-of-band amount
    t2 = ko.Table(K, True).count_bases(synth.stream_of_contigs(g, CONTIG))
    mx, cc, sp = ko.comp(t1, t2, 1.0, 1.0, 101, 101)
    assert np.array_equal(got["mx"], mx) and np.array_equal(got["cc"], cc) and np.array_equal(got["sp"], sp)
    assert np.array_equal(got["h"], t1.hist(1, 200, 1)) and np.array_equal(got["gm"], t1.gcp(1.0, 100))


def test_owner_is_strand_symmetric_and_balanced(ko):
    rng = np.random.default_rng(0)
    k = 27
    keys = rng.integers(0, 2 ** 54, size=20000, dtype=np.uint64)
    rc = np.array([ko.revcomp(int(x), k) for x in keys], dtype=np.uint64)
    assert np.array_equal(kdist._revcomp(keys, k), rc)
    for n in (2, 3, 8):
        o = kdist.owner_of(keys, k, n)
        assert np.array_equal(o, kdist.owner_of(rc, k, n))
        assert o.min() == 0 and o.max() == n - 1
        assert np.bincount(o, minlength=n).min() > 0.8 * keys.size / n


def test_shard_range_covers_everything():
    for n, w in ((10, 3), (7, 8), (0, 4), (1000, 8)):
        spans = [kdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


# ---------------------------------------------------------------- k > 32: exchange_merge_wide --------------------------------

def exchange_merge_wide(shard, group=None, force=False):
    """exchange_merge for wide tables: every (k-mer, count) record goes to owner_of_w(k-mer); on return shard.table holds exactly
    the k-mers this rank owns, counts summed over ranks (exact integer sums: bit-identical to one process).  The shard is
    duck-typed (part_sizes / partition / rebuild) like the one-word exchange's."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return shard
    rank = dist.get_rank(group)
    dev = shard.device
    meta = torch.tensor([shard.k, int(shard.canonical)], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    if any(int(m[0]) != shard.k or int(m[1]) != int(shard.canonical) for m in metas):
        raise ValueError("exchange_merge_wide: ranks disagree on k / canonical")
    sizes = shard.part_sizes(world)                                          # records I hold for each owner
    s_all = [torch.empty(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(s_all, torch.from_numpy(sizes).to(dev), group=group)
    recv_from = np.array([int(s[rank]) for s in s_all], dtype=np.int64)      # what each peer holds for me
    send_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    recv_off = np.concatenate([[0], np.cumsum(recv_from)]).astype(np.int64)
    send = shard.partition(world, send_off[:-1].astype(np.uint64), int(send_off[-1]))
    recv = [torch.empty(max(int(recv_off[-1]), 1), dtype=torch.int64, device=dev) for _ in range(3)]
    ops = []
    for p in range(world):
        a, n_out = int(send_off[p]), int(sizes[p])
        b, n_in = int(recv_off[p]), int(recv_from[p])
        if p == rank:
            for r, s in zip(recv, send):
                r[b:b + n_in].copy_(s[a:a + n_out])
            continue
        for r, s in zip(recv, send):
            if n_out:
                ops.append(dist.P2POp(dist.isend, s[a:a + n_out], p, group))
            if n_in:
                ops.append(dist.P2POp(dist.irecv, r[b:b + n_in], p, group))
    for r in (dist.batch_isend_irecv(ops) if ops else []):
        r.wait()
    if dev.type == "cuda":
        torch.cuda.current_stream().synchronize()
    del send
    shard.rebuild(recv[0], recv[1], recv[2], int(recv_off[-1]))
    return shard


class OracleWideShard:
    """Same duck type as kat_amd.dist.HipWideShard (part_sizes / partition / rebuild), backed by the wide CPU oracle table."""

    def __init__(self, table):
        self.table = table
        self.k, self.canonical = table.k, table.canonical
        self.device = torch.device("cpu")

    def _records(self):
        hi, lo, c = self.table.dump_sorted()
        return hi, lo, c

    def part_sizes(self, n_parts):
        hi, lo, _ = self._records()
        return np.bincount(kdist.owner_of_wide(hi, lo, self.k, n_parts), minlength=n_parts).astype(np.int64)

    def partition(self, n_parts, offsets, total):
        hi, lo, c = self._records()
        order = np.argsort(kdist.owner_of_wide(hi, lo, self.k, n_parts), kind="stable")
        assert total == hi.size and list(offsets) == list(np.concatenate([[0], np.cumsum(self.part_sizes(n_parts))])[:-1])
        return [torch.from_numpy(x[order].view(np.int64).copy()) for x in (hi, lo, c)]

    def rebuild(self, hi, lo, counts, n):
        from oracle import koracle as ko
        t = ko.WideTable(self.k, self.canonical)
        h, l, c = (x[:n].numpy().view(np.uint64) for x in (hi, lo, counts))
        for a, b, v in zip(h, l, c):
            t.add((int(a) << 64) | int(b), int(v))
        self.table = t


KW = 45


def _wide_worker(rank, world, port, out_dir):
    from oracle import koracle as ko
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.genome(G, seed=11)
    lo, hi = kdist.shard_range(N_READS // 2, rank, world)
    reads = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=1)
    c_lo, c_hi = kdist.shard_range(G // CONTIG, rank, world)
    asm = synth.stream_of_contigs(g[c_lo * CONTIG:c_hi * CONTIG], CONTIG)
    t1 = ko.WideTable(KW, True).count_bases(reads)
    t2 = ko.WideTable(KW, False).count_bases(asm)
    t1.add((1 << 85) + 12345, (1 << 33) + rank)                                # a count above 32 bits travels like any other
    o1 = exchange_merge_wide(OracleWideShard(t1)).table
    o2 = exchange_merge_wide(OracleWideShard(t2)).table
    hi_, lo_, _ = o1.dump_sorted()
    assert (kdist.owner_of_wide(hi_, lo_, KW, world) == rank).all()
    mx, cc, sp = ko.comp(o1, o2, 1.0, 1.0, 101, 101)
    h, gm = o1.hist(1, 200, 1), o1.gcp(1.0, 100)
    mx, cc, sp, h, gm = kdist.allreduce_u64([mx, cc, sp, h, gm], torch.device("cpu"))
    if rank == 0:
        np.savez(os.path.join(out_dir, "wide.npz"), mx=mx, cc=cc, sp=sp, h=h, gm=gm)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_wide_sharded_exchange_matches_single_process(ko, tmp_path, world):
    mp.spawn(_wide_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "wide.npz")
    g = synth.genome(G, seed=11)
    t1 = ko.WideTable(KW, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
    t1.add((1 << 85) + 12345, world * (1 << 33) + world * (world - 1) // 2)
    t2 = ko.WideTable(KW, False).count_bases(synth.stream_of_contigs(g, CONTIG))
    mx, cc, sp = ko.comp(t1, t2, 1.0, 1.0, 101, 101)
    assert np.array_equal(got["mx"], mx) and np.array_equal(got["cc"], cc) and np.array_equal(got["sp"], sp)
    assert np.array_equal(got["h"], t1.hist(1, 200, 1)) and np.array_equal(got["gm"], t1.gcp(1.0, 100))
