"""world_size-2 gloo run of the multi-GPU path (kat_amd/dist.py) on CPU.

The exchange / merge / all-reduce plumbing is the product's; the per-rank table is an oracle-backed stand-in (the HIP
table needs a GPU), so what is covered here is: read sharding, owner routing, grouped send/recv, result reduction --
and that the sharded answer is bit-identical to the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kat_amd import dist as kdist
from kat_amd import synth


class OracleShard:
    """Same duck type as kat_amd.dist.HipShard, backed by the CPU oracle table."""

    def __init__(self, table):
        self.table = table
        self.device = torch.device("cpu")

    def new_like(self, size_hint, grid_of=None):
        from oracle import koracle as ko
        return OracleShard(ko.Table(self.table.k, self.table.canonical))

    def _records_by_part(self, n_parts):
        keys, counts = self.table.dump_sorted()
        part = kdist.owner_of(keys, self.table.k, n_parts)
        order = np.argsort(part, kind="stable")
        return keys[order], counts[order], np.bincount(part, minlength=n_parts).astype(np.int64)

    def partition_sizes(self, n_parts):
        return self._records_by_part(n_parts)[2]

    def partition_into(self, n_parts, sizes):
        keys, counts, s = self._records_by_part(n_parts)
        assert np.array_equal(s, sizes)
        return torch.from_numpy(keys.view(np.int64).copy()), torch.from_numpy(counts.view(np.int64).copy())

    def merge_from(self, keys, counts, n):
        kk, cc = keys.numpy().view(np.uint64), counts.numpy().view(np.uint64)
        for i in range(int(n)):
            self.table.add(int(kk[i]), int(cc[i]))

    def empty_like(self, n):
        return torch.empty(max(n, 1), dtype=torch.int64), torch.empty(max(n, 1), dtype=torch.int64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


K, G, N_READS, CONTIG = 21, 40000, 3000, 5000


def _worker(rank, world, port, out_dir):
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
    o1 = kdist.exchange_merge(OracleShard(t1)).table
    o2 = kdist.exchange_merge(OracleShard(t2)).table
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


def test_two_rank_exchange_matches_single_process(ko, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    g = synth.genome(G, seed=11)
    t1 = ko.Table(K, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
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
