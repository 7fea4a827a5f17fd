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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import kat_amd
    from kat_amd import dist as kdist
    from kat_amd import synth

    class StagedShard(kdist.HipShard):
        """HipShard whose exchange buffers are staged through host memory (gloo transport)."""

        def __init__(self, table):
            super().__init__(table)
            self.cuda = self.device
            self.device = torch.device("cpu")

        def new_like(self, size_hint, grid_of=None):
            t = self.table
            return StagedShard(t.engine.table(t.k, t.canonical, size_hint=max(int(size_hint), 1024),
                                              like=grid_of.table if grid_of is not None else None))

        exchange_buffers = None          # not a method: exchange_merge falls back to partition_into / empty_like

        def partition_into(self, n_parts, sizes, keys=None, counts=None):
            self.device = self.cuda
            k, c = super().partition_into(n_parts, sizes)
            self.device = torch.device("cpu")
            return k.cpu(), c.cpu()

        def empty_like(self, n):
            return torch.empty(max(n, 1), dtype=torch.int64), torch.empty(max(n, 1), dtype=torch.int64)

        def merge_from(self, keys, counts, n):
            if n:
                super().merge_from(keys[:n].contiguous().to(self.cuda), counts[:n].contiguous().to(self.cuda), n)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    eng = kat_amd.Engine(0)
    g = synth.genome(G, seed=11)
    lo, hi = kdist.shard_range(N_READS // 2, rank, world)
    reads = synth.reads(g, 2 * lo, 2 * (hi - lo), seed=1)
    c_lo, c_hi = kdist.shard_range(G // CONTIG, rank, world)
    asm = synth.stream_of_contigs(g[c_lo * CONTIG:c_hi * CONTIG], CONTIG)
    rb, ab = eng.alloc(reads.size), eng.alloc(asm.size)
    rb.upload(reads)
    ab.upload(asm)
    t1 = eng.table(K, True, size_hint=1 << 22).count_bases(rb)
    t2 = eng.table(K, True, size_hint=1 << 20, like=t1).count_bases(ab)
    o1 = kdist.exchange_merge(StagedShard(t1))
    o2 = kdist.exchange_merge(StagedShard(t2), grid_of=o1)
    keys, _ = o1.table.dump_sorted()
    assert (kdist.owner_of(keys, K, world) == rank).all()
    mx, cc, sp = kat_amd.comp(o1.table, o2.table, 1.0, 1.0, 201, 101)
    h, gm = o1.table.hist(1, 300, 1), o1.table.gcp(1.0, 100)
    mx, cc, sp, h, gm = kdist.allreduce_u64([mx, cc, sp, h, gm], torch.device("cpu"))
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mx=mx, cc=cc, sp=sp, h=h, gm=gm)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_ranks_one_gpu_match_single_process(engine, ko, tmp_path):
    from kat_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    g = synth.genome(G, seed=11)
    o1 = ko.Table(K, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
    o2 = ko.Table(K, True).count_bases(synth.stream_of_contigs(g, CONTIG))
    mx, cc, sp = ko.comp(o1, o2, 1.0, 1.0, 201, 101)
    assert np.array_equal(got["cc"], cc) and np.array_equal(got["mx"], mx) and np.array_equal(got["sp"], sp)
    assert np.array_equal(got["h"], o1.hist(1, 300, 1)) and np.array_equal(got["gm"], o1.gcp(1.0, 100))
