"""One process per GPU: read sharding + owner-partitioned merge of the per-GPU partial tables over torch.distributed.

KAT has no distributed path (one process, std::thread); this is the exchange step BASELINE.json's north_star asks
for.  Each rank counts its own shard of the input into a LOCAL partial table.  Table layouts differ per GPU, so
the merge is keyed, not element-wise: every (k-mer, count) record is routed to owner(k-mer) (a hash of the
canonical form, kg_device.hpp: owner_of) with grouped point-to-point sends -- on RCCL that is one
ncclGroupStart/End of ncclSend/ncclRecv pairs that drives all xGMI links of the fully connected node at once,
which suits xGMI better than a ring all-reduce would -- and the owner adds the counts (exact integer sums, so the
result is bit-identical to a single-GPU run).  Reducers then run on the owned shards and their small outputs
(80 KB hist / 216 KB gcp / 8 MB comp matrix + counters) are summed with one all-reduce.

The table object is duck-typed (`partition_sizes`, `partition_into`, `merge_from`, `new_like`) so that the CPU
gloo tests can drive the same code with an oracle-backed stand-in; the product adapter is HipShard.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipShard:
    """Adapter over a kat_amd.Table whose exchange buffers are torch CUDA tensors (plumbing only)."""

    def __init__(self, table):
        self.table = table
        self.device = torch.device("cuda", torch.cuda.current_device())

    def new_like(self, size_hint, grid_of=None):
        t = self.table
        return HipShard(t.engine.table(t.k, t.canonical, size_hint=max(int(size_hint), 1024), like=grid_of.table if grid_of is not None else None))

    def partition_sizes(self, n_parts):
        return self.table.partition_sizes(n_parts).astype(np.int64)

    def exchange_buffers(self, n_send, n_recv):
        """Four int64 record arrays (send keys/counts, receive keys/counts).  They are views into katgpu's own arena when
        torch can wrap it (after a large count the arena holds most of the free HBM, so allocating next to it would fail);
        otherwise the arena is released and torch allocates."""
        n_send, n_recv = max(int(n_send), 1), max(int(n_recv), 1)
        eng = self.table.engine
        try:
            raw = torch.as_tensor(eng.scratch(16 * (n_send + n_recv) + 64), device=self.device).view(torch.int64)
            return raw[:n_send], raw[n_send:2 * n_send], raw[2 * n_send:2 * n_send + n_recv], raw[2 * n_send + n_recv:2 * (n_send + n_recv)]
        except Exception:
            eng.release_scratch()
            mk = lambda n: torch.empty(n, dtype=torch.int64, device=self.device)
            return mk(n_send), mk(n_send), mk(n_recv), mk(n_recv)

    def partition_into(self, n_parts, sizes, keys=None, counts=None):
        total = int(sizes.sum())
        if keys is None:
            keys = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
            counts = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        torch.cuda.synchronize()
        self.table.partition(n_parts, offsets, keys.data_ptr(), counts.data_ptr())
        return keys, counts

    def merge_from(self, keys, counts, n):
        if n:
            torch.cuda.synchronize()                 # the exchange ran on torch's stream, the merge runs on katgpu's
            self.table.merge_device(keys.data_ptr(), counts.data_ptr(), int(n))

    def empty_like(self, n):
        return (torch.empty(max(n, 1), dtype=torch.int64, device=self.device),
                torch.empty(max(n, 1), dtype=torch.int64, device=self.device))

    def free(self):
        self.table.free()


def exchange_merge(shard, group=None, load=0.6, grid_of=None):
    """Route every record of `shard` to its owner rank; returns the owner shard (same duck type).
    grid_of: an owner shard whose region grid the new owner table should adopt (comp then joins region against region).

    world_size == 1: the local table already is the owner table, returned as is.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return shard
    rank = dist.get_rank(group)
    sizes = shard.partition_sizes(world)                                   # records this rank holds for each owner
    dev = shard.device
    mine = torch.from_numpy(sizes).to(dev)
    all_sizes = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_sizes, mine, group=group)
    recv_sizes = np.array([int(s[rank]) for s in all_sizes], dtype=np.int64)   # what each peer sends to me
    send_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    recv_off = np.concatenate([[0], np.cumsum(recv_sizes)]).astype(np.int64)
    if callable(getattr(shard, "exchange_buffers", None)):
        keys, counts, rkeys, rcounts = shard.exchange_buffers(int(send_off[-1]), int(recv_off[-1]))
        keys, counts = shard.partition_into(world, sizes, keys, counts)
    else:
        keys, counts = shard.partition_into(world, sizes)
        rkeys, rcounts = shard.empty_like(int(recv_off[-1]))
    ops = []
    for p in range(world):
        if p == rank:
            continue
        if sizes[p]:
            ops.append(dist.P2POp(dist.isend, keys[send_off[p]:send_off[p + 1]], p, group))
            ops.append(dist.P2POp(dist.isend, counts[send_off[p]:send_off[p + 1]], p, group))
        if recv_sizes[p]:
            ops.append(dist.P2POp(dist.irecv, rkeys[recv_off[p]:recv_off[p + 1]], p, group))
            ops.append(dist.P2POp(dist.irecv, rcounts[recv_off[p]:recv_off[p + 1]], p, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    # size the owner table from what is about to land in it (an upper bound on its distinct count)
    owner = shard.new_like(int((int(recv_off[-1])) / load) + 1024) if grid_of is None else shard.new_like(int((int(recv_off[-1])) / load) + 1024, grid_of)
    s0, s1 = int(send_off[rank]), int(send_off[rank + 1])
    owner.merge_from(keys[s0:s1], counts[s0:s1], s1 - s0)                   # my own part needs no wire
    for r in reqs:
        r.wait()
    for p in range(world):
        if p != rank and recv_sizes[p]:
            r0, r1 = int(recv_off[p]), int(recv_off[p + 1])
            owner.merge_from(rkeys[r0:r1], rcounts[r0:r1], r1 - r0)
    return owner


def allreduce_u64(arrays, device, group=None):
    """Sum uint64 numpy result arrays over ranks (ThreadedSparseMatrix::mergeThreadedMatricies across GPUs)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return arrays
    flat = np.concatenate([np.ascontiguousarray(a, np.uint64).reshape(-1) for a in arrays]).view(np.int64)
    t = torch.from_numpy(flat.copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    res = t.cpu().numpy().view(np.uint64)
    out, o = [], 0
    for a in arrays:
        n = int(np.prod(a.shape))
        out.append(res[o:o + n].reshape(a.shape).copy())
        o += n
    return out


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of n_items for this rank (read pairs / contigs are independent units)."""
    per, extra = divmod(n_items, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


# ---- host mirror of the device's owner function (kg_device.hpp: mix64 / owner_of), for tests and host-side routing ----
def _mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33))
        x = x * np.uint64(0xff51afd7ed558ccd)
        x = x ^ (x >> np.uint64(33))
        x = x * np.uint64(0xc4ceb9fe1a85ec53)
        x = x ^ (x >> np.uint64(33))
    return x


def _revcomp(keys, k):
    x = np.asarray(keys, dtype=np.uint64)
    out = np.zeros_like(x)
    for i in range(k):                                   # base i (from the LSB) -> complemented, mirrored
        b = (x >> np.uint64(2 * i)) & np.uint64(3)
        out |= (np.uint64(3) - b) << np.uint64(2 * (k - 1 - i))
    return out


def owner_of(keys, k, n_parts):
    """Part index of each packed k-mer: a hash of its CANONICAL form, so both strands and both comp inputs co-locate."""
    from .synth import mulhi64
    x = np.asarray(keys, dtype=np.uint64)
    c = np.minimum(x, _revcomp(x, k))
    return mulhi64(_mix64(c ^ np.uint64(0x9E3779B97F4A7C15)), np.uint64(n_parts)).astype(np.int64)
