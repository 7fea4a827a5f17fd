"""The multi-GPU exchange as a MODEL in Python over torch.distributed, and the host mirrors of the device's owner functions.

The product's exchange is native: kat_amd/csrc/kg_comm.hip behind the C ABI (katgpu_comm_* / katgpu_exchange_merge /
katgpu_allreduce_u64; `kat_amd.Comm`, `katgpu <mode> --gpus N`, `bench.py --gpus N`).  What lives here is what the CPU-side tests of
that protocol need (tests/test_dist_gloo.py: world 2 / 3 / 4 over gloo, an oracle-backed stand-in for the table): the same
region-ordered, chunked, in-place protocol -- per-region counts all to all, the send list in chunks of consecutive regions with
chunk c on the wire while chunk c-1 is applied, out-of-band records for counts above 32 bits -- written against a duck-typed shard
(geometry / begin_exchange / exchange_buffers / extract / clear / merge_chunk / merge_big), plus shard_range (how reads and contigs
are dealt to ranks) and owner_of / owner_of_wide (kg_device.hpp: owner_of, owner_of_w), which the GPU tests check the device against.
"""
import numpy as np
import torch
import torch.distributed as dist

BIG_CAP = 4200


def _exchange_rows(rows_out, rows_in, rank, world, group):
    """rows_out[p] -> rank p; rows_in[s] <- rank s (grouped point-to-point)."""
    ops = [dist.P2POp(dist.isend, rows_out[p], p, group) for p in range(world) if p != rank and rows_out[p].numel()]
    ops += [dist.P2POp(dist.irecv, rows_in[p], p, group) for p in range(world) if p != rank and rows_in[p].numel()]
    for r in (dist.batch_isend_irecv(ops) if ops else []):
        r.wait()


def exchange_merge(shard, group=None, min_chunks=4):
    """Route every record of `shard` to its owner rank, IN PLACE: on return the same shard holds exactly the k-mers this rank owns,
    their counts summed over all ranks; the table keeps its storage and its region grid.  world_size == 1: nothing to do."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return shard
    rank = dist.get_rank(group)
    dev = shard.device

    # ---- geometry of every rank's table: which senders are ordered by MY regions ----
    geo = shard.geometry()
    g_all = [torch.empty(6, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(g_all, torch.from_numpy(geo).to(dev), group=group)
    geos = np.stack([g.cpu().numpy() for g in g_all])
    if not ((geos[:, 0] == geo[0]).all() and (geos[:, 1] == geo[1]).all()):
        raise ValueError("exchange_merge: ranks disagree on k / canonical")
    R_of = geos[:, 2]

    # ---- pass 1: how many records go where, per region ----
    sizes, cnt = shard.begin_exchange(world)                                  # cnt: int32 [world, R_mine]
    total_send = int(sizes.sum())
    rcnt = [cnt[rank] if s == rank else torch.empty(int(R_of[s]), dtype=torch.int32, device=dev) for s in range(world)]
    _exchange_rows([cnt[p] for p in range(world)], rcnt, rank, world, group)
    cnt_host = cnt.cpu().numpy().view(np.uint32).astype(np.int64)             # [world, R_mine]
    rcnt_host = [r.cpu().numpy().view(np.uint32).astype(np.int64) for r in rcnt]
    part_base = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cnt_cum = [np.concatenate([[0], np.cumsum(cnt_host[p])]) for p in range(world)]          # records of part p before region g
    rcnt_cum = [np.concatenate([[0], np.cumsum(rcnt_host[s_])]) for s_ in range(world)]

    # ---- chunks of consecutive regions (the product picks as few as its exchange scratch allows; the model takes min_chunks) ----
    C = max(1, min(int(min_chunks), int(R_of.min())))
    bounds = [(np.arange(C + 1, dtype=np.int64) * int(R_of[s])) // C for s in range(world)]                          # region boundaries per sender
    recv_sz = np.stack([np.diff(rcnt_cum[s][bounds[s]]) for s in range(world)]).astype(np.int64)                    # [sender, chunk]
    set_records = int(max(1, max(int(recv_sz[:, c].sum() - recv_sz[rank, c]) for c in range(C))))
    send_off = np.stack([part_base[p] + cnt_cum[p][bounds[rank]] for p in range(world)]).astype(np.int64)            # [owner, chunk boundary]

    # ---- pass 2: the send list; the emptied table becomes the owner table ----
    bufs = shard.exchange_buffers(total_send, set_records)
    big_keys, big_counts = shard.extract(world, bufs)
    shard.clear()
    skeys, scounts = bufs["send_keys"], bufs["send_counts"]

    def post(c):
        rk, rc = bufs["recv"][c % 2]
        ops, layout, o = [], [], 0
        for s in range(world):
            if s == rank:
                continue
            n_out = int(send_off[s][c + 1] - send_off[s][c])
            a, n_in = int(send_off[s][c]), int(recv_sz[s][c])
            ops += [dist.P2POp(dist.isend, b[a:a + n_out], s, group) for b in (skeys, scounts) if n_out]
            ops += [dist.P2POp(dist.irecv, b[o:o + n_in], s, group) for b in (rk, rc) if n_in]
            layout.append((s, o, n_in))
            o += n_in
        return (dist.batch_isend_irecv(ops) if ops else []), layout

    def merge(c, layout):
        rk, rc = bufs["recv"][c % 2]
        my_lo, my_hi = int(bounds[rank][c]), int(bounds[rank][c + 1])
        a, n_own = int(send_off[rank][c]), int(send_off[rank][c + 1] - send_off[rank][c])
        sources = [dict(keys=skeys[a:a + n_own], counts=scounts[a:a + n_own], rcnt=rcnt[rank][my_lo:my_hi], n=n_own,
                        p1=geo[4], p2=geo[5], own=True, offset=a)]
        for s, o, n_in in layout:
            same_regions = int(bounds[s][c]) == my_lo and int(bounds[s][c + 1]) == my_hi and geos[s][4] == geo[4] and geos[s][5] == geo[5]
            sources.append(dict(keys=rk[o:o + n_in], counts=rc[o:o + n_in], rcnt=rcnt[s][my_lo:my_hi] if same_regions else None, n=n_in,
                                p1=geos[s][4], p2=geos[s][5], own=False, offset=o))
        shard.merge_chunk(my_lo, my_hi, sources, c % 2)

    pending = None
    for c in range(C + 1):
        cur = post(c) if c < C else None                    # chunk c goes on the wire ...
        if pending is not None:                              # ... while chunk c-1 is applied
            reqs, layout = pending
            for r in reqs:
                r.wait()
            merge(c - 1, layout)
        pending = cur

    # ---- out-of-band records: counts above 32 bits and the all-ones k-mer (a handful) ----
    mine = torch.zeros(1 + 2 * BIG_CAP, dtype=torch.int64)
    mine[0] = len(big_keys)
    mine[1:1 + len(big_keys)] = torch.from_numpy(np.asarray(big_keys, np.uint64).view(np.int64).copy())
    mine[1 + BIG_CAP:1 + BIG_CAP + len(big_keys)] = torch.from_numpy(np.asarray(big_counts, np.uint64).view(np.int64).copy())
    everyone = [torch.empty_like(mine, device=dev) for _ in range(world)]
    dist.all_gather(everyone, mine.to(dev), group=group)
    k = int(geo[0])
    for t in everyone:
        t = t.cpu().numpy()
        n = int(t[0])
        bk, bc = t[1:1 + n].view(np.uint64), t[1 + BIG_CAP:1 + BIG_CAP + n].view(np.uint64)
        own = owner_of(bk, k, world) == rank if n else np.zeros(0, bool)
        shard.merge_big(bk[own], bc[own])
    return shard


def allreduce_u64(arrays, device, group=None):
    """Sum uint64 numpy result arrays over ranks (ThreadedSparseMatrix::mergeThreadedMatricies across GPUs)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return arrays
    flat = np.concatenate([np.ascontiguousarray(a, np.uint64).reshape(-1) for a in arrays]).view(np.int64)
    t = torch.from_numpy(flat.copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    parts = np.split(t.cpu().numpy().view(np.uint64), np.cumsum([int(np.prod(a.shape)) for a in arrays])[:-1])
    return [p.reshape(a.shape).copy() for p, a in zip(parts, arrays)]


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


# ---- k > 32 ("wide" tables): the k-mer is (hi, lo), the upper / lower 64 bits of its 2k-bit word ----

def _revcomp_wide(hi, lo, k):
    """Reverse complement of 2k-bit words given as (hi, lo) uint64 arrays, 33 <= k <= 64 (base-by-base, vectorised over records)."""
    hi = np.asarray(hi, dtype=np.uint64)
    lo = np.asarray(lo, dtype=np.uint64)
    rhi, rlo = np.zeros_like(hi), np.zeros_like(lo)
    for i in range(k):                                   # base i (from the LSB) -> complemented, to position k-1-i
        b = ((lo >> np.uint64(2 * i)) if i < 32 else (hi >> np.uint64(2 * (i - 32)))) & np.uint64(3)
        j = k - 1 - i
        c = np.uint64(3) - b
        if j < 32:
            rlo |= c << np.uint64(2 * j)
        else:
            rhi |= c << np.uint64(2 * (j - 32))
    return rhi, rlo


def owner_of_wide(hi, lo, k, n_parts):
    """Host mirror of kg_device.hpp: owner_of_w -- a second mix of the canonical form's table hash."""
    from .synth import mulhi64
    hi = np.asarray(hi, dtype=np.uint64)
    lo = np.asarray(lo, dtype=np.uint64)
    rhi, rlo = _revcomp_wide(hi, lo, k)
    rc_less = (rhi < hi) | ((rhi == hi) & (rlo < lo))
    chi, clo = np.where(rc_less, rhi, hi), np.where(rc_less, rlo, lo)
    a = (chi << np.uint64(1)) | (clo >> np.uint64(63))                     # the two 63-bit halves the table stores
    b = clo & np.uint64(0x7FFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _mix64(b ^ (a * np.uint64(0x9E3779B97F4A7C15)))
    return mulhi64(_mix64(h ^ np.uint64(0x9E3779B97F4A7C15)), np.uint64(n_parts)).astype(np.int64)
