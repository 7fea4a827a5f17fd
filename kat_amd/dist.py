"""One process per GPU: read sharding + owner-partitioned merge of the per-GPU partial tables over torch.distributed.

KAT has no distributed path (one process, std::thread); this is the exchange step BASELINE.json's north_star asks
for.  Each rank counts its own shard of the input into a LOCAL partial table.  Every (k-mer, count) record is then
routed to owner(k-mer) (a hash of the canonical form, kg_device.hpp: owner_of) with grouped point-to-point sends --
on RCCL one ncclGroupStart/End of ncclSend/ncclRecv pairs that drives all xGMI links of the fully connected node at
once, which suits xGMI better than a ring all-reduce would -- and the owner adds the counts (exact integer sums, so
the result is bit-identical to a single-GPU run).  Reducers then run on the owned shards and their small outputs
(80 KB hist / 216 KB gcp / 8 MB comp matrix + counters) are summed with one all-reduce.

The exchange is REGION-ORDERED and IN PLACE (include/katgpu.h, "region-ordered exchange"):
  * ranks count into tables of the same region grid, so a k-mer sits in the same region index everywhere;
  * the sender extracts its table once into a send list (8-byte key + 4-byte count per record, grouped by owner and,
    inside an owner, ordered by region), then EMPTIES the table: the emptied table is the owner table, so the
    exchange allocates nothing next to the tables (the send list and the receive buffers live in katgpu's arena);
  * the list travels in C chunks of consecutive regions, double buffered: while chunk c is on the wire, the owner
    applies chunk c-1 -- the runs of each region from every rank -- to that region in LDS (k_merge_apply): no global
    atomic per record, no re-partitioning on the receiving side, one sweep of the owner table in total;
  * a sender whose grid differs (its table regrew differently) is still exact: its records go through the direct path.

The table object is duck-typed (geometry / begin_exchange / exchange_buffers / extract / clear / merge_chunk /
merge_big) so that the CPU gloo tests drive the same code with an oracle-backed stand-in; the product adapter is
HipShard.
"""
import numpy as np
import torch
import torch.distributed as dist

BIG_CAP = 4200


def _align(n, a=256):
    return (int(n) + a - 1) // a * a


class HipShard:
    """Adapter over a kat_amd.Table.  Exchange tensors are torch views of katgpu's arena (plumbing only); with
    staged=True they are host tensors instead (gloo transport: two ranks on one GPU in the tests)."""

    def __init__(self, table, staged=False):
        self.table = table
        self.cuda = torch.device("cuda", torch.cuda.current_device())
        self.staged = staged
        self.device = torch.device("cpu") if staged else self.cuda
        self._cnt_buf = None

    def geometry(self):
        g = self.table.geometry()
        return np.array([g.k, g.canonical, g.n_regions, g.region_slots, g.p1, g.p2], dtype=np.int64)

    def begin_exchange(self, n_parts):
        """Pass 1 of the extraction: (records per owner, int32 tensor [n_parts, n_regions] of records per owner and region)."""
        eng = self.table.engine
        R = int(self.table.geometry().n_regions)
        self._cnt_buf = eng.alloc(4 * n_parts * R + 64)                     # small (27 MB at 8 x 860 K regions): outside the arena
        sizes = self.table.extract_sizes(n_parts, self._cnt_buf.ptr).astype(np.int64)
        from .binding import ScratchView
        cnt = torch.as_tensor(ScratchView(self._cnt_buf.ptr, 4 * n_parts * R), device=self.cuda).view(torch.int32).view(n_parts, R)
        self._cnt_dev = cnt
        return sizes, (cnt.cpu() if self.staged else cnt)

    def exchange_capacity(self, want_bytes):
        """Bytes of exchange scratch available (the arena, grown to want_bytes when it is smaller and the device has room)."""
        eng = self.table.engine
        cap = eng.scratch(0).capacity
        if cap < want_bytes:
            try:
                cap = eng.scratch(want_bytes).capacity          # keeps the old arena when the larger one cannot be had
            except Exception:
                cap = eng.scratch(0).capacity
        return cap

    @staticmethod
    def exchange_bytes(total_send, set_records):
        return _align(8 * max(total_send, 1)) + _align(4 * max(total_send, 1)) + 2 * (_align(8 * max(set_records, 1)) + _align(4 * max(set_records, 1))) + 256

    def exchange_buffers(self, total_send, set_records):
        """send keys / counts (int64 / int32, total_send records) and two receive sets of set_records records."""
        eng = self.table.engine
        nbytes = self.exchange_bytes(total_send, set_records)
        raw = torch.as_tensor(eng.scratch(nbytes), device=self.cuda)
        o = 0

        def take(n, width, dtype):
            nonlocal o
            t = raw[o:o + width * max(n, 1)].view(dtype)
            o += _align(width * max(n, 1))
            return t
        b = {"send_keys": take(total_send, 8, torch.int64), "send_counts": take(total_send, 4, torch.int32)}
        dev_sets = [(take(set_records, 8, torch.int64), take(set_records, 4, torch.int32)) for _ in range(2)]
        if self.staged:
            b["dev_sets"] = dev_sets
            b["recv"] = [(torch.empty(max(set_records, 1), dtype=torch.int64), torch.empty(max(set_records, 1), dtype=torch.int32)) for _ in range(2)]
        else:
            b["recv"] = dev_sets
        self._bufs = b
        return b

    def extract(self, n_parts, bufs):
        """Pass 2: fill the send list; returns the out-of-band records (counts above 32 bits, the all-ones k-mer)."""
        big = self.table.extract(n_parts, self._cnt_buf.ptr, bufs["send_keys"].data_ptr(), bufs["send_counts"].data_ptr())
        if self.staged:
            bufs["send_keys_dev"], bufs["send_counts_dev"] = bufs["send_keys"], bufs["send_counts"]
            bufs["send_keys"], bufs["send_counts"] = bufs["send_keys"].cpu(), bufs["send_counts"].cpu()
        return big

    def clear(self):
        self.table.clear()

    def wait_transport(self):
        """Host-side wait for what the transport has delivered so far (NOT a device-wide sync: the next chunk stays in flight)."""
        if not self.staged:
            torch.cuda.current_stream().synchronize()

    def merge_chunk(self, g_lo, g_hi, sources, set_index):
        """sources: dicts with keys / counts (tensor slices), rcnt (int32 slice of the sender's region counts for [g_lo, g_hi), or
        None), n, p1, p2, own (the slice lives in the send list)."""
        src = []
        if self.staged:                                                     # host tensors -> the device-side sets
            dk, dc = self._bufs["dev_sets"][set_index]
            o = 0
            for s in sources:
                n = int(s["n"])
                if s["own"]:
                    k_ptr = self._bufs["send_keys_dev"].data_ptr() + 8 * int(s["offset"])
                    c_ptr = self._bufs["send_counts_dev"].data_ptr() + 4 * int(s["offset"])
                else:
                    dk[o:o + n].copy_(s["keys"][:n])
                    dc[o:o + n].copy_(s["counts"][:n])
                    k_ptr, c_ptr = dk[o:o + n].data_ptr() if n else 0, dc[o:o + n].data_ptr() if n else 0
                    o += n
                r = s["rcnt"]
                if r is not None:
                    r = r.to(self.cuda) if r.device.type == "cpu" else r
                    s["_keep"] = r
                src.append((k_ptr, c_ptr, r.data_ptr() if r is not None and r.numel() else None, n, int(s["p1"]), int(s["p2"])))
            torch.cuda.synchronize()
        else:
            for s in sources:
                n = int(s["n"])
                r = s["rcnt"]
                src.append((s["keys"].data_ptr() if n else 0, s["counts"].data_ptr() if n else 0,
                            r.data_ptr() if r is not None and r.numel() else None, n, int(s["p1"]), int(s["p2"])))
        src = [x for x in src if x[3]]
        if src:
            self.table.merge_regions(int(g_lo), int(g_hi), src)

    def merge_big(self, keys, counts):
        if len(keys):
            self.table.merge_host(keys, counts)

    def end_exchange(self):
        if self._cnt_buf is not None:
            self.table.engine.sync()
            self._cnt_buf.free()
            self._cnt_buf = None
        self._bufs = None

    def free(self):
        self.table.free()


def _exchange_rows(rows_out, rows_in, rank, world, group):
    """rows_out[p] -> rank p; rows_in[s] <- rank s (grouped point-to-point)."""
    ops = []
    for p in range(world):
        if p == rank:
            continue
        if rows_out[p].numel():
            ops.append(dist.P2POp(dist.isend, rows_out[p], p, group))
        if rows_in[p].numel():
            ops.append(dist.P2POp(dist.irecv, rows_in[p], p, group))
    for r in (dist.batch_isend_irecv(ops) if ops else []):
        r.wait()


def exchange_merge(shard, group=None, min_chunks=4, force=False):
    """Route every record of `shard` to its owner rank, IN PLACE: on return the same shard holds exactly the k-mers this
    rank owns, with their counts summed over all ranks.  The table keeps its storage and its region grid (a second table created
    "like" the first still joins with it region by region).

    world_size == 1: the local table already is the owner table (force=True runs the protocol all the same -- extraction,
    clear, region-by-region merge of the rank's own send list, the collectives -- which is how a single-GPU box exercises it).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
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
    s_all = [torch.empty(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(s_all, torch.from_numpy(sizes).to(dev), group=group)
    recv_from = np.array([int(s[rank]) for s in s_all], dtype=np.int64)       # what each peer holds for me
    rcnt = [cnt[rank] if s == rank else torch.empty(int(R_of[s]), dtype=torch.int32, device=dev) for s in range(world)]
    _exchange_rows([cnt[p] for p in range(world)], rcnt, rank, world, group)
    cnt_host = cnt.cpu().numpy().view(np.uint32).astype(np.int64)             # [world, R_mine]
    rcnt_host = [r.cpu().numpy().view(np.uint32).astype(np.int64) for r in rcnt]
    part_base = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cnt_cum = [np.concatenate([[0], np.cumsum(cnt_host[p])]) for p in range(world)]          # records of part p before region g
    rcnt_cum = [np.concatenate([[0], np.cumsum(rcnt_host[s_])]) for s_ in range(world)]

    # ---- chunks of consecutive regions, as few as the exchange scratch allows (>= min_chunks for the overlap) ----
    recv_other = int(recv_from.sum() - recv_from[rank])
    C = max(1, min(int(min_chunks), int(R_of.min())))
    cap = shard.exchange_capacity(shard.exchange_bytes(total_send, -(-recv_other // max(C, 1)) * 5 // 4)) if hasattr(shard, "exchange_capacity") else None
    while True:
        bounds = [(np.arange(C + 1, dtype=np.int64) * int(R_of[s])) // C for s in range(world)]       # region boundaries per sender
        recv_sz = np.stack([np.diff(rcnt_cum[s][bounds[s]]) for s in range(world)]).astype(np.int64)                # [sender, chunk]
        set_records = int(max(1, max(int(recv_sz[:, c].sum() - recv_sz[rank, c]) for c in range(C))))
        fits = 1 if cap is None or shard.exchange_bytes(total_send, set_records) <= cap else 0
        f = torch.tensor([fits], dtype=torch.int64, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        if int(f.item()) or C >= int(R_of.min()):
            if not int(f.item()):
                raise MemoryError("exchange_merge: the send list and one region's receive buffers do not fit the exchange scratch")
            break
        C = min(C * 2, int(R_of.min()))
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
            if n_out:
                a = int(send_off[s][c])
                ops.append(dist.P2POp(dist.isend, skeys[a:a + n_out], s, group))
                ops.append(dist.P2POp(dist.isend, scounts[a:a + n_out], s, group))
            n_in = int(recv_sz[s][c])
            if n_in:
                ops.append(dist.P2POp(dist.irecv, rk[o:o + n_in], s, group))
                ops.append(dist.P2POp(dist.irecv, rc[o:o + n_in], s, group))
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
            if hasattr(shard, "wait_transport"):
                shard.wait_transport()
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
    if hasattr(shard, "end_exchange"):
        shard.end_exchange()
    return shard


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


# ---- k > 32 ("wide" tables): the k-mer is (hi, lo), the upper / lower 64 bits of its 2k-bit word -----------------------

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


class HipWideShard:
    """Adapter over a wide kat_amd.Table (k > 32) for exchange_merge_wide.  Record buffers are plain torch tensors (device, or host
    with staged=True for the gloo tests): this exchange is the simple one -- partition by owner, all-to-all, rebuild -- not the
    region-ordered in-place protocol of the one-word tables, whose LDS merge is built around 12-byte slots."""

    def __init__(self, table, staged=False):
        assert table.k > 32
        self.table = table
        self.k = table.k
        self.canonical = table.canonical
        self.staged = staged
        self.cuda = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device("cpu") if staged else self.cuda

    def part_sizes(self, n_parts):
        return self.table.partition_sizes(n_parts).astype(np.int64)

    def partition(self, n_parts, offsets, total):
        """(hi, lo, counts) int64 tensors of `total` records on self.device, part p starting at offsets[p]."""
        dev = [torch.empty(max(total, 1), dtype=torch.int64, device=self.cuda) for _ in range(3)]
        self.table.partition_wide(n_parts, offsets, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr())
        return [t.cpu() for t in dev] if self.staged else dev

    def rebuild(self, hi, lo, counts, n):
        """Replace the table by one holding exactly these n records (equal k-mers summed)."""
        eng = self.table.engine
        old = self.table
        new = eng.table(self.k, self.canonical, size_hint=max(int(n / 0.6) + 1024, 1 << 16))
        old.free()
        if n:
            dev = [t[:n].to(self.cuda).contiguous() for t in (hi, lo, counts)]
            new.merge_device_wide(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), n)
            eng.sync()
        self.table = new

    def free(self):
        self.table.free()


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
