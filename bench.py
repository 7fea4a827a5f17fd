#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: k-mers/s for `kat comp` reads-vs-assembly, k=27 (configs[3]) on MI355X.

A step = one pass of the hot path over one synthetic batch already resident in HBM:
    allocate tables -> count reads (K1) -> count assembly (K1) -> [N>1: owner-partitioned exchange + merge] ->
    comp join/reduce (K5) -> D2H of the 8 MB matrix + counters.
value = (valid k-mer instances of all inputs on all ranks) / (max-over-ranks wall time).  Host file parsing and
PCIe are outside the timed region by contract (inputs resident); DESIGN.md quotes the PCIe-inclusive figure.

  python bench.py                                  # N=1, full config: 300 M x 150 bp PE reads vs 1 Gbp assembly
  python bench.py --reads 20000000 --genome 100000000          # scaled-down look
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N     # weak scaling: reads per GPU fixed
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=300_000_000, help="150 bp reads PER GPU (PE: reads/2 pairs)")
    ap.add_argument("--genome", type=int, default=1_000_000_000, help="genome / assembly length in bp (shared by all ranks)")
    ap.add_argument("--contig", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-ppm", type=int, default=2000, help="substitution errors per million bases (0.2 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="sync + print per-phase wall time (diagnostic; perturbs the timing)")
    ap.add_argument("--cpu-sample-reads", type=int, default=20_000_000)
    ap.add_argument("--hint-scale", type=float, default=1.0, help="diagnostic: scale the tables' size hints (e.g. 0.02: grown on the way)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: diagnostic only -- ranks may share one GPU, records are staged through host memory")
    return ap.parse_args()


def expected_distinct(instances, genome, k, err_ppm):
    """Upper-ish estimate used to pre-size tables (KAT users pass -H; the reference numbers in BASELINE.md were
    taken with a pre-sized hash too): genomic k-mers + one new k-mer per erroneous window."""
    p_err = 1.0 - (1.0 - err_ppm / 1e6) ** k
    return int(min(instances, genome + instances * p_err * 1.05))


def pmc_traffic(a, world):
    """HBM bytes per launch (= per partition round) of the count stage, from the committed rocprofv3 PMC passes of THIS command
    (tools/profile_bench.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs with --kernel-trace; units KB; FETCH_SIZE doubled, as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).  The workload is seeded, so the traffic of a round is
    reproducible; counters cannot be collected from inside the timed run.  None when the workload is not the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_fetch_write.json")
    if world != 1 or (a.reads, a.genome, a.contig, a.k, a.read_len, a.err_ppm) != (300_000_000, 1_000_000_000, 1_000_000, 27, 150, 2000):
        return None, None
    try:
        prof = json.load(open(path))
    except Exception:
        return None, None
    stage = ("k_p1v2_count", "k_p1_scan", "k_p1v2_scatter", "k_p1v2_scatter_chunked", "k_p2", "k_p2_fast", "k_p3_apply", "k_insert_keys")
    kb, rounds = 0.0, 0
    for name, e in prof.items():
        base = name.replace("kg::", "").split("<")[0]
        if base in stage:
            kb += 2.0 * e.get("FETCH_SIZE_KB_total", 0.0) + e.get("WRITE_SIZE_KB_total", 0.0)
            if base == "k_p3_apply":
                rounds += e.get("launches", 0)
    if not rounds:
        return None, None
    return int(kb * 1024 / rounds), "profiles/r01_final_pmc_fetch_write.json: (2 x FETCH_SIZE + WRITE_SIZE) of the count-stage kernels / %d rounds" % rounds


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world

    if world > 1:
        # the exchange keeps its send list and receive buffers inside the arena and allocates nothing else; leave room for the
        # second table, RCCL's channel buffers and the small per-region count matrices (must be set before the library loads)
        os.environ.setdefault("KATGPU_ARENA_FRACTION", "0.75")
    import torch
    import kat_amd
    from kat_amd import dist as kdist

    staged = a.dist_backend == "gloo"
    if staged:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    eng = kat_amd.Engine(local_rank)
    dev = torch.device("cpu") if staged else torch.device("cuda", local_rank)      # where collective tensors live

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    k, L = a.k, a.read_len
    n_reads = a.reads - (a.reads % 2)
    # ---- synthetic inputs, generated in HBM (not timed) ----
    g = eng.synth_genome(a.genome, seed=20260927)                               # genome the reads are sampled from
    reads = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350,
                            err_ppm=a.err_ppm, seed=1)
    # assembly = that genome cut into contigs; contigs are sharded over ranks
    n_contigs = (a.genome + a.contig - 1) // a.contig
    c_lo, c_hi = kdist.shard_range(n_contigs, rank, world)
    asm_full = eng.synth_genome(a.genome, seed=20260927, contig_len=a.contig)
    asm_ptr = asm_full.ptr + c_lo * (a.contig + 1)
    asm_bytes = min(asm_full.nbytes, c_hi * (a.contig + 1)) - c_lo * (a.contig + 1)
    g.free()
    eng.sync()

    inst_reads = n_reads * (L - k + 1)
    asm_bases_local = min(a.genome, c_hi * a.contig) - c_lo * a.contig
    inst_asm = max(0, asm_bases_local - (c_hi - c_lo) * (k - 1))
    inst_asm_total = max(0, a.genome - n_contigs * (k - 1))
    hint1 = int(expected_distinct(inst_reads, a.genome, k, a.err_ppm) / 0.62) + (1 << 20)
    hint2 = int(asm_bases_local / 0.62) + (1 << 20)
    if a.hint_scale != 1.0:
        hint1, hint2 = max(1024, int(hint1 * a.hint_scale)), max(1024, int(hint2 * a.hint_scale))

    results = {}

    phases = {}

    def mark(name, t_prev):
        if a.phases:
            eng.sync()
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (now - t_prev)
            return now
        return t_prev

    def step():
        tp = time.perf_counter()
        t1 = eng.table(k, True, size_hint=hint1)
        tp = mark("alloc1", tp)
        t1.count_bases_device(reads.ptr, reads.nbytes)
        tp = mark("count_reads", tp)
        t2 = eng.table(k, True, size_hint=hint2, like=t1)
        t2.count_bases_device(asm_ptr, asm_bytes)
        tp = mark("alloc2+count_asm", tp)
        if world > 1:
            # in place: each table is extracted into a region-ordered send list, emptied, and refilled with the k-mers this
            # rank owns (kat_amd/dist.py); t2 keeps t1's region grid, so comp still joins region against region
            results["distinct1_local"] = t1.stats(want_total=False)["distinct"]
            if k > 32:                                      # wide tables: owner partition -> all-to-all -> rebuild (not in place)
                t1 = kdist.exchange_merge_wide(kdist.HipWideShard(t1, staged=staged)).table
                t2 = kdist.exchange_merge_wide(kdist.HipWideShard(t2, staged=staged)).table
            else:
                kdist.exchange_merge(kdist.HipShard(t1, staged=staged))
                kdist.exchange_merge(kdist.HipShard(t2, staged=staged))
            tp = mark("exchange", tp)
        mx, cc, sp = kat_amd.comp(t1, t2)
        tp = mark("comp", tp)
        if world > 1:
            mx, cc, sp = kdist.allreduce_u64([mx, cc, sp], dev)
        results["mx"], results["cc"], results["sp"] = mx, cc, sp
        results["distinct1"] = t1.stats(want_total=False)["distinct"]
        t1.free()
        t2.free()
        tp = mark("free", tp)

    for _ in range(a.warmup):
        step()
    barrier()
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile()

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        d1 = torch.tensor([results["distinct1"]], dtype=torch.int64, device=dev)
        dist.all_reduce(d1)
        results["distinct1"] = int(d1.item())

    total_instances = world * inst_reads + inst_asm_total
    value = total_instances * a.steps / dt

    # ---- sanity: the counters must account for every instance (cheap size-independent parity property) ----
    cc = results["cc"]
    ok = int(cc[0]) == world * inst_reads and int(cc[1]) == inst_asm_total

    if a.phases and rank == 0:
        print("phases (s, summed over steps):", {n: round(v, 3) for n, v in phases.items()}, file=sys.stderr)
    if rank == 0:
        # ---- roofline of the count stage, from HIP events recorded on katgpu's own stream ----
        # algorithmic bytes: per instance L/(L-k+1) B of ASCII + 8 B key read + 4 B count read + 4 B count
        # write, plus 8 B key write per distinct k-mer; summed over this rank's count work of the timed steps.
        per_inst = L / (L - k + 1) + 16.0 + (8.0 if k > 32 else 0.0)               # k > 32: a second key word per slot
        d1_local = results.get("distinct1_local", results["distinct1"])          # what THIS rank's count stage wrote
        alg_bytes_step = per_inst * (inst_reads + inst_asm) + (16.0 if k > 32 else 8.0) * (d1_local + max(inst_asm, 0))
        stage = ["part_l1_count", "part_l1_scatter", "part_l2", "part_apply"]
        part_ms = sum(prof[n]["ms"] for n in stage)
        direct_ms = prof["count"]["ms"]
        if part_ms >= direct_ms:
            # partitioned counter: one "launch" = one round = the four stage kernels over the round's k-mers
            rounds = max(1, prof["part_apply"]["launches"])
            stage_ms = part_ms + direct_ms
            name = "count stage (partitioned): k_p1_count+k_p1_scan, k_p1_scatter, k_p2, k_p3_apply per round"
            per_kernel = {n: {"launches": prof[n]["launches"], "avg_ms": round(prof[n]["ms"] / max(1, prof[n]["launches"]), 3)} for n in stage}
        else:
            rounds = max(1, prof["count"]["launches"])
            stage_ms = direct_ms
            name = "k_count"
            per_kernel = {"count": {"launches": prof["count"]["launches"], "avg_ms": round(direct_ms / rounds, 3)}}
        achieved = alg_bytes_step * a.steps / (stage_ms / 1e3) / 1e9 if stage_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(a, world)
        roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "launches": rounds, "avg_launch_ms": round(stage_ms / rounds, 3),
                "alg_bytes_per_launch": int(alg_bytes_step * a.steps / rounds), "per_kernel": per_kernel}
        kernels_ms = {n: round(v["ms"] / a.steps, 3) for n, v in prof.items() if v["launches"]}
        cpu = None
        if not a.no_cpu_baseline and world == 1 and k <= 32:      # the host-core baseline is an N = 1 figure (the multi-threaded oracle port is one-word)
            cpu = cpu_baseline(eng, a, k, L)
        line = {
            "metric": "k-mers/sec (whole node) for kat comp k=%d, reads vs assembly" % k,
            "value": round(value, 1), "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "kat comp reads-vs-assembly: %d x %d bp PE reads per GPU (0.2%% subst. errors) vs %d bp assembly in %d bp contigs, k=%d, canonical"
                                   % (n_reads, L, a.genome, a.contig, k),
                       "reads_per_gpu": n_reads, "genome_bp": a.genome, "k": k,
                       "parallelism": "reads sharded x%d, owner-partitioned merge" % world if world > 1 else "single GPU"},
            "kmer_instances": total_instances, "distinct_reads_table": results["distinct1"],
            "counters_account_for_all_instances": bool(ok),
            "kernel_ms_per_step": kernels_ms,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not ok:
        sys.exit("bench: comp counters do not account for every k-mer instance: %s" % list(map(int, cc)))


def cpu_baseline(eng, a, k, L):
    """The oracle (a C port of the reference algorithm, oracle/koracle.c) timed on this box's host cores over a BOUNDED
    sample of the same workload: count sample reads + count a slice of the assembly (thread team over a shared CAS table,
    like Jellyfish) + comp (T x compareSlice with private accumulators, merged under a lock, like KAT)."""
    from oracle import koracle as ko
    threads = os.cpu_count() or 1
    n = min(a.cpu_sample_reads, a.reads) & ~1
    gs = min(a.genome, 20_000_000)
    g = eng.synth_genome(gs, seed=77)
    r = eng.synth_reads(g, gs, first_read=0, n_reads=n, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=3)
    asm = eng.synth_genome(gs, seed=77, contig_len=100_000)
    rh, ah = r.download(), asm.download()
    for b in (g, r, asm):
        b.free()
    t0 = time.perf_counter()
    t1 = ko.Table(k, True).count_bases(rh, threads=threads)
    t2 = ko.Table(k, True).count_bases(ah, threads=threads)
    ko.comp(t1, t2, threads=threads)
    dt = time.perf_counter() - t0
    inst = n * (L - k + 1) + max(0, gs - (gs // 100_000) * (k - 1))
    return {"value": round(inst / dt, 1), "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": "%d reads x %d bp from a %d bp genome + that genome as assembly, k=%d; %.1f s" % (n, L, gs, k, dt)}


if __name__ == "__main__":
    main()
