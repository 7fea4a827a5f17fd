#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: k-mers/s of the kat hist / gcp / comp hot path on MI355X.

A step = one pass of the hot path over one synthetic batch already resident in HBM:
    allocate tables -> count (partition rounds) -> [N>1: owner-partitioned exchange + merge] -> reduce on the device
    (hist / gcp / comp) -> D2H of the result (80 KB / 216 KB / 8 MB + counters).
value = (valid k-mer instances of all inputs on all ranks) / (max-over-ranks wall time).  Host file parsing and PCIe are
outside the timed region by contract (inputs resident); the `end_to_end` object of the line times files -> output files
through the C++ host binary on a bounded slice of the same workload (never `value`).

Workloads (BASELINE.json configs[1..4]; --config N is an alias, N counted from 1 as SURVEY.md 8(d) does):
  --workload comp      config 4 (default): kat comp, 300 M x 150 bp PE reads vs the 1 Gbp assembly, k = 27
  --workload hist      config 2: kat hist, 50 M reads from a 100 Mbp genome, k = 27
  --workload gcp       config 3: kat gcp, 100 M reads from a 200 Mbp genome, k = 27
  --workload comp-rr   config 5: kat comp, reads library 1 vs reads library 2 (75 M + 75 M reads per GPU), k = 31

  python bench.py                                   # N = 1, config 4 at full size
  python bench.py --gpus 8                          # launches its own 8 ranks (torch.distributed.run, one per GPU, RCCL)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      # the driver's form: the same thing
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REF_KMERS_PER_CORE = 3.0e6    # the reference's count phase per host core, measured by the survey (23-25 M k-mers/s on 8 cores)

WORKLOADS = {
    #           reads/GPU     genome         k   what
    "comp":    (300_000_000, 1_000_000_000, 27, "kat comp reads-vs-assembly"),
    "hist":    (50_000_000,  100_000_000,   27, "kat hist"),
    "gcp":     (100_000_000, 200_000_000,   27, "kat gcp"),
    "comp-rr": (150_000_000, 1_000_000_000, 31, "kat comp reads-vs-reads"),
}
CONFIG_ALIAS = {2: "hist", 3: "gcp", 4: "comp", 5: "comp-rr"}
PROFILE_JSON = {"comp": "profiles/r04_final_pmc_fetch_write.json", "hist": "profiles/r04_final_hist_pmc_fetch_write.json",
                "gcp": "profiles/r04_final_gcp_pmc_fetch_write.json", "comp-rr": "profiles/r04_final_comp-rr_pmc_fetch_write.json"}
# the sources the count stage's kernels are made of: a committed profile describes the kernels of ONE state of these files
# (tools/profile_bench.sh records their digest next to the counters; pmc_traffic refuses a profile taken from other code)
STAGE_SOURCES = ["kat_amd/csrc/kg_partition.hpp", "kat_amd/csrc/kg_device.hpp", "kat_amd/csrc/kg_l1_lean.hpp", "kat_amd/csrc/kg_kernels.hpp",
                 "kat_amd/csrc/kg_count.hip", "kat_amd/csrc/kg_table.hip"]


def stage_sources_digest():
    import hashlib
    h = hashlib.sha256()
    for rel in STAGE_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIG_ALIAS), help="BASELINE.json config number (1-based)")
    ap.add_argument("--reads", type=int, default=None, help="150 bp reads PER GPU (PE: reads/2 pairs); default: the workload's")
    ap.add_argument("--genome", type=int, default=None, help="genome / assembly length in bp (shared by all ranks)")
    ap.add_argument("--contig", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-ppm", type=int, default=2000, help="substitution errors per million bases (0.2 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> output files leg")
    ap.add_argument("--e2e-reads", type=int, default=100_000_000, help="reads of the end-to-end slice (written as FASTQ to a temp dir: 32 GB at the default)")
    ap.add_argument("--phases", action="store_true", help="sync + print per-phase wall time (diagnostic; perturbs the timing)")
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--load", type=float, default=0.62, help="load factor the tables are pre-sized for (expected distinct k-mers / slots)")
    ap.add_argument("--hint-scale", type=float, default=1.0, help="diagnostic: scale the tables' size hints (e.g. 0.02: grown on the way)")
    a = ap.parse_args()
    if a.config is not None:
        if a.workload is not None and a.workload != CONFIG_ALIAS[a.config]:
            ap.error("--config %d is --workload %s" % (a.config, CONFIG_ALIAS[a.config]))
        a.workload = CONFIG_ALIAS[a.config]
    if a.workload is None:
        a.workload = "comp"
    reads, genome, k, _ = WORKLOADS[a.workload]
    a.default_size = a.reads is None and a.genome is None and a.k is None
    a.reads = reads if a.reads is None else a.reads
    a.genome = genome if a.genome is None else a.genome
    a.k = k if a.k is None else a.k
    return a


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N` of this script."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


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
    path = os.path.join(ROOT, PROFILE_JSON[a.workload])
    if world != 1 or not a.default_size or (a.contig, a.read_len, a.err_ppm) != (1_000_000, 150, 2000):
        return None, None
    try:
        prof = json.load(open(path))
    except Exception:
        return None, "no committed counter profile (%s)" % PROFILE_JSON[a.workload]
    have, want = prof.get("_stage_sources_sha256_16"), stage_sources_digest()
    if have != want:                                     # counters of other kernels than the ones that just ran: not this run's traffic
        return None, "%s was taken from other stage-kernel sources (digest %s, now %s): re-run tools/profile_bench.sh" % (PROFILE_JSON[a.workload], have, want)
    kb, rounds = 0.0, 0
    for name, e in prof.items():
        if name.startswith("_"):
            continue
        base = name.replace("kg::", "").replace("void ", "").split("<")[0].split("(")[0]
        if base.startswith(("k_p1", "k_p2", "k_p3", "k_s1", "k_s2", "k_s3", "k_insert_keys")):
            kb += 2.0 * e.get("FETCH_SIZE_KB_total", 0.0) + e.get("WRITE_SIZE_KB_total", 0.0)
            if base.startswith(("k_p1v2_scatter", "k_s1")):          # one per round
                rounds += e.get("launches", 0)
    if not rounds:
        return None, None
    return int(kb * 1024 / rounds), "%s: (2 x FETCH_SIZE + WRITE_SIZE) of the count-stage kernels / %d rounds" % (PROFILE_JSON[a.workload], rounds)


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    if world > 1:
        # the exchange keeps its send list and receive buffers inside the arena and allocates nothing else; leave room for the
        # second table, RCCL's channel buffers and the small per-region count matrices (must be set before the library loads)
        os.environ.setdefault("KATGPU_ARENA_FRACTION", "0.75")
    import torch
    import kat_amd
    from kat_amd import dist as kdist

    L0 = a.read_len
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")                     # rendezvous, barriers and the timing all-reduce only: the data path is katgpu's own communicator
    eng = kat_amd.Engine(local_rank)
    # The native communicator (kg_comm.hip behind the C ABI: RCCL over xGMI, /dev/shm when ranks share a device or RCCL cannot be had --
    # the line says which).  One code path: a communicator that cannot be made, or whose first exchange fails, fails the run.
    comm, transport, comm_info = None, None, None
    if world > 1:
        ids = [kat_amd.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = kat_amd.Comm(eng, rank, world, ids[0])
        transport = "katgpu native exchange over %s" % ("RCCL" if comm.transport == "rccl" else "/dev/shm (%s)" % (comm.transport_note or "no RCCL"))
        # first contact between the devices, on two tiny tables: a transport that cannot work says so here, not after minutes of counting
        gp = eng.synth_genome(200_000, seed=5)
        rp = eng.synth_reads(gp, 200_000, first_read=rank * 2000, n_reads=2000, read_len=L0, frag_len=350, err_ppm=2000, seed=9)
        tp = eng.table(27, True, size_hint=1 << 21)
        tp.count_bases_device(rp.ptr, rp.nbytes)
        comm.exchange_merge(tp)
        seen = comm.allreduce_u64([np.ones(1, dtype=np.uint64)])[0]
        comm_info = {"transport": comm.transport, "note": comm.transport_note, "ranks_seen": int(seen[0])}
        for x in (tp, rp, gp):
            x.free()

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wl = a.workload
    k, L = a.k, a.read_len
    two_tables = wl in ("comp", "comp-rr")
    # ---- the files -> output files leg goes first: a child process that allocates right after this one has freed a hundred GB
    # of HBM would spend seconds in the driver's scrubbing of that memory -- an artefact of benchmarking, not of the tool ----
    e2e = None
    if rank == 0 and world == 1 and not a.no_e2e:
        try:
            e2e = end_to_end(eng, a, k, L)
        except Exception as ex:                            # the engine number stands on its own; say why the leg is missing
            e2e = {"error": "%s: %s" % (type(ex).__name__, ex)}
    # ---- synthetic inputs, generated in HBM (not timed) ----
    g = eng.synth_genome(a.genome, seed=20260927)                               # genome the reads are sampled from
    if wl == "comp-rr":                                                         # two read libraries, half of --reads each
        n_reads = (a.reads // 2) & ~1
        reads = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=1)
        reads2 = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=2)
        in2_ptr, in2_bytes = reads2.ptr, reads2.nbytes
        inst_reads = n_reads * (L - k + 1)
        inst2_local, inst2_total = inst_reads, world * inst_reads
    else:
        n_reads = a.reads & ~1
        reads = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=1)
        inst_reads = n_reads * (L - k + 1)
        inst2_local = inst2_total = 0
        in2_ptr = in2_bytes = 0
    if wl == "comp":                                                            # assembly = that genome cut into contigs, sharded over ranks
        n_contigs = (a.genome + a.contig - 1) // a.contig
        c_lo, c_hi = kdist.shard_range(n_contigs, rank, world)
        asm_full = eng.synth_genome(a.genome, seed=20260927, contig_len=a.contig)
        in2_ptr = asm_full.ptr + c_lo * (a.contig + 1)
        in2_bytes = min(asm_full.nbytes, c_hi * (a.contig + 1)) - c_lo * (a.contig + 1)
        asm_bases_local = min(a.genome, c_hi * a.contig) - c_lo * a.contig
        inst2_local = max(0, asm_bases_local - (c_hi - c_lo) * (k - 1))
        inst2_total = max(0, a.genome - n_contigs * (k - 1))
    g.free()
    eng.sync()

    hint1 = int(expected_distinct(inst_reads, a.genome, k, a.err_ppm) / a.load) + (1 << 20)
    hint2 = 0
    if wl == "comp":
        hint2 = int(asm_bases_local / a.load) + (1 << 20)
    elif wl == "comp-rr":
        hint2 = hint1
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

    def exchange(t):
        comm.exchange_merge(t)                              # katgpu_exchange_merge: in place, RCCL behind the C ABI (k > 32: records all to all, the table refilled)
        return t

    def step(verify=False):
        tp = time.perf_counter()
        t1 = eng.table(k, True, size_hint=hint1)
        tp = mark("alloc1", tp)
        t1.count_bases_device(reads.ptr, reads.nbytes)
        tp = mark("count_reads", tp)
        t2 = None
        if two_tables:
            t2 = eng.table(k, True, size_hint=hint2, like=t1)
            t2.count_bases_device(in2_ptr, in2_bytes)
            tp = mark("alloc2+count_2", tp)
        results["distinct1_local"] = t1.stats(want_total=False)["distinct"]
        if k <= 32:
            g1_ = t1.geometry()
            results["geo1"] = (int(g1_.p1), int(g1_.p2), t1.slot_bytes(), t2.slot_bytes() if t2 is not None else 0)
        if world > 1:
            t1 = exchange(t1)
            if t2 is not None:
                t2 = exchange(t2)
            tp = mark("exchange", tp)
        if wl == "hist":
            out = [t1.hist()]
        elif wl == "gcp":
            out = [t1.gcp()]
        else:
            out = list(kat_amd.comp(t1, t2))
        tp = mark("reduce", tp)
        if world > 1:
            out = comm.allreduce_u64(out)
        results["out"] = out
        st = t1.stats(want_total=verify)
        results["distinct1"], results["cap1"] = st["distinct"], st["capacity"]
        if verify:
            results["total1"] = st["total"]
        if t2 is not None:
            st2 = t2.stats(want_total=False)
            results["distinct2"], results["cap2"] = st2["distinct"], st2["capacity"]
            t2.free()
        t1.free()
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

    def allsum(v):
        if world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device="cpu")
        dist.all_reduce(t)
        return int(t.item())

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    d1_local, cap1, cap2 = results["distinct1_local"], results["cap1"], results.get("cap2", 0)
    distinct1, distinct2 = allsum(results["distinct1"]), allsum(results.get("distinct2", 0))

    total_instances = world * inst_reads + inst2_total
    value = total_instances * a.steps / dt

    # ---- sanity: the result must account for every instance / every distinct k-mer (size-independent parity properties) ----
    out = results["out"]
    if two_tables:
        cc = out[1]
        ok = int(cc[0]) == world * inst_reads and int(cc[1]) == inst2_total
        what = "comp counters: hash1 total %d (expected %d), hash2 total %d (expected %d)" % (int(cc[0]), world * inst_reads, int(cc[1]), inst2_total)
    else:
        step(verify=True)                                   # once more, untimed, with the table's sum of counts
        barrier()
        tot = allsum(results["total1"])
        cells = int(results["out"][0].sum())
        # hist: every distinct k-mer lands in one bucket; gcp: in one cell, except the handful whose GC count is k -- the row the reference drops
        ok = tot == world * inst_reads and (cells == distinct1 if wl == "hist" else cells <= distinct1 and cells >= distinct1 - 64)
        what = "sum of counts %d (expected %d), result cells %d vs %d distinct" % (tot, world * inst_reads, cells, distinct1)

    if a.phases and rank == 0:
        print("phases (s, summed over steps):", {n: round(v, 3) for n, v in phases.items()}, file=sys.stderr)
    if rank == 0:
        # ---- roofline of the count stage, from HIP events recorded on katgpu's own stream ----
        # algorithmic bytes: per instance L/(L-k+1) B of ASCII + 8 B key read + 4 B count read + 4 B count
        # write, plus 8 B key write per distinct k-mer; summed over this rank's count work of the timed steps.
        per_inst = L / (L - k + 1) + 16.0 + (8.0 if k > 32 else 0.0)               # k > 32: a second key word per slot
        d2_local = results.get("distinct2", 0) if world == 1 else inst2_local      # upper bound on what the second count wrote
        alg_bytes_step = per_inst * (inst_reads + inst2_local) + (16.0 if k > 32 else 8.0) * (d1_local + d2_local)
        stage = ["part_l1_count", "part_l1_scatter", "part_l2", "part_apply"]
        part_ms = sum(prof[n]["ms"] for n in stage)
        direct_ms = prof["count"]["ms"]
        if part_ms >= direct_ms:
            # partitioned counter: one "launch" = one round = the stage kernels over the round's k-mers (level 1 once, level 2 and the
            # apply once per pass of the round: per_kernel has their own launch counts)
            rounds = max(1, prof["part_l1_scatter"]["launches"])
            stage_ms = part_ms + direct_ms
            name = "count stage (partitioned): per round level-1 scatter (+ count/scan when exact), then level 2 + apply in passes"
            per_kernel = {n: {"launches": prof[n]["launches"], "avg_ms": round(prof[n]["ms"] / max(1, prof[n]["launches"]), 3)} for n in stage}
            # each stage kernel against the roofline of its OWN bytes (what it has to move, not what the PMC counters say it moved):
            #   level 1: the ASCII stream in, an item per k-mer out (6 bytes at k = 27);  level 2: those in, (4 + hb)-byte remainders out (hb from the
            #   table's remainder bits);  apply: the remainders in + the table swept in and out once per round (slot_bytes per slot)
            if k <= 32 and results.get("geo1") is not None:
                p1_, p2_, slot_b1, slot_b2 = results["geo1"]
                rb = int(kat_amd.binding.place_keys(k, p1_, max(0, p2_.bit_length() - 1), [])[4])
                hi_bytes = lambda bits: 0 if bits <= 31 else 1 if bits <= 39 else 2 if bits <= 47 else 4
                hb = hi_bytes(rb)
                item1 = 4 + hi_bytes(2 * k - (p1_.bit_length() - 1))                   # a level-1 item: the k-mer below its level-1 digit
                items = inst_reads + inst2_local
                rounds1 = max(1, prof["part_l1_scatter"]["launches"] // a.steps - (1 if two_tables else 0))      # rounds of the first input
                own = {"part_l1_scatter": (reads.nbytes + in2_bytes) + float(item1) * items,
                       "part_l2": (item1 + 4.0 + hb) * items,
                       "part_apply": (4.0 + hb) * items + 2.0 * slot_b1 * cap1 * rounds1 + 2.0 * slot_b2 * cap2}
                for n, b in own.items():
                    ms = prof[n]["ms"] / a.steps
                    if ms > 0:
                        per_kernel[n].update({"own_bytes_per_step": int(b), "own_GBps": round(b / (ms / 1e3) / 1e9, 1), "own_frac": round(b / (ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4)})
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
        # the reducers' own roofline.  SURVEY.md 8(d) prices them at 12 B per slot (hist / gcp: 12 B x C; comp: 12 B x (C1 + C2) scan
        # + 12 B x (D1 + D2) probes) -- the reference's key + count.  A packed table stores 8 B per slot and the fused join probes in
        # LDS, so the kernels MOVE less than the formula: `frac` is priced on the bytes the slots really hold (slot_bytes x slots, the
        # conservative figure; it agrees with the PMC passes in profiles/), `survey_formula` carries the 12-byte figure beside it.
        slot = 20.0 if k > 32 else 12.0
        sb1, sb2 = (results["geo1"][2], results["geo1"][3]) if results.get("geo1") is not None else (slot, slot)
        sb1, sb2 = float(sb1 or slot), float(sb2 or slot)

        def red_entry(ms, moved, formula):
            gbps = lambda b: round(b / (ms / 1e3) / 1e9, 1) if ms else None
            return {"avg_ms": round(ms, 3), "alg_bytes": int(moved), "achieved_GBps": gbps(moved),
                    "survey_formula": {"alg_bytes": int(formula), "achieved_GBps": gbps(formula)}}
        red = {}
        if wl in ("hist", "gcp"):
            ms = prof[wl]["ms"] / max(1, prof[wl]["launches"])
            red["k_" + wl] = red_entry(ms, sb1 * cap1, slot * cap1)
        else:
            ms = (prof["comp_pass1"]["ms"] + prof["comp_pass2"]["ms"]) / max(1, prof["comp_pass1"]["launches"])
            red["k_comp pass 1 + pass 2"] = red_entry(ms, sb1 * cap1 + sb2 * cap2,
                                                      slot * (cap1 + cap2) + slot * (results["distinct1"] + results.get("distinct2", 0)))
        for v in red.values():
            v["frac"] = round(v["achieved_GBps"] / HBM_PEAK_GBPS, 4) if v["achieved_GBps"] else None
        kernels_ms = {n: round(v["ms"] / a.steps, 3) for n, v in prof.items() if v["launches"]}
        cpu = None
        if not a.no_cpu_baseline and world == 1 and k <= 32:      # the host-core baseline is an N = 1 figure (the multi-threaded oracle port is one-word)
            cpu = cpu_baseline(eng, a, k, L)
        _, _, _, wl_name = WORKLOADS[wl]
        if wl == "comp":
            desc = "%s: %d x %d bp PE reads per GPU (0.2%% subst. errors) vs %d bp assembly in %d bp contigs, k=%d, canonical" % (wl_name, n_reads, L, a.genome, a.contig, k)
        elif wl == "comp-rr":
            desc = "%s: library 1 (%d reads per GPU) vs library 2 (%d reads per GPU), %d bp PE, 0.2%% subst. errors, %d bp genome, k=%d, canonical" % (wl_name, n_reads, n_reads, L, a.genome, k)
        else:
            desc = "%s: %d x %d bp PE reads per GPU (0.2%% subst. errors) from a %d bp genome, k=%d, canonical" % (wl_name, n_reads, L, a.genome, k)
        line = {
            "metric": "k-mers/sec (whole node) for %s k=%d" % (wl_name, k),
            "value": round(value, 1), "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc, "reads_per_gpu": n_reads * (2 if wl == "comp-rr" else 1), "genome_bp": a.genome, "k": k,
                       "parallelism": "reads sharded x%d, owner-partitioned merge: %s" % (world, transport) if world > 1 else "single GPU", "comm": comm_info},
            "kmer_instances": total_instances, "distinct_table1": distinct1, "distinct_table2": distinct2 if two_tables else None,
            "result_accounts_for_every_kmer": bool(ok), "result_check": what,
            "kernel_ms_per_step": kernels_ms,
            "roofline": roof, "reducers": red, "cpu_baseline": cpu, "end_to_end": e2e,
        }
        if comm is not None:
            st = comm.stats()
            line["exchange_ms_per_step"] = {k_: round(v / (a.steps + a.warmup), 3) for k_, v in st.items() if k_.endswith("_ms")}
            line["exchange_bytes_sent_per_step"] = int(st["bytes_sent"] / (a.steps + a.warmup))
        print(json.dumps(line))
    if comm is not None:
        comm.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not ok:
        sys.exit("bench: the result does not account for every k-mer: %s" % what)


def cpu_baseline(eng, a, k, L):
    """The oracle (a C port of the reference algorithm, oracle/koracle.c) timed on this box's host cores over a BOUNDED sample of
    the same workload: count sample reads (+ a slice of the second input) with a thread team over a shared CAS table, like
    Jellyfish, then the workload's reducer (comp: T x compareSlice with private accumulators, merged under a lock, like KAT).
    The thread count is swept (a shared CAS table stops scaling long before 256 threads) and the best is reported; next to it
    `reference_scaled` = the reference's own measured per-core count rate x this box's cores."""
    from oracle import koracle as ko
    cores = os.cpu_count() or 1
    wl = a.workload
    n = min(a.cpu_sample_reads, a.reads) & ~1
    gs = min(a.genome, 20_000_000)
    g = eng.synth_genome(gs, seed=77)
    r = eng.synth_reads(g, gs, first_read=0, n_reads=n, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=3)
    rh = r.download()
    second = None
    inst = n * (L - k + 1)
    if wl == "comp":
        asm = eng.synth_genome(gs, seed=77, contig_len=100_000)
        second = asm.download()
        asm.free()
        inst += max(0, gs - (gs // 100_000) * (k - 1))
    elif wl == "comp-rr":
        r2 = eng.synth_reads(g, gs, first_read=0, n_reads=n, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=4)
        second = r2.download()
        r2.free()
        inst += n * (L - k + 1)
    for b in (g, r):
        b.free()

    def run(threads):
        t0 = time.perf_counter()
        t1 = ko.Table(k, True).count_bases(rh, threads=threads)
        if second is not None:
            t2 = ko.Table(k, True).count_bases(second, threads=threads)
            ko.comp(t1, t2, threads=threads)
        elif wl == "hist":
            t1.hist()
        else:
            t1.gcp()
        return time.perf_counter() - t0

    sweep = {}
    for th in sorted({t for t in (8, 16, 32, 64, 128, 256, cores) if t <= cores}):
        sweep[th] = run(th)
        if sum(sweep.values()) > 40.0:                      # bounded: the default run stays within minutes
            break
    best = min(sweep, key=sweep.get)
    return {"value": round(inst / sweep[best], 1), "unit": "k-mers/s", "cores": best, "kind": "port",
            "sample": "%d reads x %d bp from a %d bp genome%s, k=%d; best of a thread sweep: %.2f s at %d threads" % (
                n, L, gs, {"comp": " + that genome as assembly", "comp-rr": " + a second library of the same size"}.get(wl, ""), k, sweep[best], best),
            "thread_sweep_kmers_per_s": {str(t): round(inst / s, 1) for t, s in sweep.items()},
            "host_cores": cores,
            "reference_scaled": {"value": REF_KMERS_PER_CORE * cores, "unit": "k-mers/s",
                                 "source": "SURVEY.md section 6: the reference's count phase ran at ~3 M k-mers/s/core (8-core Xeon, hand-built reference binary); x %d host cores, assuming it scales linearly (it does not: an upper bound)" % cores}}


def write_fastq(f, bases, first_read, mate, read_len):
    """bases: uint8 [n, read_len] -> the open file f.  4-line FASTQ with fixed-width headers (@r<9-digit pair>/<mate>), quality 'I'."""
    n = bases.shape[0]
    rec = np.empty((n, 2 * read_len + 18), np.uint8)
    ids = np.arange(first_read, first_read + n, dtype=np.int64)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    for d in range(9):
        rec[:, 2 + d] = (ids // 10 ** (8 - d)) % 10 + ord("0")
    rec[:, 11] = ord("/")
    rec[:, 12] = ord("1") + mate
    rec[:, 13] = ord("\n")
    rec[:, 14:14 + read_len] = bases
    p = 14 + read_len
    rec[:, p] = ord("\n")
    rec[:, p + 1] = ord("+")
    rec[:, p + 2] = ord("\n")
    rec[:, p + 3:p + 3 + read_len] = ord("I")
    rec[:, p + 3 + read_len] = ord("\n")
    f.write(rec.data)


def end_to_end(eng, a, k, L):
    """Files -> output files through the C++ host binary (kat_amd/bin/katgpu, the mirror of KAT's drivers over the C ABI) on a
    bounded slice of the workload: the span of the reference's "Total runtime" (src/comp.cc:750; process start -> outputs
    closed, no plots).  Inputs are written to a temp dir outside the timed span; the page cache is warm,
    so this is parse + PCIe + count + reduce + write, not storage."""
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    wl = a.workload
    n = min(a.e2e_reads, a.reads) & ~1
    gs = min(a.genome, max(10_000_000, n * 5))                               # ~30x coverage of the slice's genome
    # the inputs go where reading them back cannot depend on this process's own write-back: /dev/shm (RAM-backed by construction)
    # when it has the room, else the default temp dir (page cache; the line says which)
    need = int(n * (2 * L + 14) * 1.02) + (gs if wl == "comp" else n * (2 * L + 14)) + (1 << 30)
    tmp_root = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > need + (8 << 30):
            tmp_root = "/dev/shm"
    except OSError:
        pass
    tmp = tempfile.mkdtemp(prefix="katgpu_e2e_", dir=tmp_root)
    t_e2e0 = time.perf_counter()
    try:
        g = eng.synth_genome(gs, seed=99)
        files, inst, nbytes = [], 0, 0

        def library(seed, tag):                                                 # in slices of 8 M reads: the files are tens of GB
            paths = [os.path.join(tmp, "%s_R%d.fastq" % (tag, m + 1)) for m in (0, 1)]
            files = [open(p, "wb") for p in paths]
            step = 8_000_000
            for lo in range(0, n, step):
                m = min(step, n - lo)
                r = eng.synth_reads(g, gs, first_read=lo, n_reads=m, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=seed)
                h = r.download().reshape(m, L + 1)[:, :L]
                r.free()
                for mate in (0, 1):
                    write_fastq(files[mate], h[mate::2], lo // 2, mate, L)
            for f in files:
                f.close()
            return paths
        lib1 = library(5, "lib1")
        inst += n * (L - k + 1)
        second = None
        if wl == "comp":
            asm = eng.synth_genome(gs, seed=99).download()
            second = os.path.join(tmp, "asm.fa")
            with open(second, "wb") as f:
                clen = 1_000_000
                for c in range((gs + clen - 1) // clen):
                    seq = asm[c * clen:(c + 1) * clen]
                    f.write(b">contig%d\n" % c)
                    pad = (-seq.size) % 80
                    lines = np.concatenate([seq, np.full(pad, ord("\n"), np.uint8)]).reshape(-1, 80)
                    body = np.concatenate([lines, np.full((lines.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
                    f.write(body.rstrip(b"\n") + b"\n")
                    inst += max(0, seq.size - k + 1)
        elif wl == "comp-rr":
            second = " ".join(library(6, "lib2"))
            inst += n * (L - k + 1)
        g.free()
        eng.sync()
        eng.release_scratch()                               # the child process needs the device memory this one has parked
        for root, _, fs in os.walk(tmp):
            nbytes += sum(os.path.getsize(os.path.join(root, f)) for f in fs)
        hint = int(expected_distinct(n * (L - k + 1), gs, k, a.err_ppm) / 0.62) + (1 << 20)
        outp = os.path.join(tmp, "out")
        tool = {"hist": "hist", "gcp": "gcp"}.get(wl, "comp")
        cmd = [exe, tool, "-t", "16", "-m", str(k), "-H", str(hint), "-o", outp]
        if second is not None:                              # comp: -I sizes the second hash (KAT's -H / -I)
            cmd += ["-I", str(hint if wl == "comp-rr" else int(gs / 0.62) + (1 << 20))]
        if second is not None:                              # comp takes one (quoted) argument per input group, hist / gcp a list of files
            cmd += [" ".join(lib1), second]
        else:
            cmd += lib1
        t_gen = time.perf_counter() - t_e2e0
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, KATGPU_TIMING="1"))
        dt = time.perf_counter() - t0
        if pr.returncode != 0:
            raise RuntimeError("katgpu %s exited %d: %s" % (tool, pr.returncode, (pr.stderr or pr.stdout)[-400:]))
        outs = [f for f in os.listdir(tmp) if f.startswith("out")]
        # where the span went: the binary's own timing lines (KATGPU_TIMING=1): per phase of the run and per input file
        phases, per_file = {}, []
        for line in pr.stderr.splitlines():
            if not line.startswith("katgpu_timing "):
                continue
            try:
                rec = json.loads(line[len("katgpu_timing "):])
            except ValueError:
                continue
            if "phase" in rec:
                key = rec["phase"] if rec["phase"] != "count" else "count_input_%d" % (1 + sum(1 for q in phases if q.startswith("count_input_")))
                phases[key + "_ms"] = rec["ms"]
            elif "file" in rec:
                rec["file"] = os.path.basename(rec["file"])
                rec["GB_per_s"] = round(rec["bytes"] / max(rec["wall_ms"], 1e-3) / 1e6, 2)
                per_file.append(rec)
        accounted = sum(v for q, v in phases.items() if q != "total_ms")
        in_files = sum(f.get("setup_ms", 0.0) + f.get("wall_ms", 0.0) for f in per_file)
        breakdown = {"process_wall_ms": round(dt * 1e3, 1), "phases": phases, "unaccounted_ms": round(dt * 1e3 - accounted, 1),
                     # what the two count phases spent outside the files' passes: waiting for the table / arena allocations, the last counts
                     "count_phases_outside_files_ms": round(phases.get("count_input_1_ms", 0.0) + phases.get("count_input_2_ms", 0.0) - in_files, 1),
                     "files": per_file,
                     "reading": "per file: setup = open + map + device / pinned buffers; wall = the file's whole pass; reader_wait = the main thread waiting for file bytes to reach the device (reader threads: "
                                "pread into pinned memory, then their own H2D copy; pread / h2d per thread say which of the two it was); scan = the record scan "
                                "on the device; counter_wait = waiting for the counting worker; counting = what the worker spent (hidden under the rest unless "
                                "counter_wait says otherwise)"}
        if os.environ.get("KATGPU_TRACE"):                 # diagnostic: the library's own time line of the run
            breakdown["trace"] = [l for l in pr.stderr.splitlines() if l.startswith("[katgpu")][:80]
        return {"value": round(inst / dt, 1), "breakdown": breakdown, "inputs_in": tmp_root or tempfile.gettempdir(), "unit": "k-mers/s", "seconds": round(dt, 3), "input_bytes": nbytes,
                "input_GB_per_s": round(nbytes / dt / 1e9, 2), "kmer_instances": inst,
                "span": "process start -> output files closed (src/comp.cc:750 'Total runtime'), inputs in the page cache",
                "files_written_in_s": round(t_gen, 1),
                "command": "katgpu %s -t 16 -m %d -H %d on %d reads x %d bp (2 FASTQ files%s)" % (
                    tool, k, hint, n, L, {"comp": " + a %d bp FASTA assembly" % gs, "comp-rr": " + a second library"}.get(wl, "")),
                "outputs": sorted(outs), "phases": [l.strip() for l in pr.stdout.splitlines() if "Time taken" in l or "Total runtime" in l][:8]}
    finally:
        for root, _, fs in os.walk(tmp, topdown=False):
            for f in fs:
                os.unlink(os.path.join(root, f))
            os.rmdir(root)


if __name__ == "__main__":
    main()
