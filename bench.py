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

  python bench.py                                   # N = 1, config 4 at full size (+ `workloads`: configs 2, 3, 5's per-GPU shard, 3 steps each;
                                                    #   + `end_to_end`: files -> files, config 4 at full size when /dev/shm has the room)
  python bench.py --gpus 8                          # launches its own 8 ranks (torch.distributed.run, one per GPU, RCCL); weak scaling:
                                                    #   300 M reads PER GPU
  python bench.py --gpus 8 --scaling strong         # the metric's own config: 300 M reads IN TOTAL, sharded over the ranks
  python bench.py --gpus 8 --workload comp-rr       # BASELINE.json configs[4] (config 5): 1.2 G reads vs reads, k = 31, 150 M per GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      # the driver's form: the same thing
Ranks on distinct devices exchange over RCCL or not at all: a communicator that would have to stage through /dev/shm fails the run
unless --allow-shm is given (the line's config.comm names the transport either way).
"""
import argparse
import copy
import json
import re
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
PROFILE_JSON = {"comp": "profiles/r06_final_pmc_fetch_write.json", "hist": "profiles/r06_final_hist_pmc_fetch_write.json",
                "gcp": "profiles/r06_final_gcp_pmc_fetch_write.json", "comp-rr": "profiles/r06_final_comp-rr_pmc_fetch_write.json"}
# the sources the count stage's kernels are made of: a committed profile describes the kernels of ONE state of these files
# (tools/profile_bench.sh records their digest next to the counters; pmc_traffic refuses a profile taken from other code)
STAGE_SOURCES = ["kat_amd/csrc/kg_partition.hpp", "kat_amd/csrc/kg_l1_blocks.hpp", "kat_amd/csrc/kg_l2_blocks.hpp", "kat_amd/csrc/kg_device.hpp",
                 "kat_amd/csrc/kg_l1_lean.hpp", "kat_amd/csrc/kg_kernels.hpp", "kat_amd/csrc/kg_count.hip", "kat_amd/csrc/kg_table.hip"]


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
    ap.add_argument("--no-exchange-overlap", action="store_true", help="N > 1, two tables: katgpu_exchange_merge for both instead of counting input 2 while table 1 travels")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-ppm", type=int, default=2000, help="substitution errors per million bases (0.2 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> output files leg")
    ap.add_argument("--e2e-reads", type=int, default=100_000_000, help="reads of the end-to-end slice (written as FASTQ to a temp dir: 32 GB at the default)")
    ap.add_argument("--phases", action="store_true", help="sync + print per-phase wall time (diagnostic; perturbs the timing)")
    ap.add_argument("--cpu-sample-reads", type=int, default=16_000_000)
    ap.add_argument("--cpu-sample-genome", type=int, default=200_000_000, help="genome of the CPU sample: its table must be much larger than the host's last-level cache")
    ap.add_argument("--load", type=float, default=0.62, help="load factor the tables are pre-sized for (expected distinct k-mers / slots)")
    ap.add_argument("--hint-scale", type=float, default=1.0, help="diagnostic: scale the tables' size hints (e.g. 0.02: grown on the way)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"), help="N > 1: weak = the workload's reads PER GPU (default); strong = the workload's reads in total, sharded")
    ap.add_argument("--allow-shm", action="store_true", help="N > 1: let ranks on distinct devices stage the exchange through /dev/shm when RCCL cannot be had (the line says so)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` object (configs 2, 3 and 5's shard, 3 steps each) of the default run")
    ap.add_argument("--with-workloads", action="store_true", help="diagnostic / tests: the `workloads` object at a reduced size too (each capped at --reads / --genome)")
    ap.add_argument("--no-e2e-gz", action="store_true", help="skip the .fastq.gz edition of the files -> output files leg")
    ap.add_argument("--e2e-gz-reads", type=int, default=60_000_000, help="reads of the .fastq.gz edition, the first of the leg's own files (default: a fifth of config 4 -- writing the two .gz files at "
                    "gzip's level 6 is what takes the time, ~1.4 s per million reads on the 16 CPUs a GPU box grants; 150000000, half of config 4: profiles/r06_e2e_gz.txt)")
    ap.add_argument("--e2e-gz-level", type=int, default=6, help="deflate level of the .fastq.gz files (6: gzip's default)")
    ap.add_argument("--e2e-slice", action="store_true", help="files -> files on the --e2e-reads slice even when /dev/shm has room for the whole config")
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
            if base.startswith(("k_p1v2_scatter", "k_p1b_scatter", "k_s1")):          # one per round (the group and the block edition of level 1; wide tables' k_s1)
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
        # leave room beside the counter's arena for the second table, RCCL's channel buffers and the buffers of TWO exchanges under way
        # (send list + what arrives: ~18 bytes per record sent, 50 GB for config 4's table) -- with them the second input is counted while
        # the first table travels; without room the ranks agree on the one-call exchange inside the arena (must be set before the library loads)
        os.environ.setdefault("KATGPU_ARENA_FRACTION", "0.55")
        # a bench step is seconds: a single wait of ten minutes is a wedged link, and an error beats a hang (kg_comm.hip "Liveness")
        os.environ.setdefault("KATGPU_COMM_MAX_WAIT_S", "600")
        if a.allow_shm:
            os.environ["KATGPU_COMM_ALLOW_SHM"] = "1"
    import torch
    import kat_amd

    L0 = a.read_len
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")                     # rendezvous, barriers and the timing all-reduce only: the data path is katgpu's own communicator
    eng = kat_amd.Engine(local_rank)
    # The native communicator (kg_comm.hip behind the C ABI: RCCL over xGMI; /dev/shm only when the ranks share a device, or with
    # --allow-shm -- the line says which).  One code path: a communicator that cannot be made, or whose first exchange fails, fails the run.
    comm, transport, comm_info = None, None, None
    if world > 1:
        ids = [kat_amd.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = kat_amd.Comm(eng, rank, world, ids[0])
        if comm.transport != "rccl" and comm.distinct_devices > 1 and not a.allow_shm:
            sys.exit("bench: %d ranks on %d devices would exchange through /dev/shm (%s): not an xGMI measurement; pass --allow-shm to take it knowingly"
                     % (world, comm.distinct_devices, comm.transport_note or "no RCCL"))
        transport = "katgpu native exchange over %s" % ("RCCL" if comm.transport == "rccl" else "/dev/shm (%s)" % (comm.transport_note or "no RCCL"))
        # first contact between the devices, on two tiny tables: a transport that cannot work says so here, not after minutes of counting
        gp = eng.synth_genome(200_000, seed=5)
        rp = eng.synth_reads(gp, 200_000, first_read=rank * 2000, n_reads=2000, read_len=L0, frag_len=350, err_ppm=2000, seed=9)
        tp = eng.table(27, True, size_hint=1 << 21)
        tp.count_bases_device(rp.ptr, rp.nbytes)
        comm.exchange_merge(tp)
        seen = comm.allreduce_u64([np.ones(1, dtype=np.uint64)])[0]
        comm_info = {"transport": comm.transport, "note": comm.transport_note, "ranks_seen": int(seen[0]), "distinct_devices": comm.distinct_devices,
                     "shm_allowed": bool(a.allow_shm)}
        for x in (tp, rp, gp):
            x.free()

    # ---- the files -> output files leg goes first: a child process that allocates right after this one has freed a hundred GB
    # of HBM would spend seconds in the driver's scrubbing of that memory -- an artefact of benchmarking, not of the tool ----
    e2e = None
    if rank == 0 and world == 1 and not a.no_e2e:
        try:
            e2e = end_to_end(eng, a, a.k, a.read_len)
        except Exception as ex:                            # the engine number stands on its own; say why the leg is missing
            e2e = {"error": "%s: %s" % (type(ex).__name__, ex)}
        eng.release_scratch()

    ctx = {"rank": rank, "world": world, "comm": comm, "dist": dist, "transport": transport, "comm_info": comm_info, "torch": torch}
    line, ok, what = measure(eng, a, ctx, want_cpu=True)
    all_ok = ok
    if rank == 0:
        line["end_to_end_gz"] = e2e.pop("gz", None) if isinstance(e2e, dict) else None      # the same leg from .fastq.gz files (one gzip member each)
        line["end_to_end"] = e2e
    # ---- the other single-GPU configs of BASELINE.json, briefly, in the same line: configs 2, 3 and config 5's per-GPU shard ----
    if world == 1 and a.workload == "comp" and (a.default_size or a.with_workloads) and not a.no_workloads:
        extra = {}
        for wl in ("hist", "gcp", "comp-rr"):
            eng.release_scratch()                           # the previous workload's parked tables and arena go back to the driver
            b = copy.copy(a)
            b.workload, b.steps, b.warmup, b.phases = wl, 3, 1, False
            b.reads, b.genome, b.k, _ = WORKLOADS[wl]
            if not a.default_size:                          # (--with-workloads at a reduced size)
                b.reads, b.genome = min(b.reads, a.reads), min(b.genome, a.genome)
            try:
                l2, ok2, what2 = measure(eng, b, ctx, want_cpu=False)
                extra[wl] = {"config": l2["config"]["workload"], "value": l2["value"], "unit": l2["unit"], "steps": b.steps, "warmup": b.warmup,
                             "ms_per_step": l2["ms_per_step"], "kmer_instances": l2["kmer_instances"],
                             "roofline": {k_: l2["roofline"][k_] for k_ in ("achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "ms_per_step")},
                             "kernel_ms_per_step": l2["kernel_ms_per_step"], "reducers": l2["reducers"],
                             "result_accounts_for_every_kmer": bool(ok2), "result_check": what2}
                all_ok = all_ok and ok2
            except Exception as ex:
                extra[wl] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        line["workloads"] = extra
    if rank == 0:
        print(json.dumps(line))
    if comm is not None:
        comm.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not all_ok:
        sys.exit("bench: a result does not account for every k-mer: %s" % what)
    if rank == 0 and e2e and e2e.get("result_check") is False:
        sys.exit("bench: the files -> files leg wrote other results than the resident path: %s" % e2e.get("result_check_detail"))


def measure(eng, a, ctx, want_cpu):
    """One workload: synthetic inputs generated in HBM (not timed), `a.warmup` untimed steps, `a.steps` timed ones between barriers, max
    over ranks; returns (the JSON line as a dict -- filled on rank 0 --, result ok, what was checked)."""
    import kat_amd
    from kat_amd import dist as kdist
    rank, world, comm, dist, torch = ctx["rank"], ctx["world"], ctx["comm"], ctx["dist"], ctx["torch"]

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wl = a.workload
    k, L = a.k, a.read_len
    two_tables = wl in ("comp", "comp-rr")
    strong = world > 1 and a.scaling == "strong"
    reads_per_gpu = (a.reads // world) if strong else a.reads          # strong: the workload's reads in total, sharded by pair
    # ---- synthetic inputs, generated in HBM (not timed) ----
    g = eng.synth_genome(a.genome, seed=20260927)                               # genome the reads are sampled from
    if wl == "comp-rr":                                                         # two read libraries, half of --reads each
        n_reads = (reads_per_gpu // 2) & ~1
        reads = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=1)
        reads2 = eng.synth_reads(g, a.genome, first_read=rank * n_reads, n_reads=n_reads, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=2)
        in2_ptr, in2_bytes = reads2.ptr, reads2.nbytes
        inst_reads = n_reads * (L - k + 1)
        inst2_local, inst2_total = inst_reads, world * inst_reads
    else:
        n_reads = reads_per_gpu & ~1
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
        overlap = world > 1 and two_tables and not a.no_exchange_overlap
        if overlap:                                         # table 1's records travel while input 2 is counted (katgpu_exchange_begin / _finish)
            results["distinct1_local"] = t1.stats(want_total=False)["distinct"]
            comm.exchange_begin(t1)
            tp = mark("exchange_begin", tp)
        t2 = None
        if two_tables:
            t2 = eng.table(k, True, size_hint=hint2, like=t1)
            t2.count_bases_device(in2_ptr, in2_bytes)
            tp = mark("alloc2+count_2", tp)
        if not overlap:
            results["distinct1_local"] = t1.stats(want_total=False)["distinct"]
        if k <= 32:
            g1_ = t1.geometry()
            results["geo1"] = (int(g1_.p1), int(g1_.p2), t1.slot_bytes(), t2.slot_bytes() if t2 is not None else 0)
        if world > 1:
            if overlap:                                     # table 2's records travel while table 1's are applied
                eng.sync()
                comm.exchange_begin(t2)
                comm.exchange_finish(t1)
                comm.exchange_finish(t2)
            else:
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
    comm_before = comm.stats() if comm is not None else None
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile()
    comm_after = comm.stats() if comm is not None else None

    def allsum(v):
        if world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device="cpu")
        dist.all_reduce(t)
        return int(t.item())

    def allmax(v):
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = allmax(dt)
    d1_local, cap1, cap2 = results["distinct1_local"], results["cap1"], results.get("cap2", 0)
    distinct1, distinct2 = allsum(results["distinct1"]), allsum(results.get("distinct2", 0))
    exch = None
    if comm is not None:                                        # the exchange of the TIMED steps: slowest rank per phase, bytes of all ranks
        exch = {"max_over_ranks_ms_per_step": {q: round(allmax(comm_after[q] - comm_before[q]) / a.steps, 3) for q in ("extract_ms", "exchange_ms", "merge_ms", "allreduce_ms")},
                "bytes_sent_per_step_all_ranks": allsum(comm_after["bytes_sent"] - comm_before["bytes_sent"]) // a.steps,
                "bytes_sent_per_step_this_rank": int(comm_after["bytes_sent"] - comm_before["bytes_sent"]) // a.steps,
                # the records themselves (not the count matrices, not the all-reduce): 9 bytes each -- what a slot holds of the k-mer + its count --
                # between ranks whose tables have one region grid, 12 (key + count) otherwise
                "records_sent_per_step_all_ranks": allsum(comm_after["records_sent"] - comm_before["records_sent"]) // a.steps,
                "record_bytes_per_record": round(allsum(comm_after["record_bytes_sent"] - comm_before["record_bytes_sent"]) / max(1, allsum(comm_after["records_sent"] - comm_before["records_sent"])), 3),
                "records_packed": bool(comm_after["records_packed"]),
                "second_input_counted_while_table_1_travels": bool(two_tables and not a.no_exchange_overlap),
                "reading": "extract = table -> send list; exchange = posting the chunks + waiting for them (on the wire while the previous chunk is merged); merge = k_merge_apply; allreduce = the small results"}

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
    line = {}
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
                if item1 == 6 and p1_ <= 512 and 17 <= k <= 31:
                    item1 = 6.4                                                        # ... which travel in 64-byte blocks of ten (kg_l1_blocks.hpp)
                items = inst_reads + inst2_local
                rounds1 = max(1, prof["part_l1_scatter"]["launches"] // a.steps - (1 if two_tables else 0))      # rounds of the first input
                item2 = 64.0 / 12.0 if hb == 1 else 4.0 + hb                           # a level-2 item: 5-byte remainders travel in 64-byte blocks of twelve
                own = {"part_l1_scatter": (reads.nbytes + in2_bytes) + float(item1) * items,
                       "part_l2": (item1 + item2) * items,
                       "part_apply": item2 * items + 2.0 * slot_b1 * cap1 * rounds1 + 2.0 * slot_b2 * cap2}
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
        # which kernel moved, in the object the driver keeps: milliseconds per step of the three stage kernels and of the reducer
        red_ms = prof[wl]["ms"] if wl in ("hist", "gcp") else prof["comp_pass1"]["ms"] + prof["comp_pass2"]["ms"]
        roof["ms_per_step"] = {"l1": round((prof["part_l1_count"]["ms"] + prof["part_l1_scatter"]["ms"]) / a.steps, 3), "l2": round(prof["part_l2"]["ms"] / a.steps, 3),
                               "apply": round(prof["part_apply"]["ms"] / a.steps, 3), "direct": round(direct_ms / a.steps, 3), "reduce": round(red_ms / a.steps, 3)}
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
        if want_cpu and not a.no_cpu_baseline and world == 1 and k <= 32:      # the host-core baseline is an N = 1 figure (the multi-threaded oracle port is one-word)
            cpu = cpu_baseline(eng, a, k, L)
        _, _, _, wl_name = WORKLOADS[wl]
        per = "in total, sharded over %d GPUs" % world if strong else "per GPU"
        if wl == "comp":
            desc = "%s: %d x %d bp PE reads %s (0.2%% subst. errors) vs %d bp assembly in %d bp contigs, k=%d, canonical" % (wl_name, n_reads * (world if strong else 1), L, per, a.genome, a.contig, k)
        elif wl == "comp-rr":
            desc = "%s: library 1 (%d reads %s) vs library 2 (%d reads %s), %d bp PE, 0.2%% subst. errors, %d bp genome, k=%d, canonical" % (
                wl_name, n_reads * (world if strong else 1), per, n_reads * (world if strong else 1), per, L, a.genome, k)
        else:
            desc = "%s: %d x %d bp PE reads %s (0.2%% subst. errors) from a %d bp genome, k=%d, canonical" % (wl_name, n_reads * (world if strong else 1), L, per, a.genome, k)
        line = {
            "metric": "k-mers/sec (whole node) for %s k=%d" % (wl_name, k),
            "value": round(value, 1), "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc, "reads_per_gpu": n_reads * (2 if wl == "comp-rr" else 1), "genome_bp": a.genome, "k": k,
                       "parallelism": "reads sharded x%d, owner-partitioned merge: %s" % (world, ctx["transport"]) if world > 1 else "single GPU", "comm": ctx["comm_info"]},
            "kmer_instances": total_instances, "distinct_table1": distinct1, "distinct_table2": distinct2 if two_tables else None,
            "result_accounts_for_every_kmer": bool(ok), "result_check": what,
            "kernel_ms_per_step": kernels_ms,
            "roofline": roof, "reducers": red, "cpu_baseline": cpu,
        }
        if exch is not None:
            line["exchange"] = exch
    return line, ok, what


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
    gs = min(a.genome, a.cpu_sample_genome)                     # a table much larger than the host's last-level cache, as the full config's is
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

    # the sweep starts where round 4's sweeps peaked (a shared CAS table stops scaling long before 256 threads) and stays within ~40 s
    sweep = {}
    plan = [t for t in (32, 64, 16, 128, 8) if t <= cores] or [cores]
    for th in plan:
        sweep[th] = run(th)
        if sum(sweep.values()) > 40.0:                      # bounded: the default run stays within minutes
            break
    skipped = [t for t in plan if t not in sweep]             # thread counts the time bound cut off: the line says so (the headline ratio is tied to reference_scaled anyway)
    best = min(sweep, key=sweep.get)
    table_mb = 12.0 * expected_distinct(n * (L - k + 1), gs, k, a.err_ppm) / 0.7 / 1e6
    return {"value": round(inst / sweep[best], 1), "unit": "k-mers/s", "cores": best, "kind": "port",
            "sample": "%d reads x %d bp from a %d bp genome%s, k=%d; best of a thread sweep: %.2f s at %d threads; the sample's table (~%.0f MB of keys + counts) against %s of last-level cache" % (
                n, L, gs, {"comp": " + that genome as assembly", "comp-rr": " + a second library of the same size"}.get(wl, ""), k, sweep[best], best, table_mb, host_llc()),
            "thread_sweep_kmers_per_s": {str(t): round(inst / s, 1) for t, s in sorted(sweep.items())},
            "sweep_truncated": bool(skipped), "sweep_skipped_threads": skipped, "sweep_order": plan,
            "host_cores": cores, "host_cpus_granted": effective_cpus(),       # (what the container may really use of them: affinity, cgroup cpu.max -- the GPU boxes show 256 and grant 16, which is why the sweep peaks at 16-32 threads)
            "host_last_level_cache": host_llc(),
            "note": "a port of the reference's algorithm (oracle/koracle.c), not the reference binary (unbuildable here: DESIGN.md section 5).  The sample is sized so that its table is "
                    "much larger than the last-level cache, like the full config's; a smaller sample would flatter the CPU.",
            "reference_scaled": {"value": REF_KMERS_PER_CORE * cores, "unit": "k-mers/s",
                                 "source": "SURVEY.md section 6: the reference's count phase ran at ~3 M k-mers/s/core (8-core Xeon, hand-built reference binary); x %d host cores, assuming it scales linearly (it does not: an upper bound)" % cores}}


def host_llc():
    """The host's last-level cache as the kernel reports it ("32768K x 16 instances" style), or "unknown"."""
    try:
        base = "/sys/devices/system/cpu/cpu0/cache"
        best = None
        for idx in sorted(os.listdir(base)):
            lv = os.path.join(base, idx, "level")
            if os.path.exists(lv):
                level = int(open(lv).read())
                if best is None or level >= best[0]:
                    best = (level, open(os.path.join(base, idx, "size")).read().strip(), open(os.path.join(base, idx, "shared_cpu_list")).read().strip())
        if best is None:
            return "unknown"
        return "L%d %s per instance (shared by CPUs %s)" % best
    except Exception:
        return "unknown"


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


def parse_stats(text):
    """The 13 counters of a `kat comp` .stats file (CompCounters::printCounts, lib/src/comp_counters.cc:144-206) in the order of the
    device's counter block: totals 1-3, distinct 1-3, only-total 1-2, only-distinct 1-2, shared total 1-2, shared distinct."""
    sec = {}
    cur = None
    for ln in text.splitlines():
        if ln and not ln.startswith(" "):
            cur = ln.strip().rstrip(":").strip()
            sec[cur] = {}
        elif cur is not None:
            m = re.match(r"\s*-\s*(.+?):\s*(\d+)\s*$", ln)
            if m:
                sec[cur][m.group(1)] = int(m.group(2))
    g = lambda s_, key: sec.get(s_, {}).get(key, 0)
    return [g("Total K-mers in", "Hash 1"), g("Total K-mers in", "Hash 2"), g("Total K-mers in", "Hash 3"),
            g("Distinct K-mers in", "Hash 1"), g("Distinct K-mers in", "Hash 2"), g("Distinct K-mers in", "Hash 3"),
            g("Total K-mers only found in", "Hash 1"), g("Total K-mers only found in", "Hash 2"),
            g("Distinct K-mers only found in", "Hash 1"), g("Distinct K-mers only found in", "Hash 2"),
            g("Shared K-mers", "Total shared found in hash 1"), g("Shared K-mers", "Total shared found in hash 2"), g("Shared K-mers", "Distinct shared K-mers")]


def parse_mx(path):
    """The body of a .mx file (SparseMatrix::printMatrix, lib/include/kat/sparse_matrix.hpp:269-277) as a uint64 matrix."""
    rows = []
    with open(path, "rb") as f:
        for ln in f:
            if ln.startswith(b"#") or not ln.strip():
                continue
            rows.append(np.array(ln.split(), dtype=np.uint64))
    return np.vstack(rows) if rows else np.zeros((0, 0), np.uint64)


def parse_hist(path):
    """`kat hist` output (Histogram::print, src/histogram.cc:131-144): "<count> <distinct k-mers>" per line after the # header."""
    vals = []
    with open(path, "rb") as f:
        for ln in f:
            if ln.startswith(b"#") or not ln.strip():
                continue
            vals.append(int(ln.split()[1]))
    return np.array(vals, dtype=np.uint64)


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (Python's zlib module does not export it): the CRC-32 of A + B from those of A and B and len(B)."""
    if len2 <= 0:
        return crc1
    sq = lambda m: [_gf2_times(m, m[n]) for n in range(32)]
    odd = [0xedb88320] + [1 << n for n in range(31)]
    even = sq(odd)
    odd = sq(even)
    while True:
        even = sq(odd)
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = sq(even)
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def effective_cpus():
    """CPUs this process may really use: os.cpu_count() cut down to the affinity mask and to the cgroup's CPU quota (the GPU boxes show 256
    CPUs; what a container gets of them is in cpu.max)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


_GZ_Q = None


def _gz_quals(read_len):
    global _GZ_Q
    if _GZ_Q is None or _GZ_Q.shape[1] != read_len:
        rng = np.random.default_rng(12345)
        _GZ_Q = np.frombuffer(b"F:,#", np.uint8)[rng.choice(4, size=(4099, read_len), p=[0.86, 0.08, 0.04, 0.02])]
    return _GZ_Q


def _gz_block(job):
    """One block of gzip_records_file, in a worker process: records [lo, hi) of the file deflated with the 32 KiB before them as dictionary."""
    import zlib
    src, n_records, rec_bytes, read_len, level, lo, hi = job
    back = (32768 + rec_bytes - 1) // rec_bytes
    lo0 = max(0, lo - back)
    mm = np.memmap(src, dtype=np.uint8, mode="r", shape=(n_records, rec_bytes))
    rec = np.array(mm[lo0:hi])
    del mm
    q0 = 14 + read_len + 3
    rec[:, q0:q0 + read_len] = _gz_quals(read_len)[(np.arange(lo0, hi) * 2654435761 % 4099)]
    raw = rec.reshape(-1).tobytes()
    cut = (lo - lo0) * rec_bytes
    data = memoryview(raw)[cut:]
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, raw[max(0, cut - 32768):cut]) if cut else zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    out = co.compress(data) + co.flush(zlib.Z_FINISH if hi == n_records else zlib.Z_SYNC_FLUSH)
    return out, zlib.crc32(data), len(data)


def gzip_records_file(src, dst, n_records, rec_bytes, read_len, level, workers, pool=None):
    """The first n_records fixed-size FASTQ records of `src` as ONE gzip member in `dst`, the way pigz writes one: blocks of records
    deflated by a pool of worker processes, each with the 32 KiB before it as its dictionary and closed by a sync flush, CRC-32s combined
    -- a single deflate stream with copies across the block borders, no member boundaries to cut at.  Quality lines (write_fastq's are all
    'I') are re-drawn from four symbols, as binned Illumina qualities are, so that the stream is not unrealistically compressible; no
    reader looks at them.  Returns (compressed bytes, plain bytes)."""
    import multiprocessing as mp
    per_block = max(256, (16 << 20) // rec_bytes)
    jobs = [(src, n_records, rec_bytes, read_len, level, lo, min(n_records, lo + per_block)) for lo in range(0, n_records, per_block)]
    own = pool is None
    if own:
        pool = mp.get_context("forkserver").Pool(max(1, min(workers, len(jobs))))     # (not fork: this process has the HIP runtime's threads)
    crc, total, written = 0, 0, 0
    try:
        with open(dst, "wb") as f:
            hdr = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"
            f.write(hdr)
            written += len(hdr)
            for out, c, n in pool.imap(_gz_block, jobs, chunksize=1):
                f.write(out)
                written += len(out)
                crc = crc32_combine(crc, c, n) if total else c
                total += n
            tail = (crc & 0xFFFFFFFF).to_bytes(4, "little") + (total & 0xFFFFFFFF).to_bytes(4, "little")
            f.write(tail)
            written += len(tail)
    finally:
        if own:
            pool.close()
            pool.join()
    return written, total


def end_to_end(eng, a, k, L):
    """Files -> output files through the C++ host binary (kat_amd/bin/katgpu, the mirror of KAT's drivers over the C ABI): the span of
    the reference's "Total runtime" (src/comp.cc:750; process start -> outputs closed, no plots).  At the workload's FULL size when
    /dev/shm has the room for its files (config 4: 96 GB), else on a bounded slice (--e2e-reads); `config` says which ran.  Inputs are
    written outside the timed span and stay in RAM, so this is parse + PCIe + count + reduce + write, not storage.
    result_check: the files the binary wrote are parsed back and compared, number by number, with the result of counting the SAME
    reads (same generator, same seeds) resident in HBM through the C ABI -- the path `value` times."""
    import kat_amd
    exe = os.path.join(ROOT, "kat_amd", "bin", "katgpu")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    wl = a.workload
    rec_bytes = 2 * L + 18                                                    # a FASTQ record as write_fastq writes it

    def shm_room():
        try:
            st = os.statvfs("/dev/shm")
            return st.f_bavail * st.f_frsize
        except OSError:
            return 0

    def need_bytes(n_, gs_):
        return int(n_ * rec_bytes * (2 if wl == "comp-rr" else 1) * 1.01) + (int(gs_ * 1.02) if wl == "comp" else 0) + (1 << 30)
    n_full = a.reads & ~1
    full = not a.e2e_slice and shm_room() > need_bytes(n_full, a.genome) + (16 << 30)
    n = n_full if full else min(a.e2e_reads, a.reads) & ~1
    if wl == "comp-rr":
        n = (n // 2) & ~1                                                     # two libraries of half the reads each, as the resident leg
    gs = a.genome if full else min(a.genome, max(10_000_000, n * 5))          # the slice: ~30x coverage of its genome
    # the inputs go where reading them back cannot depend on this process's own write-back: /dev/shm (RAM-backed by construction)
    # when it has the room, else the default temp dir (page cache; the line says which)
    tmp_root = "/dev/shm" if shm_room() > need_bytes(n, gs) + (8 << 30) else None
    tmp = tempfile.mkdtemp(prefix="katgpu_e2e_", dir=tmp_root)
    t_e2e0 = time.perf_counter()
    G_SEED, R_SEED, R2_SEED = 20260927, 1, 2                                  # the resident leg's generator seeds (measure())
    try:
        g = eng.synth_genome(gs, seed=G_SEED)
        inst, nbytes = 0, 0

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
        lib1 = library(R_SEED, "lib1")
        inst += n * (L - k + 1)
        second = None
        clen = a.contig
        if wl == "comp":
            asm = g.download()
            second = os.path.join(tmp, "asm.fa")
            with open(second, "wb") as f:
                for c in range((gs + clen - 1) // clen):
                    seq = asm[c * clen:(c + 1) * clen]
                    f.write(b">contig%d\n" % c)
                    pad = (-seq.size) % 80
                    lines = np.concatenate([seq, np.full(pad, ord("\n"), np.uint8)]).reshape(-1, 80)
                    body = np.concatenate([lines, np.full((lines.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
                    f.write(body.rstrip(b"\n") + b"\n")
                    inst += max(0, seq.size - k + 1)
            del asm
        elif wl == "comp-rr":
            second = " ".join(library(R2_SEED, "lib2"))
            inst += n * (L - k + 1)
        g.free()
        eng.sync()
        eng.release_scratch()                               # the child process needs the device memory this one has parked
        for root, _, fs in os.walk(tmp):
            nbytes += sum(os.path.getsize(os.path.join(root, f)) for f in fs)
        hint = int(expected_distinct(n * (L - k + 1), gs, k, a.err_ppm) / 0.62) + (1 << 20)
        hint2 = hint if wl == "comp-rr" else int(gs / 0.62) + (1 << 20)
        outp = os.path.join(tmp, "out")
        tool = {"hist": "hist", "gcp": "gcp"}.get(wl, "comp")
        cmd = [exe, tool, "-t", "16", "-m", str(k), "-H", str(hint), "-o", outp]
        if second is not None:                              # comp: -I sizes the second hash (KAT's -H / -I)
            cmd += ["-I", str(hint2)]
        if second is not None:                              # comp takes one (quoted) argument per input group, hist / gcp a list of files
            cmd += [" ".join(lib1), second]
        else:
            cmd += lib1
        t_gen = time.perf_counter() - t_e2e0
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, KATGPU_TIMING="1", KATGPU_TRACE="1"))
        dt = time.perf_counter() - t0
        if pr.returncode != 0:
            raise RuntimeError("katgpu %s exited %d: %s" % (tool, pr.returncode, (pr.stderr or pr.stdout)[-400:]))
        outs = [f for f in os.listdir(tmp) if f.startswith("out")]
        # where the span went: the binary's own timing lines (KATGPU_TIMING=1): per phase of the run and per input file
        phases, per_file, bad_lines = {}, [], 0
        for line in pr.stderr.splitlines():
            if not line.startswith("katgpu_timing "):
                continue
            try:
                rec = json.loads(line[len("katgpu_timing "):])
            except ValueError:
                bad_lines += 1                              # (flagged below: a file missing from the breakdown overstates count_phases_outside_files_ms)
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
                     "files": per_file, "unparsable_timing_lines": bad_lines,
                     "reading": "per file: setup = open + map + device / pinned buffers; wall = the file's whole pass; reader_wait = the main thread waiting for file bytes to reach the device (reader threads: "
                                "pread into pinned memory, then their own H2D copy; pread / h2d per thread say which of the two it was); scan = the record scan "
                                "on the device -- FASTQ files are stripped to their sequence lines by the reader threads instead (read_by says so): no scan kernel, scan = issuing their copies, "
                                "pread = read + strip --; counter_wait = waiting for the counting worker; counting = what the worker spent (hidden under the rest unless "
                                "counter_wait says otherwise)"}
        # the library's own time line of its allocations (KATGPU_TRACE): on some boxes the driver takes seconds to hand out tens of GB
        # (files[0].setup_ms says so); these lines say which allocation it was
        breakdown["alloc_trace"] = [l[:200] for l in pr.stderr.splitlines() if l.startswith("[katgpu") and any(w in l for w in ("alloc", "arena", "scan buffers", "context on"))][:16]
        if os.environ.get("KATGPU_TRACE"):                 # diagnostic: the whole time line
            breakdown["trace"] = [l for l in pr.stderr.splitlines() if l.startswith("[katgpu")][:80]
        # ---- result_check: what the binary wrote against the same reads counted resident through the C ABI ----
        def resident_check(n, outp, inst):
            check, detail = None, None
            try:
                g = eng.synth_genome(gs, seed=G_SEED)
                r1 = eng.synth_reads(g, gs, first_read=0, n_reads=n, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=R_SEED)
                t1 = eng.table(k, True, size_hint=hint)
                t1.count_bases_device(r1.ptr, r1.nbytes)
                r1.free()
                if wl in ("comp", "comp-rr"):
                    if wl == "comp":
                        in2 = eng.synth_genome(gs, seed=G_SEED, contig_len=clen)
                    else:
                        in2 = eng.synth_reads(g, gs, first_read=0, n_reads=n, read_len=L, frag_len=350, err_ppm=a.err_ppm, seed=R2_SEED)
                    t2 = eng.table(k, True, size_hint=hint2, like=t1)
                    t2.count_bases_device(in2.ptr, in2.nbytes)
                    in2.free()
                    mx, cc, sp = kat_amd.comp(t1, t2)
                    t2.free()
                    got_cc = parse_stats(open(outp + ".stats").read())
                    got_mx = parse_mx(outp + "-main.mx")
                    ok_cc = [int(x) for x in cc] == got_cc
                    ok_mx = got_mx.shape == mx.shape and bool(np.array_equal(got_mx, mx))
                    check = ok_cc and ok_mx and int(cc[0]) + int(cc[1]) == inst
                    detail = "out.stats: 13 counters %s the resident path's (hash1 total %d, distinct %d; hash2 total %d, distinct %d; shared distinct %d); out-main.mx: %dx%d body %s (sum %d); totals %s the %d instances written" % (
                        "==" if ok_cc else "!=", got_cc[0], got_cc[3], got_cc[1], got_cc[4], got_cc[12], got_mx.shape[0], got_mx.shape[1] if got_mx.ndim == 2 else 0,
                        "==" if ok_mx else "!=", int(got_mx.sum()), "==" if int(cc[0]) + int(cc[1]) == inst else "!=", inst)
                elif wl == "hist":
                    want = t1.hist()
                    got = parse_hist(outp)
                    check = got.shape == want.shape and bool(np.array_equal(got, want))
                    detail = "hist file: %d bins %s the resident path's (distinct %d)" % (got.size, "==" if check else "!=", int(got.sum()))
                else:
                    want = t1.gcp()
                    got = parse_mx(outp + ".mx")
                    check = got.shape == want.shape and bool(np.array_equal(got, want))
                    detail = "gcp matrix: %s body %s the resident path's (sum %d)" % ("x".join(map(str, got.shape)), "==" if check else "!=", int(got.sum()))
                t1.free()
                g.free()
                eng.sync()
            except Exception as ex:
                check, detail = False, "check failed to run: %s: %s" % (type(ex).__name__, ex)
            return check, detail
        check, detail = resident_check(n, outp, inst)

        # ---- the same run from .fastq.gz: what the reference reads every day (one zlib stream per file there: stream_manager.hpp:133-145) ----
        def gz_leg():
            import zlib
            n_gz = min(n, a.e2e_gz_reads) & ~1
            if n_gz < 2:
                return None
            import multiprocessing as mp
            threads = max(1, min(effective_cpus(), 192))
            t_c0 = time.perf_counter()
            gz_paths, gz_bytes, plain_bytes = [], 0, 0
            pool = mp.get_context("forkserver").Pool(threads)                   # (one pool for both files; not fork: this process has the HIP runtime's threads)
            try:
                for src in lib1:
                    dst = src + ".gz"
                    w, t = gzip_records_file(src, dst, n_gz // 2, rec_bytes, L, a.e2e_gz_level, threads, pool)
                    gz_paths.append(dst)
                    gz_bytes += w
                    plain_bytes += t
            finally:
                pool.close()
                pool.join()
            t_comp = time.perf_counter() - t_c0
            # one zlib stream on a bounded sample of the first file: what a single inflate thread does on this host
            dco, got, t_z0 = zlib.decompressobj(31), 0, time.perf_counter()
            with open(gz_paths[0], "rb") as f:
                fed = 0
                while fed < (192 << 20):
                    piece = f.read(8 << 20)
                    if not piece:
                        break
                    fed += len(piece)
                    got += len(dco.decompress(piece))
            t_z = time.perf_counter() - t_z0
            eng.sync()
            eng.release_scratch()                           # (the resident check's tables and arena: the child process needs the device memory)
            outg = os.path.join(tmp, "outgz")
            inst_gz = n_gz * (L - k + 1) + (inst - n * (L - k + 1) if wl == "comp" else 0)
            hint_gz = int(expected_distinct(n_gz * (L - k + 1), gs, k, a.err_ppm) / 0.62) + (1 << 20)
            cmd_gz = [exe, tool, "-t", "16", "-m", str(k), "-H", str(hint_gz), "-o", outg]
            if second is not None:
                cmd_gz += ["-I", str(hint2), " ".join(gz_paths), second]
            else:
                cmd_gz += gz_paths
            t_g0 = time.perf_counter()
            pg = subprocess.run(cmd_gz, capture_output=True, text=True, timeout=1800, env=dict(os.environ, KATGPU_TIMING="1", KATGPU_TRACE="1"))
            dt_gz = time.perf_counter() - t_g0
            if pg.returncode != 0:
                raise RuntimeError("katgpu %s on .gz inputs exited %d: %s" % (tool, pg.returncode, (pg.stderr or pg.stdout)[-400:]))
            nonlocal hint
            keep_hint, hint = hint, hint_gz
            try:
                ok, why = resident_check(n_gz, outg, inst_gz)
            finally:
                hint = keep_hint
            teams = [l[:520] for l in pg.stderr.splitlines() if "one gzip stream" in l]
            files = []
            for line in pg.stderr.splitlines():
                if line.startswith("katgpu_timing ") and '"file"' in line:
                    try:
                        rec = json.loads(line[len("katgpu_timing "):])
                        files.append({"file": os.path.basename(rec["file"]), "wall_ms": rec.get("wall_ms"), "bytes": rec.get("bytes")})
                    except ValueError:
                        pass
            gz_file_ms = sum(f_["wall_ms"] or 0.0 for f_ in files if f_["file"].endswith(".gz"))
            one = fed / max(t_z, 1e-9) / 1e9
            return {"seconds": round(dt_gz, 3), "value": round(inst_gz / dt_gz, 1), "unit": "k-mers/s", "result_check": ok, "result_check_detail": why,
                    "config": "%d reads x %d bp as two .fastq.gz files (ONE gzip member each, pigz-shaped: deflate level %d, blocks that copy across their borders), %.1f GB compressed, %.1f GB of FASTQ%s" % (
                        n_gz, L, a.e2e_gz_level, gz_bytes / 1e9, plain_bytes / 1e9, " + the %d bp FASTA assembly (plain)" % gs if wl == "comp" else ""),
                    "compressed_GB_per_s_whole_run": round(gz_bytes / dt_gz / 1e9, 3),
                    "compressed_GB_per_s_while_reading_the_gz_files": round(gz_bytes / max(gz_file_ms, 1e-3) / 1e6, 3) if gz_file_ms else None,
                    "fastq_GB_per_s_while_reading_the_gz_files": round(plain_bytes / max(gz_file_ms, 1e-3) / 1e6, 3) if gz_file_ms else None,
                    "one_zlib_stream_GB_per_s": {"compressed": round(one, 3), "fastq": round(got / max(t_z, 1e-9) / 1e9, 3), "sample": "the first %d MB of the first file through zlib.decompressobj, one thread" % (fed >> 20)},
                    "times_one_zlib_stream": round(gz_bytes / max(gz_file_ms, 1e-3) / 1e6 / max(one, 1e-9), 1) if gz_file_ms else None,
                    "files": files, "teams": teams[:4], "files_compressed_in_s": round(t_comp, 1), "compress_workers": threads, "cpus_usable": effective_cpus(), "cpus_shown": os.cpu_count(),
                    **({"trace": [l[:300] for l in pg.stderr.splitlines() if l.startswith("[katgpu")][:120]} if os.environ.get("KATGPU_TRACE") else {})}
        gz = None
        if not a.no_e2e_gz and wl != "comp-rr":
            try:
                gz = gz_leg()
            except Exception as ex:
                gz = {"error": "%s: %s" % (type(ex).__name__, ex)}
        return {"value": round(inst / dt, 1), "config": ("the workload at FULL size" if full else "a slice of the workload") + ": %d reads x %d bp, %d bp genome" % (n * (2 if wl == "comp-rr" else 1), L, gs),
                "full_size": bool(full), "result_check": check, "result_check_detail": detail,
                "breakdown": breakdown, "inputs_in": tmp_root or tempfile.gettempdir(), "unit": "k-mers/s", "seconds": round(dt, 3), "input_bytes": nbytes,
                "input_GB_per_s": round(nbytes / dt / 1e9, 2), "kmer_instances": inst,
                "span": "process start -> output files closed (src/comp.cc:750 'Total runtime'), inputs in the page cache",
                "files_written_in_s": round(t_gen, 1),
                "command": "katgpu %s -t 16 -m %d -H %d on %d reads x %d bp (2 FASTQ files%s)" % (
                    tool, k, hint, n, L, {"comp": " + a %d bp FASTA assembly" % gs, "comp-rr": " + a second library"}.get(wl, "")),
                "outputs": sorted(outs), "phases": [l.strip() for l in pr.stdout.splitlines() if "Time taken" in l or "Total runtime" in l][:8],
                "gz": gz}
    finally:
        for root, _, fs in os.walk(tmp, topdown=False):
            for f in fs:
                os.unlink(os.path.join(root, f))
            os.rmdir(root)


if __name__ == "__main__":
    main()
