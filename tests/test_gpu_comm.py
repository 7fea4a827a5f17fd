"""The native exchange (include/katgpu.h: katgpu_comm_* / katgpu_exchange_merge / katgpu_allreduce_u64; kg_comm.hip) end to end:
N processes count shards, merge by owner in place, reduce on the owned shards, all-reduce -- bit-identical to one process.
On a one-GPU box the ranks share the device, which RCCL refuses, so world > 1 runs the SHM transport (the same protocol code above
it: region-ordered extraction, chunked grouped transfers overlapped with k_merge_apply, out-of-band records); world == 1 runs the
RCCL transport itself (communicator, stream, events, ncclAllGather / ncclAllReduce with one rank; grouped send / recv have no peer)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
K, G, N_READS, CONTIG = 27, 400000, 60000, 50000


def _run(tmp_path, world, mode, env_extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("KATGPU_COMM_INIT_TIMEOUT_S", "60")
    env.update(env_extra)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_rank.py"), str(r), str(world), str(tmp_path / "id.bin"), str(tmp_path), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=400)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    if any(p.returncode for p in procs) and any("did not return within" in o and "KATGPU_COMM_INIT_TIMEOUT_S" in o for o in outs) and "FAKE_RCCL_HANG_INIT" not in env:
        pytest.skip("RCCL's bootstrap did not come back on this box: " + next(o for o in outs if "did not return within" in o)[-300:])    # the box's, not the code's
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(o[-3000:] for o in outs)
    return outs[0]


def _check(ko, tmp_path, world, mode):
    from kat_amd import synth
    got = np.load(tmp_path / "sharded.npz")
    k = 31 if mode == "rr31" else 45 if mode == "wide45" else K
    T = ko.WideTable if k > 32 else ko.Table
    g = synth.genome(G, seed=11)
    o1 = T(k, True).count_bases(synth.reads(g, 0, N_READS, seed=1))
    o1.add(12345, world * (1 << 33) + world * (world - 1) // 2)
    o2 = T(k, True).count_bases(synth.reads(g, 0, N_READS, seed=2) if mode == "rr31" else synth.stream_of_contigs(g, CONTIG))
    mx, cc, sp = ko.comp(o1, o2, 1.0, 1.0, 201, 101)
    assert np.array_equal(got["cc"], cc) and np.array_equal(got["mx"], mx) and np.array_equal(got["sp"], sp)
    assert np.array_equal(got["h"], o1.hist(1, 300, 1)) and np.array_equal(got["gm"], o1.gcp(1.0, 100))


@pytest.mark.parametrize("world,mode,extra", [
    (2, "same", {}), (3, "same", {"KATGPU_TEST_EXCHANGE_CHUNKS": "7"}), (2, "mixed", {}), (2, "rr31", {}),
    (2, "same", {"KATGPU_TEST_REGION_SLOTS": "512"}),                     # packed tables on the wire's both ends
    (3, "wide45", {}),                                                    # k = 45: the wide exchange (records all to all, table refilled)
    (4, "same", {}), (8, "same", {"KATGPU_TEST_EXCHANGE_CHUNKS": "5"})])  # the world sizes of the scaling run (1, 2, 4, 8)
def test_native_exchange_ranks_sharing_one_gpu(ko, tmp_path, world, mode, extra):
    out = _run(tmp_path, world, mode, dict(extra, KATGPU_COMM_TRANSPORT="shm"))
    assert "transport: shm" in out
    _check(ko, tmp_path, world, mode)


@pytest.mark.parametrize("world,transport,extra,packed", [
    (2, "shm", {}, True), (4, "shm", {"KATGPU_TEST_EXCHANGE_CHUNKS": "6", "KATGPU_TEST_LAZY_MIN_SLOTS": "1024"}, True),      # (lazy: the emptied table is not cleared, the merge is its first sweep -- as for every table of size)
    (8, "rccl", {"KATGPU_TEST_EXCHANGE_CHUNKS": "5", "KATGPU_TEST_LAZY_MIN_SLOTS": "1024"}, True),
    (3, "rccl", {"KATGPU_COMM_PACKED_RECORDS": "0"}, False)])
def test_records_travel_in_nine_bytes_between_ranks_that_share_the_grid(ko, tmp_path, fake_rccl, world, transport, extra, packed):
    """Ranks whose tables have one region grid (regions of 128 slots here, so that small tables are packed ones, as every table of size is) exchange what a slot holds of the k-mer + its count: 9 bytes per record, not key + count's 12
    (katgpu_comm_wire); the merged result is the oracle's either way."""
    import re
    env = dict(extra, KATGPU_COMM_TRANSPORT=transport, KATGPU_TESTING="1", KATGPU_TEST_REGION_SLOTS="128")
    if transport == "rccl":
        env["KATGPU_RCCL_LIB"] = fake_rccl
    out = _run(tmp_path, world, "same", env)
    assert "transport: %s" % transport in out, out[-2000:]
    m = re.search(r"wire after table 1: \{'records_sent': (\d+), 'record_bytes_sent': (\d+), 'records_packed': (True|False)", out)      # (the second table of these runs is a small KV12 one: key + count either way)
    assert m, out[-2000:]
    records, nbytes = int(m.group(1)), int(m.group(2))
    assert records > 0 and (m.group(3) == "True") == packed and nbytes == records * (9 if packed else 12), m.group(0)
    _check(ko, tmp_path, world, "same")


@pytest.mark.parametrize("world,transport,mode,extra,want", [
    (2, "shm", "same", {}, "all on the wire at once"), (4, "rccl", "same", {"KATGPU_TEST_EXCHANGE_CHUNKS": "6", "KATGPU_TEST_REGION_SLOTS": "128", "KATGPU_TEST_LAZY_MIN_SLOTS": "1024"}, "all on the wire at once"),
    (8, "rccl", "rr31", {"KATGPU_TEST_SPLIT_TWO": "1"}, "all on the wire at once"), (2, "rccl", "mixed", {}, "all on the wire at once"),
    (3, "shm", "same", {"KATGPU_TEST_SPLIT_TWO": "1", "KATGPU_TEST_LAZY_MIN_SLOTS": "1024"}, "all on the wire at once"),        # (two exchanges under way: begin(t2) before finish(t1))
    (3, "rccl", "same", {"KATGPU_TEST_EXCHANGE_NO_SPLIT": "1"}, "the pipelined one, now"), (2, "shm", "wide45", {}, None)])
def test_the_second_input_is_counted_while_the_first_table_travels(ko, tmp_path, fake_rccl, world, transport, mode, extra, want):
    """katgpu_exchange_begin(table 1) -- the second input counted, in the arena the exchange no longer uses -- katgpu_exchange_finish(table 1):
    the result is the oracle's, as katgpu_exchange_merge's is; a rank without room for the exchange's own buffer (here: a hook) makes every
    rank run the pipelined exchange inside begin; wide tables do it all in begin."""
    env = dict(extra, KATGPU_COMM_TRANSPORT=transport, KATGPU_TESTING="1", KATGPU_TEST_SPLIT_EXCHANGE="1", KATGPU_COMM_TRACE="1")
    if transport == "rccl":
        env["KATGPU_RCCL_LIB"] = fake_rccl
    out = _run(tmp_path, world, mode, env)
    assert "transport: %s" % transport in out, out[-2000:]
    if want:
        assert want in out, out[-3000:]
    _check(ko, tmp_path, world, mode)


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    """tests/native/fake_rccl.cc built into a shared library: the ten nccl* entry points kg_comm.hip resolves, over /dev/shm + hipMemcpy,
    for ranks that share a device (real RCCL refuses them)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = str(tmp_path_factory.mktemp("fakerccl") / "libfakerccl.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", "-O1", os.path.join(HERE, "native", "fake_rccl.cc"), "-o", so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


@pytest.mark.parametrize("world,mode,extra", [
    (2, "same", {}), (3, "same", {"KATGPU_TEST_EXCHANGE_CHUNKS": "7"}), (2, "mixed", {}), (3, "rr31", {"KATGPU_TEST_EXCHANGE_CHUNKS": "3"}),
    (2, "wide45", {}),
    (4, "mixed", {}), (8, "same", {"KATGPU_TEST_EXCHANGE_CHUNKS": "5"}), (8, "wide45", {})])      # 8: the node's world size
def test_native_exchange_rccl_branch_with_several_ranks(ko, tmp_path, fake_rccl, world, mode, extra):
    """The RCCL transport code of kg_comm.hip -- grouped ncclSend / ncclRecv per chunk on the transport stream, events, the chunk
    double-buffering against k_merge_apply, ncclAllGather of the sizes, ncclAllReduce of the results -- with 2 and 3 ranks: the
    library behind the calls is the stand-in (the box has one GPU), everything above it is the product's."""
    out = _run(tmp_path, world, mode, dict(extra, KATGPU_COMM_TRANSPORT="rccl", KATGPU_RCCL_LIB=fake_rccl, KATGPU_TESTING="1"))
    assert "transport: rccl" in out, out[-2000:]
    _check(ko, tmp_path, world, mode)


def test_shared_device_is_told_and_auto_takes_shm_there(ko, tmp_path):
    """KATGPU_COMM_TRANSPORT unset (auto) with two ranks on ONE device: real RCCL refuses them, the ranks are seen to share a device
    (katgpu_comm_distinct_devices == 1), and the staging transport carries them -- its home ground, no flag needed."""
    out = _run(tmp_path, 2, "same", {})
    assert "transport: shm" in out and "devices: 1" in out, out[-2000:]
    _check(ko, tmp_path, 2, "same")


def test_a_dead_peer_is_told_from_its_heartbeat(tmp_path):
    """A rank that dies without a word (killed) leaves its peer waiting at the exchange: the peer gives up when the dead rank's heartbeat
    has stood still for KATGPU_COMM_TIMEOUT_S -- not before (a slow peer is not a dead one), and not never."""
    import signal
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KATGPU_COMM_TRANSPORT="shm", KATGPU_COMM_TIMEOUT_S="4", KATGPU_TEST_STALL_RANK="1", KATGPU_TEST_STALL_S="600")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_rank.py"), str(r), "2", str(tmp_path / "id.bin"), str(tmp_path), "same"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        t0 = time.time()
        while not (tmp_path / "stalled.1").exists():                     # rank 1 has joined the communicator and now sleeps (alive: its heartbeat runs)
            assert time.time() - t0 < 300 and procs[1].poll() is None, "rank 1 never reached its stall"
            time.sleep(0.05)
        time.sleep(8.0)                                                  # twice the liveness bound: rank 0 waits for a LIVE peer all that time
        assert procs[0].poll() is None, "rank 0 gave up on a peer that was alive: " + (procs[0].communicate()[0] or "")[-1500:]
        procs[1].send_signal(signal.SIGKILL)
        out0 = procs[0].communicate(timeout=120)[0]
        assert procs[0].returncode != 0 and "no sign of life from rank 1" in out0, out0[-2000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize("mode", ["same", "wide45"])
def test_native_exchange_single_rank_over_rccl(ko, tmp_path, mode):
    out = _run(tmp_path, 1, mode, {"KATGPU_COMM_TRANSPORT": "rccl"})
    assert "transport: rccl" in out, out[-2000:]
    _check(ko, tmp_path, 1, mode)


def test_an_rccl_bootstrap_that_never_returns_is_an_error_not_a_hang(tmp_path, fake_rccl):
    """ncclCommInitRank that never comes back (the stand-in sleeps for ever): the rank reports it after KATGPU_COMM_INIT_TIMEOUT_S and
    exits non-zero (kg_comm.hip: rccl_boot_call) -- seen once in 200 runs with the real library on a box of the pool."""
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KATGPU_COMM_TRANSPORT="rccl", KATGPU_RCCL_LIB=fake_rccl, KATGPU_TESTING="1",
               FAKE_RCCL_HANG_INIT="1", KATGPU_COMM_INIT_TIMEOUT_S="3")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(HERE, "comm_rank.py"), "0", "1", str(tmp_path / "id.bin"), str(tmp_path), "same"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
    assert p.returncode != 0 and "ncclCommInitRank did not return within 3 s" in p.stdout, p.stdout[-2000:]
    assert time.time() - t0 < 150
