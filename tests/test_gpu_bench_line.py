"""`bench.py` as the driver runs it, at a reduced size: the line carries what DESIGN.md claims it carries -- the `workloads` object
(configs 2, 3, 5's shard), an `end_to_end` leg whose written files are checked against the resident path (`result_check`), and, for
N > 1, the communicator's transport, the ranks seen, the per-phase exchange times and bytes.  The N > 1 runs put 8 ranks on this
box's one device through the RCCL branch of kg_comm.hip against tests/native/fake_rccl.cc (real RCCL refuses ranks that share a
device): the code path of the driver's 8-GPU run, minus xGMI."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--reads", "2000000", "--genome", "5000000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[0])


def test_single_gpu_line_carries_workloads_and_a_checked_end_to_end_leg():
    line = _bench(SMALL + ["--e2e-reads", "2000000", "--with-workloads"], {"KATGPU_PGZ_CHUNK": str(1 << 20)})
    assert line["result_accounts_for_every_kmer"] and line["n_gpus"] == 1 and line["scaling"] == "weak"
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    e = line["end_to_end"]
    assert e.get("error") is None, e
    assert e["result_check"] is True, e["result_check_detail"]
    # (--reads 2000000 IS this run's whole workload, and it fits /dev/shm: the leg runs it at full size and says so)
    assert e["full_size"] is True and "FULL size" in e["config"] and e["kmer_instances"] == 2000000 * 124 + (5000000 - 5 * 26)
    assert e["breakdown"]["unparsable_timing_lines"] == 0
    z = line["end_to_end_gz"]                                                 # the same run from two .fastq.gz files, one gzip member each, inflated by the team
    assert z.get("error") is None, z
    assert z["result_check"] is True, z["result_check_detail"]
    assert len(z["teams"]) == 2 and all("one gzip stream" in t for t in z["teams"]) and z["compressed_GB_per_s_whole_run"] > 0, z
    w = line["workloads"]
    assert sorted(w) == ["comp-rr", "gcp", "hist"]
    for name, x in w.items():
        assert x.get("error") is None, (name, x)
        assert x["result_accounts_for_every_kmer"] and x["ms_per_step"] > 0 and 0 < x["roofline"]["frac"] < 1, (name, x)
    assert w["comp-rr"]["kmer_instances"] == 2 * 1000000 * 120


@pytest.mark.parametrize("wl", ["hist", "comp-rr"])
def test_end_to_end_check_of_the_other_tools(wl):
    line = _bench(["--workload", wl] + SMALL + ["--e2e-reads", "2000000"])
    e = line["end_to_end"]
    assert e.get("error") is None and e["result_check"] is True, e


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = str(tmp_path_factory.mktemp("fakerccl") / "libfakerccl.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", "-O1", os.path.join(ROOT, "tests", "native", "fake_rccl.cc"), "-o", so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


@pytest.mark.parametrize("workload,scaling,tiny_regions", [("comp", "strong", False), ("comp-rr", "weak", False), ("comp-rr", "weak", True), ("comp", "weak", True)])
def test_eight_ranks_through_the_rccl_branch(fake_rccl, workload, scaling, tiny_regions):
    """`bench.py --gpus 8` (it launches its own torch.distributed.run): 8 ranks, the communicator's RCCL branch, the line's comm block.
    With regions of 128 slots the test's small tables are what the bench's large ones are -- packed -- and
    the records of two read libraries (config 5's shape: both tables on one grid) cross the wire in 9 bytes each; `comp` at this size has
    fewer contigs than ranks, so some ranks' assembly tables are empty ones of another grid that receive far more than they held: key +
    count records through the direct path, room made for each share of the table as it arrives."""
    env = {"KATGPU_TESTING": "1", "KATGPU_RCCL_LIB": fake_rccl, "KATGPU_COMM_TRANSPORT": "rccl", "KATGPU_ARENA_FRACTION": "0.08"}
    if tiny_regions:
        env["KATGPU_TEST_REGION_SLOTS"] = "128"
        env["KATGPU_TEST_LAZY_MIN_SLOTS"] = "1024"            # (the exchanged tables are left uncleared, the merge is their first sweep: what tables of size do)
    # (comp-rr: k = 29, where these small tables have what config 5's k = 31 tables of 39 GB have -- packed slots, a remainder of more than 40 bits)
    line = _bench(["--gpus", "8", "--workload", workload, "--scaling", scaling] + SMALL + (["--k", "29"] if tiny_regions and workload == "comp-rr" else []), env, timeout=1500)
    if tiny_regions:
        x = {q: line["exchange"][q] for q in ("records_sent_per_step_all_ranks", "record_bytes_per_record", "records_packed")}
        assert x["records_sent_per_step_all_ranks"] > 0, x
        if workload == "comp-rr":
            assert x["records_packed"] and x["record_bytes_per_record"] <= 9.1, x
        else:
            assert 9.0 < x["record_bytes_per_record"] < 12.0, x                 # (the reads' table in 9-byte records, the assembly's in 12)
    assert line["n_gpus"] == 8 and line["scaling"] == scaling and line["result_accounts_for_every_kmer"], line["result_check"]
    c = line["config"]["comm"]
    assert c["transport"] == "rccl" and c["ranks_seen"] == 8 and c["distinct_devices"] == 1
    x = line["exchange"]
    assert x["bytes_sent_per_step_all_ranks"] > 0 and set(x["max_over_ranks_ms_per_step"]) == {"extract_ms", "exchange_ms", "merge_ms", "allreduce_ms"}
    per_gpu = 2000000 // (8 if scaling == "strong" else 1)
    if workload == "comp":
        assert line["kmer_instances"] == 8 * (per_gpu & ~1) * 124 + (5000000 - 5 * 26)
    else:
        assert line["kmer_instances"] == 2 * 8 * ((per_gpu // 2) & ~1) * (150 - line["config"]["k"] + 1)


def test_ranks_on_one_device_refuse_nothing_but_say_so():
    """Two ranks, transport left to the library (auto): they share this box's device, so /dev/shm staging is legitimate and the line says
    what carried the exchange; on distinct devices the same situation is an error (kg_comm.hip: katgpu_comm_init) unless --allow-shm."""
    line = _bench(["--gpus", "2"] + SMALL, {"KATGPU_ARENA_FRACTION": "0.3"}, timeout=900)
    c = line["config"]["comm"]
    assert c["transport"] == "shm" and c["distinct_devices"] == 1 and c["ranks_seen"] == 2 and line["result_accounts_for_every_kmer"]
