#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py tests/test_gpu_scan.py tests/test_gpu_comm.py -m gpu -x -q --timeout=400 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c6_tests.log 2>&1
tail -8 gpurun_out/c6_tests.log | cut -c1-300
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c6_bench.json").read().strip().splitlines()[-1])
    print("config4", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
    e = d["end_to_end"]; print("e2e", e.get("value"), e.get("seconds"), e.get("input_GB_per_s"), json.dumps(e.get("breakdown", {}).get("phases")), [ (f["file"], f["GB_per_s"], f["reader_wait_ms"], f["pread_ms_per_thread"], f["h2d_ms_per_thread"], f.get("read_by")) for f in e.get("breakdown", {}).get("files", [])])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c6_bench.err").read()[-1500:])
PY
timeout 300 python bench.py --workload comp-rr --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/c6_bench_rr.json 2> gpurun_out/c6_bench_rr.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c6_bench_rr.json").read().strip().splitlines()[-1])
    print("comp-rr", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("rr bench failed", e); print(open("gpurun_out/c6_bench_rr.err").read()[-1500:])
PY
