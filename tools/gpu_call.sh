#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 900 python -m pytest tests/test_gpu_bench_geometry.py "tests/test_gpu_partition.py::test_partitioned_counter_matches_oracle[512-100000-0-extra0]" "tests/test_gpu_partition.py::test_partitioned_counter_matches_oracle[1024-3000000-7-extra6]" -m gpu -x -q --timeout=600 -p no:cacheprovider --durations=6 2>&1 | tail -40) > gpurun_out/c7_tests.log 2>&1
tail -14 gpurun_out/c7_tests.log | cut -c1-300
KATGPU_TRACE=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c7_bench.json 2> gpurun_out/c7_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c7_bench.json").read().strip().splitlines()[-1])
    print("config4", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
    e = d["end_to_end"]; print("e2e", e.get("value"), e.get("seconds"), e.get("input_GB_per_s"), json.dumps(e.get("breakdown", {}).get("phases")))
    print("\n".join(e.get("breakdown", {}).get("trace", [])))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c7_bench.err").read()[-1500:])
PY
