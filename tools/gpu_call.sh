#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py tests/test_gpu_scale_properties.py -m gpu -x -q --timeout=400 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c2_tests.log 2>&1
tail -12 gpurun_out/c2_tests.log | cut -c1-300
for lean in 1 0; do
  KATGPU_L1_LEAN=$lean timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/c2_bench_lean$lean.json 2> gpurun_out/c2_bench_lean$lean.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c2_bench_lean$lean.json").read().strip().splitlines()[-1])
    print("lean=$lean", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("lean=$lean bench failed", e); print(open("gpurun_out/c2_bench_lean$lean.err").read()[-1500:])
PY
done
KATGPU_TRACE=1 timeout 300 python bench.py --workload comp-rr --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/c2_bench_rr.json 2> gpurun_out/c2_bench_rr.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c2_bench_rr.json").read().strip().splitlines()[-1])
    print("comp-rr", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("rr bench failed", e); print(open("gpurun_out/c2_bench_rr.err").read()[-1500:])
PY
grep -m3 "partition round" gpurun_out/c2_bench_rr.err
