#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 1200 python -m pytest tests/test_gpu_partition.py tests/test_gpu_scan.py tests/test_gpu_cli.py tests/test_gpu_feeder.py -m gpu -x -q --timeout=600 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c10_tests.log 2>&1
tail -6 gpurun_out/c10_tests.log | cut -c1-300
for rep in 1 2; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c10_bench.json 2> gpurun_out/c10_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c10_bench.json").read().strip().splitlines()[-1])
    print("config4", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:80])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c10_bench.err").read()[-1500:])
PY
done
KATGPU_TRACE=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/c10_bench_e2e.json 2> gpurun_out/c10_bench_e2e.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c10_bench_e2e.json").read().strip().splitlines()[-1])
    e = d["end_to_end"]; print("e2e", e.get("value"), e.get("seconds"), e.get("input_GB_per_s"), json.dumps(e.get("breakdown", {}).get("phases")), e.get("breakdown", {}).get("unaccounted_ms"))
    print("\n".join(l for l in e.get("breakdown", {}).get("trace", []) if "+" in l[:12]))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c10_bench_e2e.err").read()[-1500:])
PY
