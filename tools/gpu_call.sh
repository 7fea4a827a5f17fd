#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null && /tmp/ubench_valu ) > gpurun_out/c1_ubench.txt 2>&1
cat gpurun_out/c1_ubench.txt
(timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py tests/test_gpu_comp_forms.py tests/test_gpu_scale_properties.py -m gpu -x -q --timeout=400 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c1_tests.log 2>&1
tail -15 gpurun_out/c1_tests.log | cut -c1-300
for lean in 1 0; do
  KATGPU_L1_LEAN=$lean timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/c1_bench_lean$lean.json 2> gpurun_out/c1_bench_lean$lean.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c1_bench_lean$lean.json").read().strip().splitlines()[-1])
    print("lean=$lean", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("lean=$lean bench failed", e); print(open("gpurun_out/c1_bench_lean$lean.err").read()[-1500:])
PY
done
