#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/reader_bench.hip -o /tmp/reader_bench -lpthread 2>/dev/null
  timeout 150 /tmp/reader_bench /dev/shm 4; timeout 100 /tmp/reader_bench /tmp 4; uname -r; df -h /tmp /dev/shm | tail -2; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null ) > gpurun_out/c5_reader.txt 2>&1
cat gpurun_out/c5_reader.txt
(timeout 600 python -m pytest tests/test_gpu_partition.py -m gpu -x -q --timeout=400 -p no:cacheprovider -k "not L1_LEAN" 2>&1 | tail -5) > gpurun_out/c5_tests.log 2>&1
tail -3 gpurun_out/c5_tests.log | cut -c1-300
for v in "KATGPU_APPLY_UG=1" "KATGPU_APPLY_UG=2" "KATGPU_APPLY_STAMP=1"; do
  env $v timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err
  python - "$v" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/c5_bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c5_bench.err").read()[-1500:])
PY
done
grep "apply stamps" gpurun_out/c5_bench.err | tail -4
